#!/usr/bin/env python3
"""Measured vector-instruction issue rates (rpe_calibrate_valu) and HBM streaming rate of this GPU:
the roofs bench.py prices the VALU-bound kernels against.  Prints JSON."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relative_pose_estimation_amd import _capi  # noqa: E402

e = _capi.Engine(640, 480, max_batch=512, nfeatures=1000)
out = {"spec_wave_insts_per_s_at_2cyc_2.4GHz": 256 * 4 * 2.4e9 / 2, "kinds": {}}
for kind in range(_capi.CALIB_KINDS):
    row = {}
    for w in (1, 2, 4, 8):
        name, rate = e.calibrate_valu(kind, w)
        row[str(w)] = rate
    out["kinds"][name] = row
out["hbm_read_bytes_per_s"] = e.calibrate_hbm()
e.close()
print(json.dumps(out, indent=1))
