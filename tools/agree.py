#!/usr/bin/env python3
"""Per-pair agreement of the CPU oracle with the reference's 147 committed answers (est_* columns of the three
evaluation_results.csv files), under the oracle's convention knobs.  Prints, per dataset and knob set, how many rows
agree within 1e-6 / 1e-3 / 0.01 / 0.1 / 0.5 degrees.  DESIGN.md section 2 quotes these numbers.

    python tools/agree.py [--variant key=val ...] [--datasets sim,salah,phone] [--per-pair]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle                                    # noqa: E402
from relative_pose_estimation_amd import geometry            # noqa: E402
from tests import reference_rows as rr                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--datasets", default="sim,salah,phone")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--per-pair", action="store_true")
    a = ap.parse_args()
    for kv in a.variant:
        k, v = kv.split("=")
        oracle.set_variant(int(k), int(v))
    tot = np.zeros(len(rr.AGREE_EDGES), int)
    for name in a.datasets.split(","):
        ds = rr.load(name)
        out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=a.threads)
        diff = rr.euler_agreement(ds, out["R"], geometry)
        err = rr.rotation_errors(ds, out["R"], geometry)
        ref = ds["ref_rotation_error"]
        c = rr.agreement_counts(diff)
        tot += np.array(c)
        print(f"{name:6s} {len(diff):3d} pairs  agree <1e-6/<1e-3/<0.01/<0.1/<0.5 deg: {c}   median err oracle {np.median(err):.3f} ref {np.median(ref):.3f}"
              f"  flips {int((err > 90).sum())}/{int((ref > 90).sum())}", flush=True)
        if a.per_pair:
            for i in range(len(diff)):
                print(f"    frame {int(ds['frames2'][i]):4d} diff {diff[i]:10.6f}  err {err[i]:8.3f} ref {ref[i]:8.3f}  nm {out['n_matches'][i]} inl {out['inliers'][i]}")
    print("total", tot.tolist())


if __name__ == "__main__":
    main()
