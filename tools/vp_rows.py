"""VP refinement (pipeline.py:100 enables it for the runs behind the reference's evaluation CSVs) over the 147 committed
result rows: how often do its reliability gates pass, and does it move the rotation errors?  CPU only (oracle + host LSD).
DESIGN.md section 2 quotes the outcome."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from relative_pose_estimation_amd import geometry as g, vp_refinement as vp
from tests import reference_rows as rr
for name in rr.NAMES:
    ds = rr.load(name)
    t0 = time.time()
    out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=8)
    err0 = rr.rotation_errors(ds, out["R"], g)
    R2 = []; used = 0; ext = 0; rel = 0
    nl = []; acc = []; vp2 = []
    for i in range(len(err0)):
        g1 = ds["gt1"][i]
        Rp = g.euler_to_rotation(g1[5], g1[4], g1[3], ds["convention"])
        Rr, u, dbg = vp.refine_relative_rotation(np.asarray(out["R"][i]).reshape(3, 3), Rp, ds["img1"][i], ds["img2"][i], ds["K"])
        for fr in ("prev_frame", "new_frame"):
            d = dbg[fr]
            nl.append(d.get("num_lines", 0)); acc.append(d.get("acc_max", 0.0)); vp2.append(d.get("vp2_score", 0.0))
        R2.append(Rr); used += int(u); ext += int(dbg["vp_extracted"]); rel += int(dbg["reliability"]["prev_reliable"] and dbg["reliability"]["new_reliable"])
    err1 = rr.rotation_errors(ds, np.array(R2), g)
    ref = ds["ref_rotation_error"]
    nl, acc, vp2 = np.array(nl), np.array(acc), np.array(vp2)
    print(f"{name}: LSD segments per frame (restated detector, capped at max_lines): median {np.median(nl):.0f}, min {nl.min()}, max {nl.max()};"
          f" acc_max median {np.median(acc):.3g}, max {acc.max():.3g} (gate 8e5); vp2_score median {np.median(vp2):.3g}, max {vp2.max():.3g} (gate 8000);"
          f" frames passing both gates: {int(((acc >= 8e5) & (vp2 >= 8000)).sum())} / {len(acc)}")
    print(f"{name}: pairs {len(ref)} ref median {np.median(ref):.3f}; oracle {np.median(err0):.3f}; +VP {np.median(err1):.3f}; vp extracted {ext}, both reliable {rel}, applied {used}; changed pairs: "
          f"{[(i, round(err0[i],3), round(err1[i],3), round(ref[i],3)) for i in range(len(ref)) if abs(err0[i]-err1[i])>1e-9][:12]}  ({time.time()-t0:.0f} s)", flush=True)
