"""The oracle against EVERY answer the reference holds for the path: all 147 result rows of its three
committed evaluation runs (evaluation-runs/*/results/evaluation_results.csv: simulator 58, Salah 80,
phone 9), on the committed image pairs, each with its run script's camera matrix.

PER-PAIR pin.  The forward Euler triple of R_prev_GT @ R_rel (batch_processor.py:82-101) is compared with the
est_yaw / est_pitch / est_roll columns.  cv2's findEssentialMat returns the model of ONE minimal five-point sample
without a refit, so a row agrees to ~1e-9 degrees only if keypoints, their ORDER, descriptors, the match list, its
sort, the RANSAC sample stream and the winning model are all cv2's -- or it does not agree at all.  Rounds 1-2 had
0 / 147 rows inside 0.01 degrees.  Three conventions, each identified from these very rows (tools/forensic*.py,
tools/agree.py), brought that to 126 / 147 inside 1e-6 degrees:
  * the descriptor blur is sepFilter2D's f32 route with fused multiply-adds, not a fixed-point kernel;
  * BFMatcher(crossCheck=True) is strict mutual nearest neighbours (OpenCV >= 4.5.x);
  * the keypoint order inside a level is what std::nth_element + std::partition leave behind in retainBest --
    libstdc++'s for the Salah and phone files (Linux wheels), the MSVC STL's for the simulator file (a Windows
    wheel produced it: 47 / 58 rows against 10 / 58 with libstdc++'s order).
The rows that still differ are low-parallax pairs whose winner hangs on one or two borderline inliers (five of the
simulator's are pairs of IDENTICAL frames): there the last bits of cv2's own SVD / solvePoly decide, and those are not
restated (geom_oracle.c says how its solver differs).
"""
import os

import numpy as np
import pytest

from relative_pose_estimation_amd import geometry
from tests import reference_rows as rr

NTHREADS = max(1, min(16, os.cpu_count() or 1))
# which C++ runtime's nth_element ordered the keypoints of each result file, and the agreement floor (rows inside
# 1e-6 degrees; measured: 47 / 71 / 8)
RUNTIME = {"sim": "msvc", "salah": "libstdc++", "phone": "libstdc++"}
FLOOR = {"sim": 45, "salah": 69, "phone": 8}


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    yield o
    o.set_stl("libstdc++")
    o.set_ransac_seed()


@pytest.mark.parametrize("name", rr.NAMES)
def test_all_reference_rows(oracle, name):
    ds = rr.load(name)
    oracle.set_ransac_seed()
    oracle.set_stl(RUNTIME[name])
    out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=NTHREADS)   # pipeline.py:94-101 parameters
    oracle.set_stl("libstdc++")
    ref = ds["ref_rotation_error"]
    assert np.all(out["status"] == 0)                       # the reference estimated every one of these pairs
    assert np.all((out["n_matches"] >= 5) & (out["n_matches"] <= 500))   # max_matches = 500 (pose_estimator.py:150-151)
    diff = rr.euler_agreement(ds, out["R"], geometry)
    err = rr.rotation_errors(ds, out["R"], geometry)
    counts = rr.agreement_counts(diff)
    print(f"\n[{name}] {len(ref)} rows, keypoint order of {RUNTIME[name]}: est_* agree within "
          + ", ".join(f"{e:g} deg: {c}" for e, c in zip(rr.AGREE_EDGES, counts)))
    print(f"[{name}] median rotation error: oracle {np.median(err):.3f} deg, reference {np.median(ref):.3f};"
          f" flips (> 90 deg): oracle {int((err > 90).sum())}, reference {int((ref > 90).sum())}")
    print(f"[{name}] rows outside 1e-6 deg (frame: difference): "
          + ", ".join(f"{int(f)}: {d:.2g}" for f, d in zip(ds["frames2"], diff) if d >= 1e-6))
    assert counts[0] >= FLOOR[name], counts
    # rotation error <= reference on identical inputs (BASELINE.json north_star): on the agreeing rows it IS the reference's
    # error; over the whole file the median may not be worse by more than the few non-agreeing rows can move it
    agree = diff < 1e-6
    assert np.allclose(err[agree], ref[agree], atol=1e-5)
    assert np.median(err) <= 1.05 * np.median(ref) + 1e-9, (np.median(err), np.median(ref))


def test_wrong_conventions_do_not_agree(oracle):
    """The pin is sharp: each of the three conventions alone, set to its round-2 value, loses the phone rows."""
    ds = rr.load("phone")
    try:
        for key, val in ((0, 0), (1, 0), (2, 1)):
            oracle.set_variant(key, val)
            out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=NTHREADS)
            n = rr.agreement_counts(rr.euler_agreement(ds, out["R"], geometry))[0]
            oracle.set_variant(key, {0: 3, 1: 3, 2: 0}[key])
            assert n <= 3, (key, val, n)
    finally:
        oracle.set_variant(0, 3); oracle.set_variant(1, 3); oracle.set_variant(2, 0)
