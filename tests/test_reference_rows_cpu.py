"""The oracle against EVERY answer the reference holds for the path: all 147 result rows of its three
committed evaluation runs (evaluation-runs/*/results/evaluation_results.csv: simulator 58, Salah 80,
phone 9), on the committed image pairs, each with its run script's camera matrix.

What can and cannot agree: cv2 is not available offline, so keypoint ORDER (std::nth_element-defined in
cv2) differs from the oracle's raster order; matcher ties and the top-500 cut then pick other matches and the
fixed-seed RANSAC sample stream lands on other points.  One pair's error therefore moves as under a change of
the RANSAC seed, and agreement is statistical.  The test measures that spread (RANSAC + recoverPose replayed
from the matched points under other seeds) and asserts that the reference's per-dataset median lies inside
it, that the oracle's typical median is not worse than the reference's, and that its default-seed run is
not an outlier of its own spread.  Numbers observed are in DESIGN.md section 2.
"""
import os

import numpy as np
import pytest

from relative_pose_estimation_amd import geometry
from tests import reference_rows as rr

NSEEDS = 24
NTHREADS = max(1, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    yield o
    o.set_ransac_seed()


def _run(oracle, name):
    ds = rr.load(name)
    oracle.set_ransac_seed()
    out, pts = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=NTHREADS,
                                          return_points=True)                  # pipeline.py:94-101 parameters
    err = rr.rotation_errors(ds, out["R"], geometry)
    sweep = []
    for s in range(NSEEDS):
        oracle.set_ransac_seed((0x9E3779B97F4A7C15 * (s + 1)) & 0xFFFFFFFFFFFFFFFF)
        o2 = oracle.pose_from_points_batch(pts, out["n_matches"], ds["K"], NTHREADS)
        assert np.all(o2["status"] == 0)
        sweep.append(rr.rotation_errors(ds, o2["R"], geometry))
    oracle.set_ransac_seed()
    # replaying the default seed from the points reproduces the end-to-end run bit for bit
    o3 = oracle.pose_from_points_batch(pts, out["n_matches"], ds["K"], NTHREADS)
    assert np.array_equal(o3["R"], out["R"]) and np.array_equal(o3["inliers"], out["inliers"])
    return ds, out, err, np.array(sweep)


def _report(name, ds, err, sweep):
    ref = ds["ref_rotation_error"]
    meds = np.median(sweep, axis=1)
    lo, hi = sweep.min(0), sweep.max(0)
    edges = [0, 0.25, 0.5, 1, 2, 5, 10, 45, 90, 181]
    print(f"\n[{name}] {len(ref)} pairs: oracle median {np.median(err):.3f} deg (default seed), reference {np.median(ref):.3f};"
          f" seed sweep x{len(sweep)}: medians min {meds.min():.3f} / median {np.median(meds):.3f} / max {meds.max():.3f}")
    print(f"[{name}] reference error inside the oracle's per-pair seed range: {int(((ref >= lo) & (ref <= hi)).sum())}/{len(ref)};"
          f" flips (> 90 deg): oracle {int((err > 90).sum())} (sweep mean {np.mean((sweep > 90).sum(1)):.1f}), reference {int((ref > 90).sum())}")
    print(f"[{name}] histogram of rotation error, edges {edges}:\n   oracle    {np.histogram(err, edges)[0].tolist()}"
          f"\n   reference {np.histogram(ref, edges)[0].tolist()}")
    d = err - ref
    print(f"[{name}] per-pair oracle - reference: within 0.5 deg {int((np.abs(d) <= 0.5).sum())}, oracle worse {int((d > 0.5).sum())}, better {int((d < -0.5).sum())}")
    return meds


@pytest.mark.parametrize("name", rr.NAMES)
def test_all_reference_rows(oracle, name):
    ds, out, err, sweep = _run(oracle, name)
    ref = ds["ref_rotation_error"]
    assert np.all(out["status"] == 0)                       # the reference estimated every one of these pairs
    assert np.all((out["n_matches"] >= 5) & (out["n_matches"] <= 500))   # max_matches = 500 (pose_estimator.py:150-151)
    meds = _report(name, ds, err, sweep)
    ref_med = np.median(ref)
    # 1. the reference's median lies inside the oracle's seed spread
    assert meds.min() <= ref_med <= meds.max(), (meds.min(), ref_med, meds.max())
    # 2. the oracle's typical median is not worse than the reference's (10 % slack = well inside the spread)
    assert np.median(meds) <= 1.10 * ref_med, (np.median(meds), ref_med)
    # 3. the default-seed run is one of that family
    assert np.median(err) <= meds.max() * 1.15
    # 4. gross failures (pose flips) are the reference's own: no more of them on average
    assert np.mean((sweep > 90).sum(1)) <= (ref > 90).sum() + 1.0
