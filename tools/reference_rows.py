#!/usr/bin/env python3
"""Runs the CPU oracle over EVERY result row the reference holds (147 pairs:
simulator 58, Salah 80, phone 9) and compares with the reference's own
evaluation_results.csv.  Build-container tool: reads /root/reference (the GPU box never
sees it).  The committed fixtures tests/golden/reference_rows_*.npz are written by
tests/golden/make_reference_rows.py; this script is the experiment bench around them
(seed sweeps, convention bisection) whose numbers DESIGN.md section 2 quotes.

    python tools/reference_rows.py [--seeds N] [--variant key=val ...] [--datasets sim,salah,phone]
"""
import argparse
import csv
import ctypes as C
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle                                    # noqa: E402
from relative_pose_estimation_amd import geometry as g       # noqa: E402

REF = "/root/reference/evaluation-runs"

DATASETS = {
    # name: (dir, gt file, convention, step)  -- run_simulator_data.py:22-31, run_vo_database_salah.py:38-49, run_phone_data.py:22-31
    "sim": ("simulator-data", "camera_poses.txt", "yup", 15),
    "salah": ("vo_dataset_salah", "camera_poses_zyx.txt", "zyx", 5),
    "phone": ("phone-data", "camera_poses_zyx.txt", "zyx", 5),
}


def gray(path):
    """cv2.imread + cvtColor(BGR2GRAY) (image_loader.py:23-28): PIL decode + cv2's fixed-point formula."""
    a = np.asarray(Image.open(path).convert("RGB")).astype(np.int64)
    return ((a[..., 2] * 3735 + a[..., 1] * 19235 + a[..., 0] * 9798 + 16384) >> 15).astype(np.uint8)


def camera_matrix(name, w, h):
    if name == "sim":                                        # default CameraCalibration scaled to the image
        return g.default_camera_matrix(w, h)
    if name == "salah":                                      # run_vo_database_salah.py:12-26,38
        z = np.load(f"{REF}/vo_dataset_salah/data/calibration.npz")
        K = z["K"] if "K" in z else z["intrinsic_matrix"]
        return np.asarray(K, float).reshape(3, 3)
    z = np.load(f"{REF}/phone-data/data/calibration_scaled.npz")   # run_phone_data.py:25, camera_calibration.py (file -> 'K')
    return np.asarray(z["K"], float).reshape(3, 3)


def load_dataset(name):
    d, gtf, conv, step = DATASETS[name]
    gt = {}
    for ln in open(f"{REF}/{d}/data/{gtf}").read().split("\n")[1:]:
        p = ln.split()
        if len(p) == 7:
            gt[int(p[0])] = [float(v) for v in p[1:]]
    rows = list(csv.DictReader(open(f"{REF}/{d}/results/evaluation_results.csv")))
    frames2 = [int(r["frame"]) for r in rows]
    idx = [f for f in sorted(gt) if f % step == 0]               # ground_truth_loader.py:84 get_frame_indices(step)
    assert idx[1:] == frames2, "CSV rows must be the consecutive pairs of the ground-truth index list"
    frames1 = idx[:-1]                                           # batch_processor.py:71-74: consecutive entries of that list
    imgs = {f: gray(f"{REF}/{d}/data/images/{f:06d}.png") for f in sorted(set(frames1 + frames2))}
    h, w = imgs[frames2[0]].shape
    return dict(name=name, conv=conv, K=camera_matrix(name, w, h), gt=gt, rows=rows, frames1=frames1, frames2=frames2,
                img1=np.stack([imgs[f] for f in frames1]), img2=np.stack([imgs[f] for f in frames2]))


def run(ds, nthreads=8):
    out = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, 500, nthreads=nthreads)   # pipeline.py:94-101
    err = np.full(len(out), np.nan)
    for i, r in enumerate(out):
        if r["status"] != 0:
            continue
        g1, g2 = ds["gt"][ds["frames1"][i]], ds["gt"][ds["frames2"][i]]
        Rp = g.euler_to_rotation(g1[5], g1[4], g1[3], ds["conv"])                  # batch_processor.py:82-89
        Rn = Rp @ r["R"].reshape(3, 3)                                            # :97
        err[i] = g.rotation_error(Rn, g.euler_to_rotation(g2[5], g2[4], g2[3], ds["conv"]))
    return out, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=0)
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--datasets", default="sim,salah,phone")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--per-pair", action="store_true")
    a = ap.parse_args()
    lib = oracle.lib()
    lib.orc_debug_set_ransac_seed.argtypes = [C.c_uint64]
    for kv in a.variant:
        k, v = kv.split("=")
        lib.orc_debug_set_variant(int(k), int(v))
    for name in a.datasets.split(","):
        t0 = time.time()
        ds = load_dataset(name)
        ref = np.array([float(r["rotation_error"]) for r in ds["rows"]])
        lib.orc_debug_set_ransac_seed(0xFFFFFFFFFFFFFFFF)
        out, err = run(ds, a.threads)
        ok = ~np.isnan(err)
        print(f"== {name}: {len(ref)} pairs  K={np.round(ds['K'].diagonal()[:2], 2)}  ({time.time() - t0:.1f}s)")
        print(f"   oracle median {np.nanmedian(err):.3f}  mean {np.nanmean(err):.2f}  | reference median {np.median(ref):.3f} mean {ref.mean():.2f}"
              f" | failures {int((~ok).sum())}")
        print(f"   worse than ref+0.5: {int((err > ref + 0.5).sum())}  better than ref-0.5: {int((err < ref - 0.5).sum())}"
              f"  |err-ref|<0.01: {int((np.abs(err - ref) < 0.01).sum())}  flips(>90) oracle {int((err > 90).sum())} ref {int((ref > 90).sum())}")
        if a.per_pair:
            for i in range(len(ref)):
                print(f"     frame {ds['frames2'][i]:4d}  oracle {err[i]:8.3f}  ref {ref[i]:8.3f}  nm {out[i]['n_matches']} inl {out[i]['inliers']}")
        if a.seeds:
            meds, allerr = [], []
            for s in range(a.seeds):
                lib.orc_debug_set_ransac_seed((0x9E3779B97F4A7C15 * (s + 1)) & 0xFFFFFFFFFFFFFFFF)
                _, e = run(ds, a.threads)
                meds.append(np.nanmedian(e)); allerr.append(e)
            meds = np.array(meds); allerr = np.array(allerr)
            print(f"   seed sweep ({a.seeds}): median of medians {np.median(meds):.3f}  min {meds.min():.3f}  max {meds.max():.3f}"
                  f"  5-95% [{np.percentile(meds, 5):.3f}, {np.percentile(meds, 95):.3f}]  flips per run {np.mean((allerr > 90).sum(1)):.1f}")
            # per pair: is the reference's error inside the oracle's seed spread for that pair?
            lo, hi = np.nanmin(allerr, 0), np.nanmax(allerr, 0)
            inside = (ref >= lo - 1e-9) & (ref <= hi + 1e-9)
            print(f"   per pair: reference error inside the oracle's seed range for {int(inside.sum())}/{len(ref)} pairs;"
                  f" below it {int((ref < lo).sum())}, above it {int((ref > hi).sum())}")
            pm = np.nanmedian(allerr, 0)
            print(f"   per pair: median-over-seeds: median {np.median(pm):.3f};  pairs where ref < seed-median: {int((ref < pm).sum())}")
            lib.orc_debug_set_ransac_seed(0xFFFFFFFFFFFFFFFF)


if __name__ == "__main__":
    main()
