#!/usr/bin/env python3
"""Writes tests/golden/reference_rows/: every image pair the reference's committed result
files cover (evaluation-runs/*/results/evaluation_results.csv: simulator 58 rows, Salah 80,
phone 9), their ground-truth rows, the camera matrix each run script uses, and the CSV rows
themselves.  DATA ONLY (images, numbers); no reference source text.

Run in the build container only (needs /root/reference; the GPU box never sees it):
    python tests/golden/make_reference_rows.py

Images: Salah / phone frames are the reference's PNG files byte for byte (4.2 + 0.8 MB);
simulator frames are stored as 8-bit gray PNGs (cv2's BGR2GRAY fixed-point formula applied
here; 7 MB instead of 22 MB of RGB) -- tests/reference_rows.py:load() applies the gray
formula to colour files and reads gray ones as they are.
"""
import csv
import io
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference/evaluation-runs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_rows")

DATASETS = {
    # name: (dir, gt file, convention, step, store gray?)  run_simulator_data.py:22-31, run_vo_database_salah.py:38-49, run_phone_data.py:22-31
    "sim": ("simulator-data", "camera_poses.txt", "yup", 15, True),
    "salah": ("vo_dataset_salah", "camera_poses_zyx.txt", "zyx", 5, False),
    "phone": ("phone-data", "camera_poses_zyx.txt", "zyx", 5, False),
}


def gray(path):
    a = np.asarray(Image.open(path).convert("RGB")).astype(np.int64)
    return ((a[..., 2] * 3735 + a[..., 1] * 19235 + a[..., 0] * 9798 + 16384) >> 15).astype(np.uint8)


def camera_matrix(name, w, h):
    if name == "sim":        # calibration_file=None -> CameraCalibration() defaults scaled to the image (camera_calibration.py:65-87)
        sx, sy = w / 960, h / 720
        return np.array([[924.82939686 * sx, 0, 468.24930789 * sx], [0, 920.4766382 * sy, 353.65863024 * sy], [0, 0, 1]])
    if name == "salah":      # run_vo_database_salah.py:12-26,38
        z = np.load(f"{REF}/vo_dataset_salah/data/calibration.npz")
        return np.asarray(z["K"] if "K" in z else z["intrinsic_matrix"], float).reshape(3, 3)
    return np.asarray(np.load(f"{REF}/phone-data/data/calibration_scaled.npz")["K"], float).reshape(3, 3)   # run_phone_data.py:25


def main():
    for name, (d, gtf, conv, step, as_gray) in DATASETS.items():
        gt = {}
        for ln in open(f"{REF}/{d}/data/{gtf}").read().split("\n")[1:]:
            p = ln.split()
            if len(p) == 7:
                gt[int(p[0])] = [float(v) for v in p[1:]]
        rows = list(csv.DictReader(open(f"{REF}/{d}/results/evaluation_results.csv")))
        cols = list(rows[0].keys())
        frames2 = [int(r["frame"]) for r in rows]
        idx = [f for f in sorted(gt) if f % step == 0]            # ground_truth_loader.py:84
        assert idx[1:] == frames2
        frames1 = idx[:-1]                                        # batch_processor.py:71-74
        os.makedirs(f"{OUT}/{name}", exist_ok=True)
        size = None
        for f in sorted(set(frames1 + frames2)):
            src = f"{REF}/{d}/data/images/{f:06d}.png"
            dst = f"{OUT}/{name}/{f:06d}.png"
            if as_gray:
                Image.fromarray(gray(src)).save(dst, format="PNG", optimize=True)
            else:
                shutil.copyfile(src, dst)
            size = Image.open(src).size
        np.savez(f"{OUT}/{name}.npz", frames1=np.array(frames1), frames2=np.array(frames2),
                 gt1=np.array([gt[f] for f in frames1]), gt2=np.array([gt[f] for f in frames2]),   # x y z roll pitch yaw
                 K=camera_matrix(name, *size), convention=np.array(conv), step=np.array(step),
                 columns=np.array(cols),
                 table=np.array([[float(r[c]) if r[c] != "" else np.nan for c in cols] for r in rows]))
        print(name, len(rows), "rows", size)


if __name__ == "__main__":
    main()
