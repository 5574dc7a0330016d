#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the parts of the reference that ARE importable here.

Run in the build container only (needs /root/reference; the GPU box never sees it):
    python tests/golden/make_golden.py

1. geometry_golden.npz  -- outputs of the reference's src/utils/geometry.py and
   src/core/camera_calibration.py (imported by file path; `import src` itself pulls in
   cv2, which is not installed) on seeded random inputs.
2. forward_pairs.npz    -- known-answer rows: committed image pairs of the reference's
   evaluation runs (decoded from PNG with PIL + cv2's BGR2GRAY fixed-point formula),
   their ground-truth rows and the reference's own result rows
   (evaluation-runs/*/results/evaluation_results.csv).  Data only; no reference source.
3. evaluator_rows.npz   -- the first 16 rows of the simulator run's evaluation_results.csv with
   the ground-truth rows of their frames (data only): known answers for the evaluator columns.
"""
import csv
import importlib.util
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gray(path):
    a = np.asarray(Image.open(path).convert("RGB")).astype(np.int64)
    return ((a[..., 2] * 3735 + a[..., 1] * 19235 + a[..., 0] * 9798 + 16384) >> 15).astype(np.uint8)


def main():
    g = load(f"{REF}/src/utils/geometry.py", "ref_geometry")
    cc = load(f"{REF}/src/core/camera_calibration.py", "ref_calib")
    rng = np.random.default_rng(20260104)
    ang = rng.uniform(-180, 180, (64, 3))
    ang[:, 1] = rng.uniform(-89, 89, 64)
    R_yup = np.stack([g.euler_to_rotation(*a, "yup") for a in ang])
    R_zyx = np.stack([g.euler_to_rotation(*a, "zyx") for a in ang])
    e_yup = np.array([g.rotation_to_euler(R, "yup") for R in R_yup])
    e_zyx = np.array([g.rotation_to_euler(R, "zyx") for R in R_zyx])
    rot_err = np.array([g.rotation_error(R_zyx[i], R_zyx[(i + 1) % 64]) for i in range(64)])
    tv = rng.normal(size=(64, 3))
    t_err = np.array([g.translation_direction_error(tv[i], tv[(i + 7) % 64]) for i in range(64)])
    sizes = np.array([[640, 480], [1920, 1080], [848, 478], [960, 720]])
    Ks = np.stack([cc.CameraCalibration().get_matrix(int(w), int(h)) for w, h in sizes])
    np.savez(f"{OUT}/geometry_golden.npz", ang=ang, R_yup=R_yup, R_zyx=R_zyx, e_yup=e_yup, e_zyx=e_zyx,
             rot_err=rot_err, tv=tv, t_err=t_err, sizes=sizes, Ks=Ks)

    base = f"{REF}/evaluation-runs/simulator-data"
    gt = {}
    for ln in open(f"{base}/data/camera_poses.txt").read().split("\n")[1:]:
        p = ln.split()
        if len(p) == 7:
            gt[int(p[0])] = [float(v) for v in p[1:]]
    ref = {int(r["frame"]): r for r in csv.DictReader(open(f"{base}/results/evaluation_results.csv"))}
    pairs = [(0, 15), (270, 285), (420, 435)]
    K = cc.CameraCalibration().get_matrix(640, 480)
    np.savez_compressed(
        f"{OUT}/forward_pairs.npz",
        img1=np.stack([gray(f"{base}/data/images/{a:06d}.png") for a, _ in pairs]),
        img2=np.stack([gray(f"{base}/data/images/{b:06d}.png") for _, b in pairs]),
        frames=np.array(pairs), K=K,
        gt1=np.array([gt[a] for a, _ in pairs]),   # x y z roll pitch yaw of the first frame
        gt2=np.array([gt[b] for _, b in pairs]),
        ref_est=np.array([[float(ref[b][k]) for k in ("est_roll", "est_pitch", "est_yaw")] for _, b in pairs]),
        ref_rot_err=np.array([float(ref[b]["rotation_error"]) for _, b in pairs]),
        convention=np.array("yup"))
    # 3. evaluator_rows.npz -- the first rows of the reference's own evaluation_results.csv (data) with the
    # ground-truth rows of their frames: pins the evaluator's angle-error columns and the CSV column order
    rows = list(csv.DictReader(open(f"{base}/results/evaluation_results.csv")))[:16]
    cols = list(rows[0].keys())
    np.savez(f"{OUT}/evaluator_rows.npz",
             columns=np.array(cols),
             table=np.array([[float(r[c]) if r[c] != "" else np.nan for c in cols] for r in rows]),   # empty field = NaN row of the run
             gt=np.array([[int(r["frame"])] + gt[int(r["frame"])] for r in rows]))   # frame x y z roll pitch yaw
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
