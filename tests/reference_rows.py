"""Loader for tests/golden/reference_rows/ (the image pairs, ground truth, camera matrices and result
rows of the reference's three committed evaluation runs; written by tests/golden/make_reference_rows.py).
Used by the CPU oracle test (all 147 rows) and by the GPU parity tests (HD fixtures)."""
import os

import numpy as np
from PIL import Image

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_rows")
NAMES = ("sim", "salah", "phone")


def gray(path):
    """cv2.imread + cvtColor(BGR2GRAY) (reference src/utils/image_loader.py:23-28): lossless PNG decode
    + cv2's fixed-point formula; single-channel files (the simulator fixtures) are already gray."""
    im = Image.open(path)
    if im.mode == "L":
        return np.asarray(im).copy()
    a = np.asarray(im.convert("RGB")).astype(np.int64)
    return ((a[..., 2] * 3735 + a[..., 1] * 19235 + a[..., 0] * 9798 + 16384) >> 15).astype(np.uint8)


def load(name, rows=None):
    """rows: optional index list/slice into the run's result rows."""
    z = np.load(os.path.join(DIR, name + ".npz"))
    sel = np.arange(len(z["frames2"]))[rows if rows is not None else slice(None)]
    f1, f2 = z["frames1"][sel], z["frames2"][sel]
    cache = {}

    def img(f):
        if f not in cache:
            cache[f] = gray(os.path.join(DIR, name, f"{int(f):06d}.png"))
        return cache[f]
    cols = [str(c) for c in z["columns"]]
    tab = z["table"][sel]
    return dict(name=name, K=z["K"], convention=str(z["convention"]), frames1=f1, frames2=f2,
                gt1=z["gt1"][sel], gt2=z["gt2"][sel], columns=cols, table=tab,
                ref_rotation_error=tab[:, cols.index("rotation_error")],
                img1=np.stack([img(f) for f in f1]), img2=np.stack([img(f) for f in f2]))


def rotation_errors(ds, R_rel, geometry):
    """batch_processor.py:82-101 + pose_evaluator.py:96-98: R_new = R_prev_GT @ R_rel against the GT of frame 2."""
    conv = ds["convention"]
    err = np.zeros(len(R_rel))
    for i, R in enumerate(R_rel):
        g1, g2 = ds["gt1"][i], ds["gt2"][i]
        Rp = geometry.euler_to_rotation(g1[5], g1[4], g1[3], conv)
        err[i] = geometry.rotation_error(Rp @ np.asarray(R).reshape(3, 3), geometry.euler_to_rotation(g2[5], g2[4], g2[3], conv))
    return err


def euler_agreement(ds, R_rel, geometry):
    """Per pair: largest wrapped difference (degrees) between the forward Euler triple of R_prev_GT @ R_rel
    (batch_processor.py:82-101) and the est_yaw / est_pitch / est_roll the reference's CSV holds for that row."""
    conv = ds["convention"]
    cols = ds["columns"]
    ref = ds["table"][:, [cols.index("est_yaw"), cols.index("est_pitch"), cols.index("est_roll")]]
    out = np.zeros(len(R_rel))
    for i, R in enumerate(R_rel):
        g1 = ds["gt1"][i]
        Rp = geometry.euler_to_rotation(g1[5], g1[4], g1[3], conv)
        e = np.array(geometry.rotation_to_euler(Rp @ np.asarray(R).reshape(3, 3), conv))
        d = np.abs((e - ref[i] + 180.0) % 360.0 - 180.0)
        out[i] = d.max()
    return out


AGREE_EDGES = (1e-6, 1e-3, 0.01, 0.1, 0.5)


def agreement_counts(diff):
    return [int((diff < e).sum()) for e in AGREE_EDGES]
