#!/usr/bin/env python3
"""Forensics on the reference's committed answers (CPU, bounded).  The reference's R_rel for a row is recovered from
the CSV's est_* angles (R_rel = R_prev_GT^T R(est), batch_processor.py:82-101; 'zyx' inverts in closed form, 'yup' by
Newton from the oracle's own answer).  cv2's findEssentialMat returns the model of ONE minimal 5-point sample without a
refit, so that R_rel (with some t) fits the five sampled matches to ~1e-13 while ordinary inliers sit at 1e-6..1e-4.
For every pair of the oracle's matches (i, j) the translation is fixed by t ~ c_i x c_j with c_k = (R x1_k) x x2_k;
matches k with |c_k . t| < tol are counted.  >= 5 exact fits = the reference's minimal sample exists among the ORACLE's
matches (same keypoints to the last float bit), and the positions of those five in the oracle's sorted list can be
compared with cv2's fixed sample stream for that M.
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


def ref_rel_rotation(ds, i, R_guess=None):
    conv = ds["convention"]
    cols = ds["columns"]
    est = ds["table"][i, [cols.index("est_yaw"), cols.index("est_pitch"), cols.index("est_roll")]]
    g1 = ds["gt1"][i]
    Rp = geometry.euler_to_rotation(g1[5], g1[4], g1[3], conv)
    if conv == "zyx":
        return Rp.T @ geometry.euler_to_rotation(est[0], est[1], est[2], conv)
    # 'yup': rotation_to_euler is not the inverse of euler_to_rotation: solve R_new from the triple
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation as Rot
    R0 = Rp @ R_guess

    def f(w):
        R = Rot.from_rotvec(w).as_matrix() @ R0
        e = np.array(geometry.rotation_to_euler(R, conv))
        return (e - est + 180.0) % 360.0 - 180.0
    best = None
    for seed in range(6):
        w0 = np.zeros(3) if seed == 0 else np.random.default_rng(seed).normal(0, 0.5, 3)
        s = least_squares(f, w0, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        if best is None or s.cost < best.cost:
            best = s
        if best.cost < 1e-20:
            break
    return Rp.T @ (Rot.from_rotvec(best.x).as_matrix() @ R0)


def exact_fits(R, p1, p2, K, tol):
    Ki = np.linalg.inv(K)
    x1 = (Ki @ np.c_[p1.astype(np.float64), np.ones(len(p1))].T).T
    x2 = (Ki @ np.c_[p2.astype(np.float64), np.ones(len(p2))].T).T
    c = np.cross(x1 @ R.T, x2)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    M = len(c)
    best = (0, None)
    for i in range(M):
        t = np.cross(c[i][None, :], c[i + 1:])
        n = np.linalg.norm(t, axis=1)
        ok = n > 1e-6
        if not ok.any():
            continue
        t = t[ok] / n[ok, None]
        res = np.abs(t @ c.T)                                 # (J, M)
        cnt = (res < tol).sum(1)
        j = int(cnt.argmax())
        if cnt[j] > best[0]:
            best = (int(cnt[j]), (np.nonzero(res[j] < tol)[0], t[j], np.sort(res[j])[:8]))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", default="phone,salah,sim")
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--max-matches", type=int, default=500)
    ap.add_argument("--tol", type=float, default=1e-9)
    ap.add_argument("--rows", type=int, default=0)
    a = ap.parse_args()
    for kv in a.variant:
        k, v = kv.split("=")
        oracle.set_variant(int(k), int(v))
    for name in a.datasets.split(","):
        ds = rr.load(name, rows=slice(0, a.rows) if a.rows else None)
        out, pts = oracle.estimate_pose_batch(ds["img1"], ds["img2"], ds["K"], 4000, a.max_matches, nthreads=8, return_points=True)
        hits = 0
        exact = 0
        for i in range(len(out)):
            M = int(out["n_matches"][i])
            R = ref_rel_rotation(ds, i, out["R"][i].reshape(3, 3))
            n, info = exact_fits(R, pts[i, 0, :M], pts[i, 1, :M], ds["K"], a.tol)
            hits += n >= 5
            idx = info[0].tolist() if info else []
            small = np.array2string(info[2][:6], precision=1) if info else ""
            near = ""
            if n >= 5:
                S = np.sort(oracle.ransac_subsets(M, 1000), 1)
                d = np.abs(S - np.array(idx[:5])[None, :]).sum(1)
                k = int(d.argmin())
                near = f"  nearest cv2 sample: iteration {k} {S[k].tolist()} L1 {int(d[k])}"
                exact += int(d[k]) == 0
            print(f"{name} frame {int(ds['frames2'][i]):4d} M {M:4d}  exact fits {n:2d}  at sorted positions {idx}  smallest residuals {small}{near}", flush=True)
        print(f"== {name}: reference minimal sample found among the oracle's matches for {hits}/{len(out)} pairs; at the positions of one of cv2's samples for that M: {exact}")


if __name__ == "__main__":
    main()
