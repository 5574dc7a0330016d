"""Vanishing-point refinement of the relative rotation.

Post-step of PoseEstimator.estimate when `use_vp_refinement` is set and `R_prev` is given
(reference src/core/pose_estimator.py:160-481 helpers, :536-567 / :636-686 application).  It acts on R
only, per pair, after the GPU hot path has produced (R_rel, t); the reference runs it on the CPU and
so does this module: line segments come from the library's LSD restatement (rpe_lsd_detect,
csrc/lsd_host.cpp), everything after that is the small dense algebra below.

Steps (same quantities, gates and defaults as the reference):
  1. longest `vp_max_lines` segments -> homogeneous lines, angles, lengths            (:277-290)
  2. line pairs (all, or `vp_max_pairs` draws of numpy's default_rng(seed): seed 0 for the first image,
     1 for the second) vote for their intersection on a 90 x 360 (lat, lon) grid of the half sphere with
     weight |l1||l2||sin 2theta|                                                     (:292-333)
  3. VP1 = strongest cell, VP2 = best cell on the great circle orthogonal to VP1 (1 degree steps),
     VP3 = VP1 x VP2, re-orthogonalised                                              (:335-384)
  4. gates acc_max >= vp_acc_min and vp2_score >= vp_vp2_min for BOTH images         (:544-548)
  5. Levenberg-Marquardt on SO(3) for E(R) = sum_k arccos(delta_k . R d_k), left update exp(dw) R,
     accepted only if the cost drops by more than vp_cost_improve_eps                (:429-481, :553-565)
Parity: the algebra follows the reference line by line (same RNG draws); the line detector is a
restatement of cv2's LSD whose output is unpinned (no cv2 offline), so this step is 'parity unpinned'.
"""
import itertools

import numpy as np

from . import _capi

N_LAT, N_LON = 90, 360


def detect_lines(gray):
    return _capi.lsd_detect(gray)


def _segment_geometry(lines):
    d = lines[:, 2:4] - lines[:, 0:2]
    length = np.hypot(d[:, 0], d[:, 1]) + 1e-9
    angle = np.arctan2(d[:, 1], d[:, 0])
    return angle, length


def _homogeneous(lines):
    p1 = np.column_stack([lines[:, 0], lines[:, 1], np.ones(len(lines))])
    p2 = np.column_stack([lines[:, 2], lines[:, 3], np.ones(len(lines))])
    l = np.cross(p1, p2)
    return l / (np.hypot(l[:, 0], l[:, 1]) + 1e-12)[:, None]


def _pairs(m, max_pairs, seed):
    if m * (m - 1) // 2 <= max_pairs:
        return np.array(list(itertools.combinations(range(m), 2)), dtype=np.int64).reshape(-1, 2)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(max_pairs):                      # scalar draws, i then j, as the reference does
        i = int(rng.integers(0, m)); j = int(rng.integers(0, m))
        if i != j:
            out.append((i, j) if i < j else (j, i))
    return np.array(out, dtype=np.int64).reshape(-1, 2)


def _grid_index(d):
    """(lat, lon) cell of unit directions d[..., 3] on the z > 0 half sphere."""
    lat = np.rad2deg(np.arctan2(np.hypot(d[..., 0], d[..., 1]), d[..., 2]))
    lon = (np.rad2deg(np.arctan2(d[..., 1], d[..., 0])) + 360.0) % 360.0
    return np.clip(lat, 0, N_LAT - 1).astype(np.int64), np.clip(lon, 0, N_LON - 1).astype(np.int64)


def estimate_manhattan_dirs(gray, K, max_lines=120, max_pairs=3000, rng_seed=0):
    """(Delta 3x3 with the Manhattan directions as columns | None, ok, debug dict)."""
    lines = detect_lines(gray)
    dbg = {"num_lines": int(lines.shape[0])}
    if lines.shape[0] < 10:
        return None, False, dbg
    angle_all, len_all = _segment_geometry(lines)
    keep = np.argsort(-len_all)[:min(max_lines, len(lines))]
    lines, lens, angles = lines[keep], len_all[keep], angle_all[keep]
    hl = _homogeneous(lines)
    m = len(lines)
    pr = _pairs(m, max_pairs, rng_seed)
    acc = np.zeros((N_LAT, N_LON), dtype=np.float64)
    if len(pr):
        i, j = pr[:, 0], pr[:, 1]
        vp = np.cross(hl[i], hl[j])
        theta = np.abs(angles[i] - angles[j])
        theta = np.abs((theta + np.pi) % (2 * np.pi) - np.pi)
        wgt = lens[i] * lens[j] * np.abs(np.sin(2.0 * theta))
        ok = (np.abs(vp[:, 2]) >= 1e-9) & (wgt > 0)
        vp, wgt = vp[ok], wgt[ok]
        v = np.column_stack([vp[:, 0] / vp[:, 2], vp[:, 1] / vp[:, 2], np.ones(len(vp))])
        d = v @ np.linalg.inv(np.asarray(K, np.float64)).T
        d /= (np.linalg.norm(d, axis=1) + 1e-12)[:, None]
        d[d[:, 2] < 0] *= -1.0
        la, lo = _grid_index(d)
        np.add.at(acc, (la, lo), wgt)               # unbuffered: accumulates in pair order
    acc_max = float(acc.max())
    dbg["acc_max"] = acc_max
    dbg["lines_used"] = int(m)
    if acc_max <= 0:
        return None, False, dbg
    lat1, lon1 = np.unravel_index(int(np.argmax(acc)), acc.shape)
    a1, o1 = np.deg2rad(lat1 + 0.5), np.deg2rad(lon1 + 0.5)
    v1 = np.array([np.sin(a1) * np.cos(o1), np.sin(a1) * np.sin(o1), np.cos(a1)])
    v1 /= np.linalg.norm(v1) + 1e-12
    helper = np.array([1.0, 0.0, 0.0]) if abs(v1[0]) <= 0.9 else np.array([0.0, 1.0, 0.0])
    a = np.cross(v1, helper); a /= np.linalg.norm(a) + 1e-12
    b = np.cross(v1, a); b /= np.linalg.norm(b) + 1e-12
    ang = np.deg2rad(np.arange(360))
    cand = np.cos(ang)[:, None] * a + np.sin(ang)[:, None] * b
    cand /= (np.linalg.norm(cand, axis=1) + 1e-12)[:, None]
    la, lo = _grid_index(cand)
    scores = acc[la, lo]
    best = int(np.argmax(scores))                   # first maximum = the reference's strict '>' scan
    dbg["vp2_score"] = float(scores[best])
    if scores[best] <= 0:
        return None, False, dbg
    v2 = cand[best]
    v3 = np.cross(v1, v2); v3 /= np.linalg.norm(v3) + 1e-12
    v2 = np.cross(v3, v1); v2 /= np.linalg.norm(v2) + 1e-12
    return np.stack([v1, v2, v3], axis=1), True, dbg


def so3_exp(w):
    """Rodrigues formula (cv2.Rodrigues on a rotation vector)."""
    w = np.asarray(w, np.float64).reshape(3)
    th = float(np.linalg.norm(w))
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    c, s = np.cos(th), np.sin(th)
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


def vp_cost(R_iw, Delta_cam, D_world):
    s = np.clip(np.einsum("ik,ik->k", Delta_cam, R_iw @ D_world), -1.0, 1.0)
    return float(np.sum(np.arccos(s)))


def optimize_rotation_from_vps(R_init, Delta_cam, D_world, iters=12, lm_lambda=1e-2):
    R = np.array(R_init, dtype=np.float64)
    for _ in range(iters):
        U = R @ D_world                                            # columns u_k = R d_k
        s = np.clip(np.einsum("ik,ik->k", Delta_cam, U), -1.0, 1.0)
        r = np.arccos(s).reshape(3, 1)
        denom = np.sqrt(np.maximum(1e-12, 1.0 - s * s))
        J = -(np.cross(Delta_cam.T, U.T) / denom[:, None])          # rows d e_k / d w
        H = J.T @ J + lm_lambda * np.eye(3)
        g = J.T @ r
        try:
            dw = -np.linalg.solve(H, g).reshape(3)
        except np.linalg.LinAlgError:
            break
        R = so3_exp(dw) @ R
        if np.linalg.norm(dw) < 1e-7:
            break
    return R


def refine_relative_rotation(R_rel, R_prev, img1, img2, K, *, max_lines=120, max_pairs=3000, acc_min=8e5,
                             vp2_min=8000.0, iters=12, lm_lambda=1e-2, cost_improve_eps=1e-3):
    """Returns (R_rel possibly refined, vp_used, vp_debug) with the reference's debug structure (:644-684)."""
    R_prev = np.asarray(R_prev, np.float64)
    R_new_init = R_prev @ R_rel
    D1, ok1, dbg1 = estimate_manhattan_dirs(img1, K, max_lines, max_pairs, rng_seed=0)
    D2, ok2, dbg2 = estimate_manhattan_dirs(img2, K, max_lines, max_pairs, rng_seed=1)
    debug = {"prev_frame": dbg1, "new_frame": dbg2, "vp_extracted": ok1 and ok2}
    good1 = bool(ok1 and dbg1.get("acc_max", 0.0) >= acc_min and dbg1.get("vp2_score", 0.0) >= vp2_min)
    good2 = bool(ok2 and dbg2.get("acc_max", 0.0) >= acc_min and dbg2.get("vp2_score", 0.0) >= vp2_min)
    debug["reliability"] = {"prev_reliable": good1, "new_reliable": good2}
    used = False
    if good1 and good2:
        D_world = R_prev.T @ D1                                    # delta ~ R d  =>  d = R^T delta
        c0 = vp_cost(R_new_init, D2, D_world)
        R_opt = optimize_rotation_from_vps(R_new_init, D2, D_world, iters, lm_lambda)
        c1 = vp_cost(R_opt, D2, D_world)
        improved = c1 < c0 - cost_improve_eps
        debug["optimization"] = {"cost_init": c0, "cost_opt": c1, "cost_improved": bool(improved)}
        if improved:
            R_rel = R_prev.T @ R_opt
            used = True
    return R_rel, used, debug
