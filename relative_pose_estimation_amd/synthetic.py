"""Seeded synthetic stereo pairs with known relative pose (own code).

Scene (camera-1 frame): three fronto-parallel textured planes at depths 4, 7, 12.
The two near planes exist only on a pseudo-random set of tiles, the far plane
everywhere, so every pixel sees real parallax and occlusion.  Texture is a
procedural multi-octave random-cell mosaic (cells of roughly 12..96 px), which
gives FAST corners on all 12 ORB pyramid levels.  Both views are ray-cast from
the same procedural scene (no image warping), then lightly blurred and noised.

Pose convention = cv2.recoverPose / the reference: X2 = R @ X1 + t, |t| = 1.
"""
import numpy as np

DEPTHS = (4.0, 7.0, 12.0)
TILE = (1.6, 2.6)          # world-size of presence tiles of the two near planes
PRESENT = (0.38, 0.5)      # fraction of tiles on which the near planes exist
# cell sizes / weights chosen so that FAST statistics match the reference's own 640x480
# frames (evaluation-runs/simulator-data: 1.1-1.3 % of pixels are FAST-15 corners, 1200-1760
# keypoints survive NMS + border filter on level 0); this texture gives ~3 % and ~1600.
OCT_PX = (12.0, 24.0, 48.0, 96.0)
OCT_W = (0.35, 0.30, 0.20, 0.15)


def _hash01(ix, iy, salt):
    h = (ix.astype(np.int64) * 73856093) ^ (iy.astype(np.int64) * 19349663) ^ (np.asarray(salt).astype(np.int64) * 83492791)
    h = h.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0x5BD1E995)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x27D4EB2F)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return (h & np.uint64(0xFFFF)).astype(np.float32) * np.float32(1.0 / 65535.0)


def _rot(yaw, pitch, roll):
    y, p, r = np.deg2rad([yaw, pitch, roll])
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    Rz = np.array([[np.cos(r), -np.sin(r), 0], [np.sin(r), np.cos(r), 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


def _render(K, R, t, W, H, seed, focal):
    """Ray-cast the scene from the camera X_c = R X_w + t."""
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    Kinv = np.linalg.inv(K)
    d_c = np.stack([u, v, np.ones_like(u)], -1) @ Kinv.T          # rays in camera frame
    d_w = d_c @ R                                                  # R^T applied to row vectors
    o = -R.T @ t
    hitX = np.zeros((H, W)); hitY = np.zeros((H, W)); plane = np.full((H, W), -1, np.int32)
    for k, depth in enumerate(DEPTHS):
        s = (depth - o[2]) / d_w[..., 2]
        X = o[0] + s * d_w[..., 0]; Y = o[1] + s * d_w[..., 1]
        if k < 2:
            solid = _hash01(np.floor(X / TILE[k]), np.floor(Y / TILE[k]), seed * 7 + k) < PRESENT[k]
        else:
            solid = np.ones((H, W), bool)
        take = (plane < 0) & solid & (s > 0)
        hitX[take] = X[take]; hitY[take] = Y[take]; plane[take] = k
    depth = np.asarray(DEPTHS)[np.clip(plane, 0, 2)]
    val = np.zeros((H, W), np.float32)
    for o_i, (px, w) in enumerate(zip(OCT_PX, OCT_W)):
        cell = px * depth / focal                                   # world size of a px-sized cell at that depth
        val += np.float32(w) * _hash01(np.floor(hitX / cell), np.floor(hitY / cell), seed * 131 + o_i * 17 + plane)
    return val


def _finish(val, rng):
    img = 20.0 + 215.0 * val
    p = np.pad(img, 1, mode="edge")
    img = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] + 4 * p[1:-1, 1:-1]) / 8.0
    img = img + rng.normal(0.0, 2.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_pair(seed, K, W=640, H=480, max_angle_deg=5.0, baseline=0.4):
    """Returns img1, img2 (uint8 HxW), R_gt (3x3), t_gt (3x1 unit)."""
    rng = np.random.default_rng(int(seed))
    yaw, pitch, roll = rng.uniform(-max_angle_deg, max_angle_deg, 3)
    R = _rot(yaw, pitch, roll)
    tdir = rng.normal(size=3); tdir /= np.linalg.norm(tdir)
    t = tdir * baseline
    K = np.asarray(K, np.float64)
    focal = 0.5 * (K[0, 0] + K[1, 1])
    v1 = _render(K, np.eye(3), np.zeros(3), W, H, int(seed), focal)
    v2 = _render(K, R, t, W, H, int(seed), focal)
    return _finish(v1, rng), _finish(v2, rng), R, tdir.reshape(3, 1)


def _job(args):
    seed, K, W, H = args
    return make_pair(seed, K, W, H)


def make_batch(n, K, W=640, H=480, cfg=2, first=0, workers=1):
    """n seeded pairs (seed = 1_000_003*cfg + index): imgs1, imgs2 [n,H,W], R [n,3,3], t [n,3,1]."""
    seeds = [1_000_003 * cfg + first + i for i in range(n)]
    jobs = [(s, np.asarray(K, np.float64), W, H) for s in seeds]
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            out = pool.map(_job, jobs, chunksize=max(1, n // (workers * 4)))
    else:
        out = [_job(j) for j in jobs]
    i1 = np.stack([o[0] for o in out]); i2 = np.stack([o[1] for o in out])
    R = np.stack([o[2] for o in out]); t = np.stack([o[3] for o in out])
    return i1, i2, R, t


def _stream_job(args):
    K, R, t, W, H, seed, focal, nseed = args
    rng = np.random.default_rng(int(nseed))
    return _finish(_render(K, R, t, W, H, int(seed), focal), rng)


def make_stream(n_frames, K, W=640, H=480, seed=5_000_011, max_angle_deg=2.0, step=0.12, workers=1, frame_range=None):
    """One camera moving through ONE scene (KITTI-like consecutive-frame stream, BASELINE config 5
    stand-in): frames [n,H,W], and the ground-truth relative pose of every consecutive pair
    (X_{i+1} = R_rel X_i + t_rel, |t_rel| = 1).  The camera random-walks with small rotations and
    a bounded position so that the scene stays in view.
    frame_range=(lo, hi): render only frames [lo, hi) of the n_frames-long sequence (the trajectory is always
    computed whole, so every shard of a sharded stream sees the same sequence); poses returned are those of the
    pairs inside the range."""
    rng = np.random.default_rng(int(seed))
    K = np.asarray(K, np.float64)
    focal = 0.5 * (K[0, 0] + K[1, 1])
    Rs, ts = [np.eye(3)], [np.zeros(3)]
    for i in range(1, n_frames):
        Rr = _rot(*rng.uniform(-max_angle_deg, max_angle_deg, 3))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        c_prev = -Rs[-1].T @ ts[-1]                       # camera centre in the scene frame
        if np.linalg.norm(c_prev) > 1.0:                  # drift back towards the origin
            d = -Rs[-1] @ (c_prev / np.linalg.norm(c_prev))
        tr = d * step
        Rs.append(Rr @ Rs[-1]); ts.append(Rr @ ts[-1] + tr)
    lo, hi = (0, n_frames) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
    jobs = [(K, Rs[i], ts[i], W, H, seed, focal, seed * 31 + i) for i in range(lo, hi)]
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            frames = pool.map(_stream_job, jobs, chunksize=max(1, len(jobs) // (workers * 4)))
    else:
        frames = [_stream_job(j) for j in jobs]
    R_rel = np.stack([Rs[i + 1] @ Rs[i].T for i in range(lo, hi - 1)])
    t_rel = np.stack([(ts[i + 1] - Rs[i + 1] @ Rs[i].T @ ts[i]) for i in range(lo, hi - 1)])
    t_rel = t_rel / np.linalg.norm(t_rel, axis=1, keepdims=True)
    return np.stack(frames), R_rel, t_rel.reshape(-1, 3, 1)
