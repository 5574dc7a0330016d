"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE -- see oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The shipped package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")
ORB_LEVELS = 12


def build(force=False):
    """Compile oracle/*.c with gcc (Makefile in this directory)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("angle", C.c_float), ("response", C.c_float),
                ("octave", C.c_int32), ("lx", C.c_int32), ("ly", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("lx", "<i4"), ("ly", "<i4")])


class Layout(C.Structure):
    _fields_ = [("w", C.c_int32 * ORB_LEVELS), ("h", C.c_int32 * ORB_LEVELS),
                ("quota", C.c_int32 * ORB_LEVELS), ("scale", C.c_float * ORB_LEVELS),
                ("offset", C.c_int64 * ORB_LEVELS), ("total", C.c_int64)]


class RansacInfo(C.Structure):
    _fields_ = [("found", C.c_int32), ("best_count", C.c_int32), ("best_iter", C.c_int32),
                ("best_model", C.c_int32), ("iters_run", C.c_int32)]


class PoseResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_kp1", C.c_int32), ("n_kp2", C.c_int32),
                ("n_matches", C.c_int32), ("inliers", C.c_int32), ("overflow", C.c_int32),
                ("R", C.c_double * 9), ("t", C.c_double * 3)]


POSE_DTYPE = np.dtype([("status", "<i4"), ("n_kp1", "<i4"), ("n_kp2", "<i4"), ("n_matches", "<i4"),
                       ("inliers", "<i4"), ("overflow", "<i4"), ("R", "<f8", (9,)), ("t", "<f8", (3,))])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_rng_next.restype = C.c_uint32
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_orb_pattern.restype = C.POINTER(C.c_int8)
        _lib.orc_ransac_update_niters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
        _lib.orc_find_essential.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double,
                                            C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        assert C.sizeof(PoseResult) == POSE_DTYPE.itemsize
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def rng_stream(n, seed=0xFFFFFFFFFFFFFFFF):
    st = C.c_uint64(seed)
    return [lib().orc_rng_next(C.byref(st)) for _ in range(n)]


def ransac_subsets(M, iters=1000):
    idx = np.zeros((iters, 5), np.int32)
    lib().orc_ransac_subsets(M, iters, _p(idx))
    return idx


def update_niters(p, ep, model_points, max_iters):
    return lib().orc_ransac_update_niters(p, ep, model_points, max_iters)


def match_hamming(d1, d2, max_matches=500):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    n1, n2 = len(d1), len(d2)
    cap = max(n1, 1)
    q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.int32)
    n = lib().orc_match_hamming(_p(d1), n1, _p(d2), n2, int(max_matches), _p(q), _p(t), _p(d))
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def match_l2(d1, d2, max_matches=500):
    d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
    n1, n2 = len(d1), len(d2)
    cap = max(n1, 1)
    q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32)
    n = lib().orc_match_l2(_p(d1), n1, _p(d2), n2, d1.shape[1], int(max_matches), _p(q), _p(t), _p(d))
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def match_hamming_ratio(d1, d2, ratio=0.75, max_matches=500):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    n1, n2 = len(d1), len(d2)
    cap = max(n1, 1)
    q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.int32)
    n = lib().orc_match_hamming_ratio(_p(d1), n1, _p(d2), n2, C.c_double(ratio), int(max_matches), _p(q), _p(t), _p(d))
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def match_l2_ratio(d1, d2, ratio=0.75, max_matches=500):
    d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
    n1, n2 = len(d1), len(d2)
    cap = max(n1, 1)
    q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32)
    n = lib().orc_match_l2_ratio(_p(d1), n1, _p(d2), n2, d1.shape[1], C.c_double(ratio), int(max_matches), _p(q), _p(t), _p(d))
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    E = np.zeros((10, 9), np.float64)
    n = lib().orc_five_point(_p(x1), _p(x2), _p(E))
    return E[:n].reshape(n, 3, 3).copy()


def find_essential(pts1, pts2, K, prob=0.999, threshold=1.0, max_iters=1000):
    pts1 = np.ascontiguousarray(pts1, np.float32); pts2 = np.ascontiguousarray(pts2, np.float32)
    K = np.ascontiguousarray(K, np.float64)
    M = len(pts1)
    E = np.zeros(9, np.float64); mask = np.zeros(max(M, 1), np.uint8); info = RansacInfo()
    ok = lib().orc_find_essential(_p(pts1), _p(pts2), M, _p(K), prob, threshold, max_iters, _p(E), _p(mask),
                                  C.addressof(info))
    inf = {f: getattr(info, f) for f, _ in RansacInfo._fields_}
    return (E.reshape(3, 3) if ok else None), mask[:M].copy(), inf


def recover_pose(E, pts1, pts2, K):
    pts1 = np.ascontiguousarray(pts1, np.float32); pts2 = np.ascontiguousarray(pts2, np.float32)
    E = np.ascontiguousarray(E, np.float64); K = np.ascontiguousarray(K, np.float64)
    R = np.zeros(9); t = np.zeros(3)
    n = lib().orc_recover_pose(_p(E), _p(pts1), _p(pts2), len(pts1), _p(K), _p(R), _p(t))
    return n, R.reshape(3, 3), t.reshape(3, 1)


def decompose_essential(E):
    E = np.ascontiguousarray(E, np.float64)
    R1 = np.zeros(9); R2 = np.zeros(9); t = np.zeros(3)
    lib().orc_decompose_essential(_p(E), _p(R1), _p(R2), _p(t))
    return R1.reshape(3, 3), R2.reshape(3, 3), t


def orb_layout(W, H, nfeatures):
    L = Layout()
    lib().orc_orb_layout_init(W, H, nfeatures, C.byref(L))
    return L


def orb_detect_and_compute(img, nfeatures=4000, fast_threshold=15, cap=None, return_flags=False):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    cap = cap or nfeatures + 64
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    flags = C.c_uint32(0)
    n = lib().orc_orb_detect_and_compute_ex(_p(img), W, H, nfeatures, fast_threshold, _p(kps), _p(desc), cap, C.byref(flags))
    if return_flags:
        return kps[:n].copy(), desc[:n].copy(), int(flags.value)
    return kps[:n].copy(), desc[:n].copy()


def build_pyramid(img, nfeatures=1000):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    L = orb_layout(W, H, nfeatures)
    pyr = np.zeros(L.total, np.uint8)
    lib().orc_orb_build_pyramid(_p(img), W, H, C.byref(L), _p(pyr))
    return pyr, L


def fast_score_map(lvl, thr=15):
    lvl = np.ascontiguousarray(lvl, np.uint8)
    h, w = lvl.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_orb_fast_score_map(_p(lvl), w, h, thr, _p(out))
    return out


def nms_map(score):
    score = np.ascontiguousarray(score, np.uint8)
    h, w = score.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_orb_nms_map(_p(score), w, h, _p(out))
    return out


def blur_level(lvl):
    lvl = np.ascontiguousarray(lvl, np.uint8)
    h, w = lvl.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_orb_blur_level(_p(lvl), w, h, _p(out))
    return out


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def orb_pattern():
    p = lib().orc_orb_pattern()
    return np.array([p[i] for i in range(1024)], np.int8).reshape(256, 4)


def estimate_pose(img1, img2, K, nfeatures=4000, max_matches=500, norm="Hamming", ratio=None):
    """max_matches None = the reference's 'no truncation' (pose_estimator.py:150-151)."""
    img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8)
    K = np.ascontiguousarray(K, np.float64)
    H, W = img1.shape
    res = PoseResult()
    mm = -1 if max_matches is None else int(max_matches)
    lib().orc_estimate_pose_orb_ratio(_p(img1), _p(img2), W, H, _p(K), nfeatures, mm, 1 if norm.upper() == "L2" else 0,
                                      C.c_double(ratio if ratio else 0.), C.byref(res))
    return {"status": res.status, "n_kp1": res.n_kp1, "n_kp2": res.n_kp2, "n_matches": res.n_matches,
            "inliers": res.inliers, "overflow": res.overflow, "R": np.array(res.R).reshape(3, 3), "t": np.array(res.t).reshape(3, 1)}


def set_ransac_seed(seed=0xFFFFFFFFFFFFFFFF):
    """Experiment knob: cv2's RANSAC seed is always (uint64)-1 (the default)."""
    lib().orc_debug_set_ransac_seed(C.c_uint64(seed))


def set_variant(key, val):
    lib().orc_debug_set_variant(int(key), int(val))


STL = {"libstdc++": 3, "msvc": 5, "libc++": 4}


def set_stl(name="libstdc++"):
    """Which C++ runtime's std::nth_element orders the keypoints inside a level (cv2's retainBest): the Linux
    wheels (default, the reference's Dockerfile), the Windows wheels ('msvc': the reference's simulator result file
    was produced by one) or the macOS wheels ('libc++')."""
    set_variant(1, STL[name])


def estimate_pose_batch(imgs1, imgs2, K, nfeatures=4000, max_matches=500, nthreads=1, method="ORB", return_points=False):
    imgs1 = np.ascontiguousarray(imgs1, np.uint8); imgs2 = np.ascontiguousarray(imgs2, np.uint8)
    K = np.ascontiguousarray(K, np.float64)
    B, H, W = imgs1.shape
    out = np.zeros(B, POSE_DTYPE)
    if return_points:
        assert method.upper() == "ORB" and max_matches > 0
        pts = np.zeros((B, 2, max_matches, 2), np.float32)
        lib().orc_estimate_pose_batch_pts(_p(imgs1), _p(imgs2), B, W, H, _p(K), nfeatures, max_matches, _p(out), nthreads, _p(pts))
        return out, pts
    if method.upper() == "SIFT":
        lib().orc_estimate_pose_sift_batch(_p(imgs1), _p(imgs2), B, W, H, _p(K), nfeatures, max_matches, _p(out), nthreads)
        return out
    lib().orc_estimate_pose_batch(_p(imgs1), _p(imgs2), B, W, H, _p(K), nfeatures, max_matches, _p(out), nthreads)
    return out


def pose_from_points_batch(pts, n_matches, K, nthreads=1):
    """findEssentialMat + recoverPose on the matched points returned by estimate_pose_batch(return_points=True)."""
    pts = np.ascontiguousarray(pts, np.float32); nm = np.ascontiguousarray(n_matches, np.int32)
    K = np.ascontiguousarray(K, np.float64)
    B, _, mm, _ = pts.shape
    out = np.zeros(B, POSE_DTYPE)
    lib().orc_pose_from_points_batch(_p(pts), _p(nm), B, mm, _p(K), _p(out), nthreads)
    return out


SIFT_KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")])


def sift_detect_and_compute(img, nfeatures=0, seed_cap=None, cap=None, return_flags=False):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    seed_cap = seed_cap or max(16384, 4 * W * H // 16)      # the HIP path's workspace rule
    cap = cap or (nfeatures + 64 if nfeatures > 0 else 4 * seed_cap)
    kps = np.zeros(cap, SIFT_KP_DTYPE); desc = np.zeros((cap, 128), np.float32)
    flags = C.c_uint32(0)
    n = lib().orc_sift_detect_and_compute_ex(_p(img), W, H, int(nfeatures), int(seed_cap), _p(kps), _p(desc), cap, C.byref(flags))
    if return_flags:
        return kps[:n].copy(), desc[:n].copy(), int(flags.value)
    return kps[:n].copy(), desc[:n].copy()


def sift_gauss_pyramid(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    dims = np.zeros(64, np.int32)
    lib().orc_sift_gauss_pyramid.restype = C.c_int64
    n = lib().orc_sift_gauss_pyramid(_p(img), W, H, None, _p(dims))
    out = np.zeros(n, np.float32)
    lib().orc_sift_gauss_pyramid(_p(img), W, H, _p(out), _p(dims))
    return out, dims
