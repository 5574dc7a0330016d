"""ctypes binding of include/rpe_amd.h (librpe_amd.so, HIP / gfx950).

There is no CPU fallback: if the shared library is missing or no HIP device is
visible, loading / handle creation raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RPE_LIB", os.path.join(_HERE, "librpe_amd.so"))   # RPE_LIB: diagnostic builds only

ABI_VERSION = 3
ORB_LEVELS = 12
PAIR_OK, PAIR_NO_DESCRIPTORS, PAIR_INSUFFICIENT_MATCHES, PAIR_NO_ESSENTIAL, PAIR_AMBIGUOUS_ESSENTIAL = 0, 1, 2, 3, 4
FEATURE_ORB, FEATURE_SIFT = 0, 1
NORM_HAMMING, NORM_L2 = 0, 1
MATCH_CROSSCHECK, MATCH_RATIO = 0, 1
STL_LIBSTDCXX, STL_MSVC = 0, 1          # whose nth_element orders the keypoints (include/rpe_amd.h RPE_STL_*)
# capacity flags (rpe_fetch_overflow)
OVF_ORB_CANDIDATES, OVF_ORB_KEYPOINTS = 1 << 0, 1 << 1
OVF_SIFT_SEEDS, OVF_SIFT_RAW, OVF_SIFT_PREFILTER, OVF_SIFT_CAP, OVF_SIFT_KEYPOINTS = 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8
SIFT_UNCAPPED_CAPACITY = 16320   # RPE_SIFT_UNCAPPED_CAPACITY: keypoints per image kept by SIFT with nfeatures = 0
MAX_MATCHES_LIMIT = 8064          # rpe_config.max_matches upper bound
CALIB_KINDS = 16
STAGE_COUNT = 12
ORDER_BGR, ORDER_RGB = 0, 1

EXPORTS = [
    "rpe_default_config", "rpe_create", "rpe_destroy", "rpe_last_error", "rpe_device_count",
    "rpe_keypoint_capacity", "rpe_device_malloc", "rpe_device_free", "rpe_memcpy_h2d", "rpe_memcpy_d2h",
    "rpe_synchronize", "rpe_host_alloc", "rpe_host_free", "rpe_host_register", "rpe_host_unregister", "rpe_estimate_batch", "rpe_estimate_batch_device", "rpe_enqueue_batch_device",
    "rpe_fetch_results", "rpe_fetch_matched_points", "rpe_orb_detect_and_compute", "rpe_orb_debug_fetch",
    "rpe_orb_pyramid_pixels", "rpe_match_hamming", "rpe_find_essential", "rpe_recover_pose",
    "rpe_set_profiling", "rpe_get_stage_ms", "rpe_stage_name",
    "rpe_sift_detect_and_compute", "rpe_sift_debug_gauss", "rpe_match_l2",
    "rpe_estimate_stream", "rpe_enqueue_stream_device",
    "rpe_bgr_to_gray_device", "rpe_bgr_to_gray", "rpe_lsd_detect",
    "rpe_fetch_overflow", "rpe_calibrate_valu", "rpe_calibrate_valu_name", "rpe_calibrate_hbm",
    "rpe_comm_unique_id", "rpe_comm_create", "rpe_comm_prepare", "rpe_comm_connect", "rpe_comm_destroy", "rpe_comm_last_error", "rpe_gather_poses",
    "rpe_comm_allreduce_max", "rpe_comm_barrier",
]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("max_batch", C.c_int32), ("feature_method", C.c_int32), ("norm_type", C.c_int32),
                ("max_matches", C.c_int32), ("nfeatures", C.c_int32), ("fast_threshold", C.c_int32),
                ("ransac_max_iters", C.c_int32), ("ransac_prob", C.c_double), ("ransac_threshold", C.c_double),
                ("match_mode", C.c_int32), ("stl_runtime", C.c_int32), ("match_ratio", C.c_double)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("lx", "<i4"), ("ly", "<i4")])

SIFT_KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                          ("octave", "<i4")])

_lib = None


class RpeError(RuntimeError):
    pass


def load():
    """dlopen librpe_amd.so; raises (loudly) when the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RpeError(f"{LIB_PATH} not found: build it with __graft_entry__.build() "
                       "(make -C relative_pose_estimation_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32p = C.c_void_p, C.c_void_p
    lib.rpe_default_config.argtypes = [C.POINTER(Config)]; lib.rpe_default_config.restype = None
    lib.rpe_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]; lib.rpe_create.restype = C.c_int
    lib.rpe_destroy.argtypes = [vp]; lib.rpe_destroy.restype = None
    lib.rpe_last_error.argtypes = [vp]; lib.rpe_last_error.restype = C.c_char_p
    lib.rpe_device_count.argtypes = []; lib.rpe_device_count.restype = C.c_int
    lib.rpe_keypoint_capacity.argtypes = [vp]; lib.rpe_keypoint_capacity.restype = C.c_int
    lib.rpe_device_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]; lib.rpe_device_malloc.restype = C.c_int
    lib.rpe_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]; lib.rpe_host_alloc.restype = C.c_int
    lib.rpe_host_free.argtypes = [vp, vp]; lib.rpe_host_free.restype = C.c_int
    lib.rpe_host_register.argtypes = [vp, vp, C.c_size_t]; lib.rpe_host_register.restype = C.c_int
    lib.rpe_host_unregister.argtypes = [vp, vp]; lib.rpe_host_unregister.restype = C.c_int
    lib.rpe_device_free.argtypes = [vp, vp]; lib.rpe_device_free.restype = C.c_int
    lib.rpe_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]; lib.rpe_memcpy_h2d.restype = C.c_int
    lib.rpe_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]; lib.rpe_memcpy_d2h.restype = C.c_int
    lib.rpe_synchronize.argtypes = [vp]; lib.rpe_synchronize.restype = C.c_int
    lib.rpe_estimate_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, i32p, i32p, i32p]
    lib.rpe_estimate_batch.restype = C.c_int
    lib.rpe_estimate_batch_device.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, i32p, i32p, i32p]
    lib.rpe_estimate_batch_device.restype = C.c_int
    lib.rpe_enqueue_batch_device.argtypes = [vp, vp, vp, C.c_int, vp]; lib.rpe_enqueue_batch_device.restype = C.c_int
    lib.rpe_fetch_results.argtypes = [vp, C.c_int, vp, vp, i32p, i32p, i32p]; lib.rpe_fetch_results.restype = C.c_int
    lib.rpe_fetch_matched_points.argtypes = [vp, C.c_int, vp, vp]; lib.rpe_fetch_matched_points.restype = C.c_int
    lib.rpe_orb_detect_and_compute.argtypes = [vp, vp, C.c_int, vp, vp, i32p]
    lib.rpe_orb_detect_and_compute.restype = C.c_int
    lib.rpe_orb_debug_fetch.argtypes = [vp, C.c_int, C.c_int, vp]; lib.rpe_orb_debug_fetch.restype = C.c_int
    lib.rpe_orb_pyramid_pixels.argtypes = [vp]; lib.rpe_orb_pyramid_pixels.restype = C.c_int64
    lib.rpe_match_hamming.argtypes = [vp, vp, i32p, vp, i32p, C.c_int, i32p, i32p, i32p, i32p]
    lib.rpe_match_hamming.restype = C.c_int
    lib.rpe_find_essential.argtypes = [vp, vp, vp, i32p, C.c_int, vp, vp, vp, i32p, i32p]
    lib.rpe_find_essential.restype = C.c_int
    lib.rpe_recover_pose.argtypes = [vp, vp, vp, vp, i32p, C.c_int, vp, vp, vp, i32p]
    lib.rpe_recover_pose.restype = C.c_int
    lib.rpe_set_profiling.argtypes = [vp, C.c_int]; lib.rpe_set_profiling.restype = C.c_int
    lib.rpe_get_stage_ms.argtypes = [vp, vp]; lib.rpe_get_stage_ms.restype = C.c_int
    lib.rpe_stage_name.argtypes = [C.c_int]; lib.rpe_stage_name.restype = C.c_char_p
    lib.rpe_sift_detect_and_compute.argtypes = [vp, vp, C.c_int, vp, vp, i32p]
    lib.rpe_sift_detect_and_compute.restype = C.c_int
    lib.rpe_sift_debug_gauss.argtypes = [vp, C.c_int, vp]; lib.rpe_sift_debug_gauss.restype = C.c_int64
    lib.rpe_match_l2.argtypes = [vp, vp, i32p, vp, i32p, C.c_int, i32p, i32p, vp, i32p]
    lib.rpe_match_l2.restype = C.c_int
    lib.rpe_estimate_stream.argtypes = [vp, vp, C.c_int, vp, vp, vp, i32p, i32p, i32p]; lib.rpe_estimate_stream.restype = C.c_int
    lib.rpe_enqueue_stream_device.argtypes = [vp, vp, C.c_int, vp]; lib.rpe_enqueue_stream_device.restype = C.c_int
    lib.rpe_bgr_to_gray_device.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]; lib.rpe_bgr_to_gray_device.restype = C.c_int
    lib.rpe_bgr_to_gray.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]; lib.rpe_bgr_to_gray.restype = C.c_int
    lib.rpe_lsd_detect.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, i32p]; lib.rpe_lsd_detect.restype = C.c_int
    lib.rpe_fetch_overflow.argtypes = [vp, C.c_int, vp]; lib.rpe_fetch_overflow.restype = C.c_int
    lib.rpe_calibrate_valu.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]; lib.rpe_calibrate_valu.restype = C.c_int
    lib.rpe_calibrate_valu_name.argtypes = [C.c_int]; lib.rpe_calibrate_valu_name.restype = C.c_char_p
    lib.rpe_calibrate_hbm.argtypes = [vp, C.POINTER(C.c_double)]; lib.rpe_calibrate_hbm.restype = C.c_int
    lib.rpe_comm_unique_id.argtypes = [vp]; lib.rpe_comm_unique_id.restype = C.c_int
    lib.rpe_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]; lib.rpe_comm_create.restype = C.c_int
    lib.rpe_comm_prepare.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]; lib.rpe_comm_prepare.restype = C.c_int
    lib.rpe_comm_connect.argtypes = [vp, vp]; lib.rpe_comm_connect.restype = C.c_int
    lib.rpe_comm_destroy.argtypes = [vp]; lib.rpe_comm_destroy.restype = C.c_int
    lib.rpe_comm_last_error.argtypes = []; lib.rpe_comm_last_error.restype = C.c_char_p
    lib.rpe_gather_poses.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]; lib.rpe_gather_poses.restype = C.c_int
    lib.rpe_comm_allreduce_max.argtypes = [vp, C.POINTER(C.c_double)]; lib.rpe_comm_allreduce_max.restype = C.c_int
    lib.rpe_comm_barrier.argtypes = [vp]; lib.rpe_comm_barrier.restype = C.c_int
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def lsd_detect(gray, capacity=8192):
    """cv2.createLineSegmentDetector(LSD_REFINE_STD).detect(gray)[0] restated: (N, 4) float64 segments
    (pose_estimator.py:160-175).  Host code inside librpe_amd.so; needs no GPU handle."""
    g = np.ascontiguousarray(gray, np.uint8)
    assert g.ndim == 2
    out = np.zeros((capacity, 4), np.float32)
    n = np.zeros(1, np.int32)
    rc = load().rpe_lsd_detect(_p(g), g.shape[1], g.shape[0], _p(out), capacity, _p(n))
    if rc != 0:
        raise RpeError(f"rpe_lsd_detect failed ({rc})")
    return out[:min(int(n[0]), capacity)].astype(np.float64)


class Engine:
    """One rpe_handle: one GPU, one stream, fixed image size and capacities."""

    def __init__(self, width, height, max_batch=1, nfeatures=4000, max_matches=500, device=0,
                 feature_method=FEATURE_ORB, norm_type=NORM_HAMMING, fast_threshold=15,
                 ransac_max_iters=1000, ransac_prob=0.999, ransac_threshold=1.0,
                 match_mode=MATCH_CROSSCHECK, match_ratio=0.75, stl_runtime=STL_LIBSTDCXX):
        self.lib = load()
        cfg = Config()
        self.lib.rpe_default_config(C.byref(cfg))
        cfg.device = device; cfg.width = width; cfg.height = height; cfg.max_batch = max_batch
        cfg.feature_method = feature_method; cfg.norm_type = norm_type
        cfg.max_matches = max_matches; cfg.nfeatures = nfeatures; cfg.fast_threshold = fast_threshold
        cfg.ransac_max_iters = ransac_max_iters; cfg.ransac_prob = ransac_prob; cfg.ransac_threshold = ransac_threshold
        cfg.match_mode = match_mode; cfg.match_ratio = match_ratio; cfg.stl_runtime = stl_runtime
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.rpe_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RpeError(f"rpe_create failed ({rc}): {self.lib.rpe_last_error(None).decode()}")
        self.h = h
        self.width, self.height, self.max_batch = width, height, max_batch
        self.max_matches = max_matches
        self.kcap = self.lib.rpe_keypoint_capacity(h)
        self.desc_dim = 128 if feature_method == FEATURE_SIFT else 32

    def close(self):
        if getattr(self, "h", None):
            self.free_pinned()
            self.lib.rpe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RpeError(f"librpe_amd error {rc}: {self.lib.rpe_last_error(self.h).decode()}")

    # ---- device buffers
    def device_malloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.rpe_device_malloc(self.h, nbytes, C.byref(p)))
        return p

    def device_free(self, p):
        self._chk(self.lib.rpe_device_free(self.h, p))

    def pinned_empty(self, shape, dtype=np.uint8):
        """numpy array on page-locked host memory (rpe_host_alloc): image batches for estimate_batch / estimate_stream that
        upload without blocking and at the full PCIe rate.  Freed with the engine (or free_pinned)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self.lib.rpe_host_alloc(self.h, n, C.byref(p)))
        buf = (C.c_uint8 * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return arr

    def free_pinned(self):
        for p in getattr(self, "_pinned", []):
            self.lib.rpe_host_free(self.h, p)
        self._pinned = []

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.device_malloc(arr.nbytes)
        self._chk(self.lib.rpe_memcpy_h2d(self.h, p, _p(arr), arr.nbytes))
        return p

    def synchronize(self):
        self._chk(self.lib.rpe_synchronize(self.h))

    # ---- image ingest (image_loader.py:27-28: BGR2GRAY)
    def bgr_to_gray(self, images, order=ORDER_BGR):
        """(..., H, W, 3) uint8 -> (..., H, W) uint8 gray, cv2's fixed-point weights, computed on the GPU."""
        a = np.ascontiguousarray(images, np.uint8)
        assert a.shape[-1] == 3, a.shape
        out = np.empty(a.shape[:-1], np.uint8)
        self._chk(self.lib.rpe_bgr_to_gray(self.h, _p(a), out.size, order, _p(out)))
        return out

    def upload_bgr_as_gray(self, images, order=ORDER_BGR):
        """Upload interleaved 3-channel frames and convert them in HBM; returns the device pointer of
        the gray frames (caller frees) -- feeds enqueue_batch_device / enqueue_stream_device."""
        a = np.ascontiguousarray(images, np.uint8)
        assert a.shape[-1] == 3, a.shape
        n = a.size // 3
        d_bgr = self.upload(a)
        d_gray = self.device_malloc(n)
        try:
            self._chk(self.lib.rpe_bgr_to_gray_device(self.h, d_bgr, n, order, d_gray))
            self.synchronize()
        finally:
            self.device_free(d_bgr)
        return d_gray

    # ---- hot path
    def _outs(self, B):
        return (np.zeros((B, 3, 3)), np.zeros((B, 3, 1)), np.zeros(B, np.int32), np.zeros(B, np.int32),
                np.zeros(B, np.int32))

    def estimate_batch(self, imgs1, imgs2, K):
        imgs1 = np.ascontiguousarray(imgs1, np.uint8); imgs2 = np.ascontiguousarray(imgs2, np.uint8)
        B = imgs1.shape[0]
        assert imgs1.shape == imgs2.shape == (B, self.height, self.width), (imgs1.shape, imgs2.shape)
        K = np.ascontiguousarray(K, np.float64)
        R, t, inl, nm, st = self._outs(B)
        self._chk(self.lib.rpe_estimate_batch(self.h, _p(imgs1), _p(imgs2), B, _p(K), _p(R), _p(t), _p(inl), _p(nm), _p(st)))
        return R, t, inl, nm, st

    def estimate_stream(self, frames, K):
        frames = np.ascontiguousarray(frames, np.uint8)
        F = frames.shape[0]
        assert frames.shape == (F, self.height, self.width)
        K = np.ascontiguousarray(K, np.float64)
        R, t, inl, nm, st = self._outs(F - 1)
        self._chk(self.lib.rpe_estimate_stream(self.h, _p(frames), F, _p(K), _p(R), _p(t), _p(inl), _p(nm), _p(st)))
        return R, t, inl, nm, st

    def enqueue_stream_device(self, d_frames, F, K):
        K = np.ascontiguousarray(K, np.float64)
        self._chk(self.lib.rpe_enqueue_stream_device(self.h, d_frames, F, _p(K)))

    def estimate_batch_device(self, d_imgs1, d_imgs2, B, K):
        K = np.ascontiguousarray(K, np.float64)
        R, t, inl, nm, st = self._outs(B)
        self._chk(self.lib.rpe_estimate_batch_device(self.h, d_imgs1, d_imgs2, B, _p(K), _p(R), _p(t), _p(inl), _p(nm), _p(st)))
        return R, t, inl, nm, st

    def enqueue_batch_device(self, d_imgs1, d_imgs2, B, K):
        K = np.ascontiguousarray(K, np.float64)
        self._chk(self.lib.rpe_enqueue_batch_device(self.h, d_imgs1, d_imgs2, B, _p(K)))

    def fetch_results(self, B):
        R, t, inl, nm, st = self._outs(B)
        self._chk(self.lib.rpe_fetch_results(self.h, B, _p(R), _p(t), _p(inl), _p(nm), _p(st)))
        return R, t, inl, nm, st

    def fetch_overflow(self, n_pairs):
        """OVF_* capacity flags of the last batch / stream, one word per pair."""
        f = np.zeros(n_pairs, np.uint32)
        self._chk(self.lib.rpe_fetch_overflow(self.h, n_pairs, _p(f)))
        return f

    def fetch_matched_points(self, B):
        p1 = np.zeros((B, self.max_matches, 2), np.float32); p2 = np.zeros_like(p1)
        self._chk(self.lib.rpe_fetch_matched_points(self.h, B, _p(p1), _p(p2)))
        return p1, p2

    # ---- stage API
    def orb_detect_and_compute(self, imgs):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        n = imgs.shape[0]
        kps = np.zeros((n, self.kcap), KP_DTYPE); desc = np.zeros((n, self.kcap, 32), np.uint8)
        cnt = np.zeros(n, np.int32)
        self._chk(self.lib.rpe_orb_detect_and_compute(self.h, _p(imgs), n, _p(kps), _p(desc), _p(cnt)))
        return kps, desc, cnt

    def orb_debug_fetch(self, index, which):
        out = np.zeros(self.lib.rpe_orb_pyramid_pixels(self.h), np.uint8)
        self._chk(self.lib.rpe_orb_debug_fetch(self.h, index, which, _p(out)))
        return out

    def sift_detect_and_compute(self, imgs):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        n = imgs.shape[0]
        kps = np.zeros((n, self.kcap), SIFT_KP_DTYPE); desc = np.zeros((n, self.kcap, 128), np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(self.lib.rpe_sift_detect_and_compute(self.h, _p(imgs), n, _p(kps), _p(desc), _p(cnt)))
        return kps, desc, cnt

    def sift_debug_gauss(self, index):
        n = self.lib.rpe_sift_debug_gauss(self.h, index, None)
        out = np.zeros(n, np.float32)
        self.lib.rpe_sift_debug_gauss(self.h, index, _p(out))
        return out

    def match_l2(self, desc1, n1, desc2, n2):
        B = len(n1)
        d1 = np.zeros((B, self.kcap, self.desc_dim), np.float32); d2 = np.zeros_like(d1)
        for i in range(B):
            d1[i, :n1[i]] = desc1[i][:n1[i]]; d2[i, :n2[i]] = desc2[i][:n2[i]]
        n1 = np.ascontiguousarray(n1, np.int32); n2 = np.ascontiguousarray(n2, np.int32)
        mm = self.max_matches
        q = np.zeros((B, mm), np.int32); t = np.zeros((B, mm), np.int32); d = np.zeros((B, mm), np.float32)
        nm = np.zeros(B, np.int32)
        self._chk(self.lib.rpe_match_l2(self.h, _p(d1), _p(n1), _p(d2), _p(n2), B, _p(q), _p(t), _p(d), _p(nm)))
        return q, t, d, nm

    def match_hamming(self, desc1, n1, desc2, n2):
        B = len(n1)
        d1 = np.zeros((B, self.kcap, 32), np.uint8); d2 = np.zeros_like(d1)
        for i in range(B):
            d1[i, :n1[i]] = desc1[i][:n1[i]]; d2[i, :n2[i]] = desc2[i][:n2[i]]
        n1 = np.ascontiguousarray(n1, np.int32); n2 = np.ascontiguousarray(n2, np.int32)
        mm = self.max_matches
        q = np.zeros((B, mm), np.int32); t = np.zeros((B, mm), np.int32); d = np.zeros((B, mm), np.int32)
        nm = np.zeros(B, np.int32)
        self._chk(self.lib.rpe_match_hamming(self.h, _p(d1), _p(n1), _p(d2), _p(n2), B, _p(q), _p(t), _p(d), _p(nm)))
        return q, t, d, nm

    def _pack_points(self, pts1, pts2):
        B = len(pts1)
        mm = self.max_matches
        p1 = np.zeros((B, mm, 2), np.float32); p2 = np.zeros_like(p1); m = np.zeros(B, np.int32)
        for i in range(B):
            m[i] = len(pts1[i]); p1[i, :m[i]] = pts1[i]; p2[i, :m[i]] = pts2[i]
        return p1, p2, m

    def find_essential(self, pts1, pts2, K):
        p1, p2, m = self._pack_points(pts1, pts2)
        B = len(m); K = np.ascontiguousarray(K, np.float64)
        E = np.zeros((B, 3, 3)); mask = np.zeros((B, self.max_matches), np.uint8)
        found = np.zeros(B, np.int32); info = np.zeros((B, 4), np.int32)
        self._chk(self.lib.rpe_find_essential(self.h, _p(p1), _p(p2), _p(m), B, _p(K), _p(E), _p(mask), _p(found), _p(info)))
        return E, mask, found, info

    def recover_pose(self, E, pts1, pts2, K):
        p1, p2, m = self._pack_points(pts1, pts2)
        B = len(m); K = np.ascontiguousarray(K, np.float64); E = np.ascontiguousarray(E, np.float64)
        R = np.zeros((B, 3, 3)); t = np.zeros((B, 3, 1)); inl = np.zeros(B, np.int32)
        self._chk(self.lib.rpe_recover_pose(self.h, _p(E), _p(p1), _p(p2), _p(m), B, _p(K), _p(R), _p(t), _p(inl)))
        return R, t, inl

    # ---- roofline calibration
    def calibrate_valu(self, kind, waves_per_simd):
        """(instruction name, measured wave-instructions per second of the whole chip)"""
        v = C.c_double(0.)
        self._chk(self.lib.rpe_calibrate_valu(self.h, kind, waves_per_simd, C.byref(v)))
        return self.lib.rpe_calibrate_valu_name(kind).decode(), float(v.value)

    def calibrate_hbm(self):
        v = C.c_double(0.)
        self._chk(self.lib.rpe_calibrate_hbm(self.h, C.byref(v)))
        return float(v.value)

    # ---- profiling
    def set_profiling(self, on):
        self._chk(self.lib.rpe_set_profiling(self.h, 1 if on else 0))

    def stage_ms(self):
        ms = np.zeros(STAGE_COUNT, np.float32)
        self._chk(self.lib.rpe_get_stage_ms(self.h, _p(ms)))
        return {self.lib.rpe_stage_name(i).decode(): float(ms[i]) for i in range(STAGE_COUNT)}
