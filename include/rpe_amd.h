/*
 * rpe_amd.h -- C-ABI of the MI355X-native relative-pose engine.
 *
 * Drop-in boundary for ONE path of ofekm5/relative-pose-estimation:
 *   PoseEstimator.estimate(img1, img2)  (reference src/core/pose_estimator.py:487-569)
 *     = ORB detectAndCompute x2        (:85-91, :108, :505-506)
 *     -> BFMatcher(HAMMING, crossCheck=True).match + sorted + [:max_matches]  (:131, :144-151)
 *     -> gather matched points                                         (:518-519)
 *     -> cv2.findEssentialMat(RANSAC, 0.999, 1.0)                       (:522-527)
 *     -> cv2.recoverPose                                               (:533)
 *
 * The reference has no FFI (it is Python calling cv2); the entry points below
 * are what a ctypes binding for that path binds (INTEGRATION.md shows the
 * stub).  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *  - every function returns an int status: 0 = OK, <0 = library error
 *    (rpe_last_error() gives text).  Per-pair outcomes are reported in the
 *    status[] output array (RPE_PAIR_*), never as a failed call: a batch
 *    does not abort on one bad pair (the reference raises RuntimeError,
 *    pose_estimator.py:508-509,514-515,529-530; the Python wrapper maps the
 *    per-pair code back to the same exception text).
 *  - a handle owns one HIP device, one stream and its workspaces; it is not
 *    thread-safe; different handles are independent (one per GPU).
 *  - pointers named h_* are host memory, d_* are device (HBM) memory.
 *  - images are uint8 gray, row-major, tightly packed H x W, one after another.
 *  - matrices are row-major doubles: K[9], R[9] per pair, t[3] per pair.
 *  - results are bit-deterministic run to run (integer atomics only).
 */
#ifndef RPE_AMD_H
#define RPE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPE_ABI_VERSION 3
#define RPE_ORB_LEVELS 12

/* per-pair status codes (status[] outputs) */
enum {
    RPE_PAIR_OK = 0,
    RPE_PAIR_NO_DESCRIPTORS = 1,       /* pose_estimator.py:508-509 */
    RPE_PAIR_INSUFFICIENT_MATCHES = 2, /* pose_estimator.py:514-515 */
    RPE_PAIR_NO_ESSENTIAL = 3,         /* pose_estimator.py:529-530 */
    /* exactly 5 matches and the five-point solver returned more than one model: cv2.findEssentialMat then
     * returns all of them stacked (3n x 3, ptsetreg.cpp count == modelPoints) and the reference's
     * cv2.recoverPose(E, ...) call (pose_estimator.py:533) fails its `E.cols == 3 && E.rows == 3` assertion */
    RPE_PAIR_AMBIGUOUS_ESSENTIAL = 4
};

/* Capacity flags of the last batch (rpe_fetch_overflow): a fixed-size workspace truncated a list that cv2
 * would have kept whole, so the pair's features differ from the reference's although status is RPE_PAIR_OK. */
enum {
    RPE_OVF_ORB_CANDIDATES = 1 << 0,   /* a pyramid level had more FAST corners than clamp(w h / 64, 1024, 8192), or more than 4*quota+256 survived retainBest(2*quota) (score ties) */
    RPE_OVF_ORB_KEYPOINTS  = 1 << 1,   /* more than nfeatures+64 keypoints after the Harris retainBest (ties) */
    RPE_OVF_SIFT_SEEDS     = 1 << 4,   /* more scale-space extrema than base_pixels/16 */
    RPE_OVF_SIFT_RAW       = 1 << 5,   /* more oriented keypoints than the raw list holds */
    RPE_OVF_SIFT_PREFILTER = 1 << 6,   /* response ties overflowed the pre-sort window */
    RPE_OVF_SIFT_CAP       = 1 << 7,   /* the nfeatures cap removed keypoints: differs from the reference's uncapped SIFT_create() (pose_estimator.py:93-94) */
    RPE_OVF_SIFT_KEYPOINTS = 1 << 8    /* more than nfeatures+64 keypoints after retainBest (ties), or more than RPE_SIFT_UNCAPPED_CAPACITY+64 without a cap */
};

/* library error codes (function return values) */
enum {
    RPE_OK = 0,
    RPE_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    RPE_ERR_HIP = -2,       /* HIP runtime failure (no device, OOM, launch error) */
    RPE_ERR_CAPACITY = -3   /* batch larger than the handle's max_batch */
};

enum { RPE_FEATURE_ORB = 0, RPE_FEATURE_SIFT = 1 };
#define RPE_SIFT_UNCAPPED_CAPACITY 16320   /* keypoints per image held when SIFT runs without a cap (nfeatures = 0) */
enum { RPE_NORM_HAMMING = 0, RPE_NORM_L2 = 1 };
/* RPE_MATCH_CROSSCHECK = the reference (BFMatcher(norm, crossCheck=True).match, pose_estimator.py:131,144).
 * RPE_MATCH_RATIO = opt-in extension named by the project brief, NOT in the reference: knnMatch(k=2) + Lowe's
 * ratio test (best < ratio * second best), no cross check; then the same sort / top-max_matches. */
enum { RPE_MATCH_CROSSCHECK = 0, RPE_MATCH_RATIO = 1 };
/* Which C++ runtime's std::nth_element / std::partition orders the ORB keypoints inside a pyramid level.  cv2's
 * KeyPointsFilter::retainBest (called twice per level by ORB, orb.cpp computeKeyPoints) leaves its vector in the
 * order those two library calls happen to produce, descriptor rows follow that order, and BFMatcher ties / the
 * reference's stable sort before the top-500 cut (pose_estimator.py:147-151) / the fixed-seed RANSAC then depend on
 * it.  The reference's committed result files pin it row by row: RPE_STL_LIBSTDCXX reproduces the Salah and phone
 * files (Linux wheels; the reference's Dockerfile is python:3.9-slim), RPE_STL_MSVC the simulator file (a Windows
 * wheel produced it).  The keypoint SET is the same under both. */
enum { RPE_STL_LIBSTDCXX = 0, RPE_STL_MSVC = 1 };

/* Mirrors PoseEstimator.__init__ kwargs (pose_estimator.py:19-32) plus the
 * constants hard-coded at the reference's cv2 call sites. */
typedef struct rpe_config {
    int32_t abi_version;      /* RPE_ABI_VERSION */
    int32_t device;           /* HIP device ordinal */
    int32_t width, height;    /* image size served by this handle */
    int32_t max_batch;        /* max pairs per call */
    int32_t feature_method;   /* RPE_FEATURE_ORB        (pose_estimator.py:22) */
    int32_t norm_type;        /* RPE_NORM_HAMMING       (pose_estimator.py:23) */
    int32_t max_matches;      /* default 500            (pose_estimator.py:24); 5 .. 8064; >= nfeatures+64 (the keypoint capacity) = "no truncation" (:150-151) */
    int32_t nfeatures;        /* default 4000           (pose_estimator.py:25): ORB 1 .. 8000.  SIFT: 0 = no cap, what the reference's
                                 cv2.SIFT_create() (pose_estimator.py:93-94) does -- arrays then hold RPE_SIFT_UNCAPPED_CAPACITY
                                 keypoints per image and RPE_OVF_SIFT_KEYPOINTS reports an image that has more; 1 .. 16320 = SIFT_create(nfeatures) */
    int32_t fast_threshold;   /* 15                     (pose_estimator.py:89) */
    int32_t ransac_max_iters; /* 1000 (cv2 default maxIters) */
    double  ransac_prob;      /* 0.999                  (pose_estimator.py:525) */
    double  ransac_threshold; /* 1.0 px                 (pose_estimator.py:526) */
    int32_t match_mode;       /* RPE_MATCH_CROSSCHECK   (pose_estimator.py:131) */
    int32_t stl_runtime;      /* RPE_STL_LIBSTDCXX: cv2's keypoint order on the reference's Linux wheels (see above) */
    double  match_ratio;      /* Lowe ratio for RPE_MATCH_RATIO (default 0.75; unused by the reference mode) */
} rpe_config;

typedef struct rpe_handle rpe_handle;

/* ORB keypoint record returned by the stage API (cv2.KeyPoint fields the path uses) */
typedef struct rpe_keypoint {
    float x, y;          /* kp.pt  (level coordinates * level scale) */
    float angle;         /* kp.angle, degrees */
    float response;      /* kp.response (Harris) */
    int32_t octave;      /* kp.octave */
    int32_t lx, ly;      /* integer coordinates inside the pyramid level */
} rpe_keypoint;

/* SIFT keypoint record (cv2.KeyPoint fields), stage API */
typedef struct rpe_sift_keypoint {
    float x, y;          /* kp.pt in input-image coordinates */
    float size;          /* kp.size */
    float angle;         /* kp.angle, degrees */
    float response;      /* kp.response = |contrast| */
    int32_t octave;      /* kp.octave, OpenCV's packed (octave | layer<<8 | offset<<16) form */
} rpe_sift_keypoint;

/* ------------------------------------------------------------ lifecycle */
void rpe_default_config(rpe_config *cfg);
/* replaces PoseEstimator.__init__ / _create_feature_extractor / _create_matcher
 * (pose_estimator.py:19-69, :75-96, :115-131) */
int rpe_create(const rpe_config *cfg, rpe_handle **out);
void rpe_destroy(rpe_handle *h);
const char *rpe_last_error(const rpe_handle *h); /* h may be NULL: last create error */
int rpe_device_count(void);
/* capacity of per-image keypoint arrays used by the stage API */
int rpe_keypoint_capacity(const rpe_handle *h);

/* ------------------------------------------------------- device buffers */
/* thin wrappers so a ctypes host can keep image batches resident in HBM */
int rpe_device_malloc(rpe_handle *h, size_t bytes, void **d_ptr);
int rpe_device_free(rpe_handle *h, void *d_ptr);
int rpe_memcpy_h2d(rpe_handle *h, void *d_dst, const void *h_src, size_t bytes);
int rpe_memcpy_d2h(rpe_handle *h, void *h_dst, const void *d_src, size_t bytes);
int rpe_synchronize(rpe_handle *h);
/* page-locked host memory for image batches handed to rpe_estimate_batch / rpe_estimate_stream (the reference's callers
 * hold numpy arrays from cv2.imread, src/utils/image_loader.py:23-28): uploads from pinned memory do not block the
 * calling thread and run at the full PCIe rate.  rpe_host_register pins a buffer the caller already owns. */
int rpe_host_alloc(rpe_handle *h, size_t bytes, void **h_ptr);
int rpe_host_free(rpe_handle *h, void *h_ptr);
int rpe_host_register(rpe_handle *h, void *h_ptr, size_t bytes);
int rpe_host_unregister(rpe_handle *h, void *h_ptr);

/* ------------------------------------------------------------- hot path */
/* replaces PoseEstimator.estimate (pose_estimator.py:487-533) for B pairs.
 * h_imgs1/h_imgs2: B images each (host).  Outputs (host, caller-allocated):
 * R[B*9], t[B*3], inliers[B] (= recoverPose return value, :621),
 * n_matches[B] (= len(matches), :627; may be NULL), status[B].
 * Large host batches (B >= 512 and >= 64 MiB per image set) are processed in four chunks whose uploads run on a
 * copy stream behind the kernels of the previous chunk; results are identical, but the per-pair debug arrays
 * (rpe_fetch_matched_points) then hold nothing usable and that call reports an error. */
int rpe_estimate_batch(rpe_handle *h, const uint8_t *h_imgs1, const uint8_t *h_imgs2, int B,
                       const double K[9], double *R, double *t, int32_t *inliers,
                       int32_t *n_matches, int32_t *status);
/* same, images already resident in HBM (the timed configuration) */
int rpe_estimate_batch_device(rpe_handle *h, const uint8_t *d_imgs1, const uint8_t *d_imgs2, int B,
                              const double K[9], double *R, double *t, int32_t *inliers,
                              int32_t *n_matches, int32_t *status);
/* asynchronous form: enqueue on the handle's stream, results stay on the
 * device until rpe_fetch_results(); lets the host overlap the next upload.
 * The image batches are READ IN PLACE by the ORB kernels (level 0 of the pyramid is the input itself when the
 * width is a multiple of 16 and the batches are 16-byte aligned; other shapes are copied first): they must stay
 * valid and unmodified until rpe_fetch_results() / rpe_synchronize() returns -- and until rpe_orb_debug_fetch()
 * if that is called afterwards. */
int rpe_enqueue_batch_device(rpe_handle *h, const uint8_t *d_imgs1, const uint8_t *d_imgs2, int B,
                             const double K[9]);
int rpe_fetch_results(rpe_handle *h, int B, double *R, double *t, int32_t *inliers,
                      int32_t *n_matches, int32_t *status);
/* RPE_OVF_* flags of the last batch / stream, one word per pair (the OR of its two images' flags) */
int rpe_fetch_overflow(rpe_handle *h, int n_pairs, uint32_t *flags);
/* Consecutive-frame stream = the pair loop of BatchProcessor.process_sequence (reference
 * src/core/batch_processor.py:71-109): F frames -> F-1 relative poses (frame i -> i+1), features
 * extracted once per frame.  F <= 2*max_batch, F-1 <= max_batch.  Outputs sized F-1. */
int rpe_estimate_stream(rpe_handle *h, const uint8_t *h_frames, int F, const double K[9],
                        double *R, double *t, int32_t *inliers, int32_t *n_matches, int32_t *status);
int rpe_enqueue_stream_device(rpe_handle *h, const uint8_t *d_frames, int F, const double K[9]);

/* Image ingest, the step before the path (reference src/utils/image_loader.py:23-28:
 * cv2.imread -> BGR, cv2.cvtColor(BGR2GRAY)): interleaved 3-channel uint8 images -> gray with cv2's
 * fixed-point weights, gray = (B*3735 + G*19235 + R*9798 + 16384) >> 15, on the handle's stream.
 * order: RPE_ORDER_BGR (cv2.imread layout) or RPE_ORDER_RGB (PIL layout).  n_pixels = total pixels
 * of all images (tightly packed).  The *_device form is asynchronous: its output can be handed
 * straight to rpe_enqueue_batch_device / rpe_enqueue_stream_device. */
enum { RPE_ORDER_BGR = 0, RPE_ORDER_RGB = 1 };
int rpe_bgr_to_gray_device(rpe_handle *h, const uint8_t *d_bgr, size_t n_pixels, int order, uint8_t *d_gray);
int rpe_bgr_to_gray(rpe_handle *h, const uint8_t *h_bgr, size_t n_pixels, int order, uint8_t *h_gray);

/* VP-refinement post-step (pose_estimator.py:160-175 _detect_lsd_lines): line segments of one gray image,
 * cv2.createLineSegmentDetector(LSD_REFINE_STD).detect(gray) restated (csrc/lsd_host.cpp).  Host code:
 * the reference runs this step on the CPU after the pose, and so does this library; it needs no handle.
 * h_lines[capacity*4] receives (x1, y1, x2, y2) per segment; *n_lines is the number found (may exceed
 * capacity: only the first `capacity` are stored). */
int rpe_lsd_detect(const uint8_t *h_gray, int width, int height, float *h_lines, int capacity, int32_t *n_lines);

/* matched point arrays of the last batch (estimate_with_debug's pts1/pts2,
 * pose_estimator.py:606-607,628-629): pts[B*max_matches*2] f32 */
int rpe_fetch_matched_points(rpe_handle *h, int B, float *pts1, float *pts2);

/* ---------------------------------------------------------- stage entry */
/* replaces extractor.detectAndCompute(image, None) (pose_estimator.py:108)
 * for n_images images (n_images <= 2*max_batch).  kps[n_images*cap],
 * desc[n_images*cap*32], counts[n_images]; cap = rpe_keypoint_capacity(). */
int rpe_orb_detect_and_compute(rpe_handle *h, const uint8_t *h_imgs, int n_images,
                               rpe_keypoint *kps, uint8_t *desc, int32_t *counts);
/* intermediate images of image `index` of the last rpe_orb_detect_and_compute run, in the ORACLE's
 * packed pyramid layout (levels back to back, total = sum w_l*h_l):
 * which = 0 pyramid, 2 NMS map (FAST score where the pixel survived 3x3 NMS and the 31-px border
 * filter, else 0), 3 blurred pyramid.  (The FAST score map before NMS is never materialised by the
 * fused kernel; which = 1 is rejected.)  ORB handles only. */
int rpe_orb_debug_fetch(rpe_handle *h, int index, int which, uint8_t *h_out);
int64_t rpe_orb_pyramid_pixels(const rpe_handle *h);

/* replaces matcher.match + sorted + truncate (pose_estimator.py:144-151) for B
 * descriptor-set pairs.  desc1/desc2: B*cap*32 bytes (cap = keypoint
 * capacity), n1/n2: B counts.  Outputs sized B*max_matches. */
int rpe_match_hamming(rpe_handle *h, const uint8_t *h_desc1, const int32_t *n1,
                      const uint8_t *h_desc2, const int32_t *n2, int B,
                      int32_t *qidx, int32_t *tidx, int32_t *dist, int32_t *n_matches);

/* replaces cv2.SIFT_create().detectAndCompute(image, None) (pose_estimator.py:93-94, :108); the
 * handle must have been created with feature_method = RPE_FEATURE_SIFT, norm_type = RPE_NORM_L2.
 * cfg.nfeatures is the keypoint cap (SIFT_create(nfeatures); BASELINE config 3 uses 2048 -- the
 * reference itself passes no cap).  kps[n_images*cap], desc[n_images*cap*128] f32, counts[n_images]. */
int rpe_sift_detect_and_compute(rpe_handle *h, const uint8_t *h_imgs, int n_images,
                                rpe_sift_keypoint *kps, float *desc, int32_t *counts);
/* Gaussian pyramid of image `index` of the last SIFT run (octave-major, 6 levels per octave, tight
 * rows); returns the float count (out may be NULL to query it) */
int64_t rpe_sift_debug_gauss(rpe_handle *h, int index, float *out);

/* replaces BFMatcher(NORM_L2, crossCheck=True).match + sorted + truncate (pose_estimator.py:127-131,
 * :144-151) for byte-valued descriptors given as f32: SIFT handles B*cap*128 each (cv2 returns SIFT descriptors as
 * integer-valued f32), ORB handles created with NORM_L2 B*cap*32 each.  dist: f32 L2 distances. */
int rpe_match_l2(rpe_handle *h, const float *h_desc1, const int32_t *n1, const float *h_desc2,
                 const int32_t *n2, int B, int32_t *qidx, int32_t *tidx, float *dist, int32_t *n_matches);

/* replaces cv2.findEssentialMat(pts1, pts2, K, RANSAC, prob, threshold)
 * (pose_estimator.py:522-527).  pts: B*max_matches*2 f32, m[B] counts.
 * Outputs: E[B*9], mask[B*max_matches], found[B], info[B*4] =
 * {best_count, best_iter, best_model, iters_run}. */
int rpe_find_essential(rpe_handle *h, const float *h_pts1, const float *h_pts2, const int32_t *m, int B,
                       const double K[9], double *E, uint8_t *mask, int32_t *found, int32_t *info);

/* replaces cv2.recoverPose(E, pts1, pts2, K) (pose_estimator.py:533) */
int rpe_recover_pose(rpe_handle *h, const double *h_E, const float *h_pts1, const float *h_pts2,
                     const int32_t *m, int B, const double K[9],
                     double *R, double *t, int32_t *inliers);

/* ------------------------------------------------------------ profiling */
/* Per-stage device time of the last hot-path call, from hipEvents recorded
 * on the handle's stream around each kernel group. */
enum {
    RPE_STAGE_PYRAMID = 0, RPE_STAGE_FAST, RPE_STAGE_NMS, RPE_STAGE_SELECT, RPE_STAGE_HARRIS,
    RPE_STAGE_KEYPOINTS, RPE_STAGE_ANGLE, RPE_STAGE_BLUR, RPE_STAGE_DESCRIBE, RPE_STAGE_MATCH,
    RPE_STAGE_RANSAC, RPE_STAGE_POSE, RPE_STAGE_COUNT
};
int rpe_set_profiling(rpe_handle *h, int enable);
int rpe_get_stage_ms(rpe_handle *h, float *ms /* RPE_STAGE_COUNT */);
const char *rpe_stage_name(int stage);

/* ------------------------------------------------------------- multi-GPU */
/* The path shards by independent pairs (the reference never chains estimates, batch_processor.py:82-92): one process
 * and one handle per GPU, no data-path exchange.  The single collective is the final pose gather: every rank contributes
 * per_rank 128-byte records and receives all of them -- ncclAllGather over RCCL / xGMI on the handle's stream.  librccl is
 * loaded lazily by the first rpe_comm_* call.  Bootstrap: rank 0 calls rpe_comm_unique_id and hands the 128 bytes to the
 * other ranks by any means (sharding.py uses a file next to the launcher's MASTER_PORT); every rank then calls
 * rpe_comm_create.  Record layout (RPE_POSE_RECORD_BYTES = 128): double R[9]; double t[3]; int32 inliers, status,
 * n_matches, pair (global pair index, -1 for padding); 16 bytes reserved. */
#define RPE_COMM_ID_BYTES 128
#define RPE_POSE_RECORD_BYTES 128
typedef struct rpe_comm rpe_comm;
int rpe_comm_unique_id(uint8_t id[RPE_COMM_ID_BYTES]);
int rpe_comm_create(rpe_handle *h, int rank, int world, const uint8_t id[RPE_COMM_ID_BYTES], rpe_comm **out);
/* the same in two steps, for launchers that let the ranks agree in between (sharding.PoseComm): rpe_comm_prepare does
 * everything that can fail on one rank alone (dlopen of librccl, device buffers), rpe_comm_connect enters the collective
 * ncclCommInitRank -- a rank whose local step failed never leaves the others waiting inside it */
int rpe_comm_prepare(rpe_handle *h, int rank, int world, rpe_comm **out);
int rpe_comm_connect(rpe_comm *c, const uint8_t id[RPE_COMM_ID_BYTES]);
int rpe_comm_destroy(rpe_comm *c);
const char *rpe_comm_last_error(void);
/* packs the results of the handle's last batch (n_local pairs, global indices first_pair ..) into records on the
 * device, all-gathers per_rank records per rank (n_local <= per_rank <= max_batch; padded with pair = -1) and copies
 * the world * per_rank records to h_records (host, world * per_rank * 128 bytes), rank-major. */
int rpe_gather_poses(rpe_handle *h, rpe_comm *c, int n_local, int per_rank, int first_pair, void *h_records);
/* max over ranks of one host double (step timing); rpe_comm_barrier = the same exchange without a value */
int rpe_comm_allreduce_max(rpe_comm *c, double *value);
int rpe_comm_barrier(rpe_comm *c);

/* ---------------------------------------------------- roofline calibration */
/* Measured vector-instruction ISSUE rate of this device (wave-instructions per second, whole chip) for one
 * instruction kind at `waves_per_simd` (1..8) resident waves per SIMD: the roof bench.py prices the VALU-bound
 * kernels against.  kind: 0 v_xor+v_bcnt (Hamming), 1 v_pk_min/max_i16 (FAST pair test), 2 v_perm_b32,
 * 3 v_dot4_u32_u8, 4 v_min3/v_max3_i32 (FAST score), 5 v_mad_u32_u24, 6 v_mul_f64+v_add_f64 (RANSAC, pose),
 * 7 v_fma_f64, 8 v_fma_f32, 9 v_pk_fma_f32 (the last two only to place the integer / f64 cadence next to the f32 one),
 * 10 v_dot2_u32_u16, 11 v_alignbyte_b32, 12 v_mul_lo_u32, 13 v_mad_u64_u32, 14 v_mul_u32_u24_sdwa, 15 v_pk_mad_u16
 * (which instructions are full rate and which are not decides how the integer kernels are written).
 * rpe_calibrate_hbm: measured 16-B-per-lane streaming read rate (bytes/s). */
int rpe_calibrate_valu(rpe_handle *h, int kind, int waves_per_simd, double *wave_insts_per_s);
const char *rpe_calibrate_valu_name(int kind);
int rpe_calibrate_hbm(rpe_handle *h, double *bytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* RPE_AMD_H */
