// rpe_internal.h -- shared declarations of the MI355X (gfx950) relative-pose engine.
// Host handle, HBM workspace layout and kernel-launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/rpe_amd.h"

#define RPE_NLEVELS RPE_ORB_LEVELS
#define RPE_EDGE 31            // ORB edgeThreshold (cv2 default; pose_estimator.py:85-91 leaves it)
#define RPE_HALF_PATCH 15      // patchSize 31
#define RPE_RANSAC_CHUNK 64    // solver wave granularity: 64 RANSAC iterations per wave
#define RPE_RANSAC_MAXCHUNK 512 // largest number of iterations evaluated per launch group (8 waves per pair)
#define RPE_MAX_MODELS 10
#define RPE_GRAPH_MAX_PAIRS 16    // batches up to this many pairs are replayed as a captured hipGraph
#define RPE_MATCH_SPLIT_PAIRS 64   // batches up to this many pairs split a pair's Hamming matching over several workgroups
#define RPE_RESULT_BYTES 108     // per pair: R 72 + t 24 + inliers 4 + status 4 + n_matches 4
// FAST tile = 64 x FAST_TH output pixels, FAST_TH * 4 threads.  Unlike the resize kernel (latency bound: smaller tiles
// won), FAST is bound by instruction issue and 32-row tiles with two waves only add halo work: 4.43 -> 4.79 ms.
#ifndef FAST_TH
#define FAST_TH 64
#endif
#define RPE_FAST_TILE_CAP (FAST_TH * 16) // entries of one FAST tile list = the most strict 3x3 maxima a 64 x FAST_TH tile can hold

// ---- HBM layout of one image's pyramid-shaped buffers ---------------------
// Level l is stored with row pitch align16(w_l) at byte offset off[l] (256-B
// aligned); images are `stride` bytes apart.  The raw pyramid, the FAST score /
// blurred pyramid buffer and the NMS buffer all use this layout.
struct RpeLevel {
    int w, h, pitch;
    int quota;        // features kept on this level (orb.cpp nfeaturesPerLevel)
    int ccap;         // capacity of the level's raster corner list: clamp(w h / 64, 1024, 8192) (cv2 has none: flagged when hit)
    int corner_off;   // offset of this level inside the per-image corner array
    int kcap2;        // capacity of the candidate list after retainBest(2 quota) (FAST-score ties extend it): 4 quota + 256
    int cand_off;     // offset of this level inside per-image candidate arrays
    float scale;      // (float)pow(1.1f, l)
    long long off;    // byte offset inside the per-image pyramid buffer
    int coef_off;     // offset of xo/xa (w entries) then yo/ya (h entries) in the HOST coefficient table
    int dcoef_off;    // device table: [align128(w) packed x][align64(h) packed y], 16-B aligned, last entry replicated
    int tile0, ntile; // this level's run inside the FAST tile table (raster order inside the level)
};

struct RpeDeviceLayout {       // passed by value to kernels
    RpeLevel lv[RPE_NLEVELS];
    long long stride;          // bytes per image in pyramid-shaped buffers
    int corner_total;          // raster corner capacity per image (sum ccap)
    int cand_total;            // candidates capacity per image (sum kcap2)
    int stl;                   // C++ runtime whose nth_element orders the keypoints (rb::RT_LIBSTDCXX / rb::RT_MSVC, cfg.stl_runtime)
    int kcap;                  // keypoints capacity per image
    int fast_thr;
    // level 0 read IN PLACE from the caller's image batches when its pitch equals the image width (width % 16 == 0):
    // slot i < in_na is image i of in_a, the others image i - in_na of in_b (a consecutive-frame stream passes one
    // buffer and in_na = frames).  in_a == nullptr: level 0 was copied into the pyramid buffer like the other levels.
    const uint8_t *in_a, *in_b;
    int in_na, in_img;         // in_img = width * height
};

// base address of level l of image slot img (see in_a above)
__device__ __forceinline__ const uint8_t *rpe_level_base(const uint8_t *pyr, const RpeDeviceLayout &lay, int img, int l)
{
    if (l == 0 && lay.in_a)
        return img < lay.in_na ? lay.in_a + (long long)img * lay.in_img : lay.in_b + (long long)(img - lay.in_na) * lay.in_img;
    return pyr + (long long)img * lay.stride + lay.lv[l].off;
}

// Inclusive prefix operations over the 64 lanes with DPP row shifts / row broadcasts: 6 v_<op>_dpp instead of 6 rounds of
// ds_bpermute + select + op (~30 vector + LDS instructions).  Shifted-out lanes read the `old` operand, the identity.
// The wave total is the value of lane 63 (__builtin_amdgcn_readlane(x, 63)).
#define RPE_WAVE_SCAN(NAME, OP, IDENT)                                                                            \
    __device__ __forceinline__ int NAME(int v)                                                                    \
    {                                                                                                             \
        v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x111, 0xF, 0xF, false));      /* row_shr:1 */            \
        v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x112, 0xF, 0xF, false));      /* row_shr:2 */            \
        v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x114, 0xF, 0xF, false));      /* row_shr:4 */            \
        v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x118, 0xF, 0xF, false));      /* row_shr:8 */            \
        v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x142, 0xA, 0xF, false));      /* row_bcast:15 -> rows 1, 3 */ \
        v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x143, 0xC, 0xF, false));      /* row_bcast:31 -> rows 2, 3 */ \
        return v;                                                                                                 \
    }
#define RPE_OP_ADD(a, b) ((a) + (b))
#define RPE_OP_MAX(a, b) max((a), (b))
#define RPE_OP_MIN(a, b) min((a), (b))
RPE_WAVE_SCAN(wave_inclusive_sum, RPE_OP_ADD, 0)
RPE_WAVE_SCAN(wave_inclusive_max, RPE_OP_MAX, (int)0x80000000)
RPE_WAVE_SCAN(wave_inclusive_min, RPE_OP_MIN, 0x7FFFFFFF)
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_inclusive_sum(v), 63); }

struct RpeTile { short level, tx, ty, pad; };
// PYR_TW x PYR_TH destination tile of the resize kernel: destination origin and origin of its source window in the level below
struct RpePyrTile { short x0, y0, a0, sy0; };
// destination tile 256 x 16, ONE wave per tile: a lane computes 4 columns x 2 groups of 8 rows.  Small one-wave tiles put
// many independent windows (7.7 KB each) on a CU -- the phases of a tile overlap only through other workgroups: 64-row
// tiles of two waves 1.89 ms, 128 x 32 of one wave 1.68 -- and the longer the contiguous rows the better the mixed read /
// write stream runs: 64-wide 1.82, 128-wide 1.67, 256 x 16 1.64 ms (partial tiles at the right edge idle lanes, which a
// memory-bound kernel does not feel)
#ifndef PYR_TH
#define PYR_TH 16
#endif
#ifndef PYR_TW
#define PYR_TW 256                   // destination tile width
#endif
#define PYR_DW (PYR_TW == 256 ? 84 : PYR_TW == 128 ? 44 : 24)   // window row in dwords: 336 B = 21 x 16-B loads (origin aligned down to 16 B); 176 B / 96 B for 128- / 64-wide tiles
#ifndef PYR_THREADS
#define PYR_THREADS (PYR_TW / 4 * PYR_TH / 16)   // PYR_TW / 4 column groups x PYR_TH / 16 row-group pairs
#endif
#define PYR_ROWS (PYR_TH == 64 ? 74 : PYR_TH == 16 ? 23 : 39)   // source rows staged per tile (checked against the tables at create time)

// per-pair RANSAC state in HBM
struct RpeRansacState {
    int best_count;   // maxGoodCount
    int best_iter;
    int best_model;
    int niters;       // current loop bound
    int next_iter;    // first iteration of the next chunk
    int done;
    int found;
    int M;
    int iters_run;    // loop trip count so far (ptsetreg.cpp `iter` at exit)
    int pad_;
    double E[9];      // best model so far
};

struct RpeSiftState;

struct rpe_handle {
    rpe_config cfg;
    RpeSiftState *sift = nullptr;             // SIFT workspace (feature_method == RPE_FEATURE_SIFT)
    int img2_base = 0;                        // 0: pair p = slots (p, B+p); 1: stream, pair p = slots (p, p+1)
    int desc_bytes = 32;                      // 32 (rBRIEF) or 128 (SIFT, stored as u8)
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;          // uploads of a chunked host batch (rpe_estimate_batch), created on first use
    hipEvent_t ev_up[8] = {};                   // 'chunk c is resident' events
    bool last_chunked = false;                  // the last host batch ran in chunks: per-pair debug arrays hold its last chunk only
    // hipGraphs of the whole launch sequence of small batches (rpe_enqueue_batch_device): the drop-in's estimate() is a batch
    // of ONE pair, ~50 launches of a few microseconds each; replaying them as one graph removes the per-launch host cost
    struct GraphEntry { const uint8_t *a, *b; int B; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    std::vector<uint32_t> ovf_pairs;            // capacity flags of a chunked host batch, per pair (OR of the pair's two images), kept across its chunks
    RpeDeviceLayout lay{};
    int n_img_cap = 0;              // 2*max_batch
    // tile tables
    RpeTile *d_tiles_full = nullptr;  int n_tiles_full = 0;   // 64x16 tiles covering every level
    RpeTile *d_tiles_fast = nullptr;  int n_tiles_fast = 0;   // tiles covering [28,w-28)x[28,h-28)
    int *d_coef = nullptr;            // resize coefficient tables
    RpePyrTile *d_pyr_tiles = nullptr;  // resize tiles of levels 1..11, level l at pyr_tile_off[l], raster order
    int pyr_tile_off[RPE_NLEVELS] = {}, pyr_tile_cnt[RPE_NLEVELS] = {};
    // image-shaped buffers
    uint8_t *d_pyr = nullptr;
    uint8_t *d_bufA = nullptr;        // ONE image's blurred pyramid (rpe_orb_debug_fetch only)
    unsigned *d_tile_list = nullptr;  // [img][n_tiles_fast][RPE_FAST_TILE_CAP] score << 24 | y << 12 | x
    int *d_tile_cnt = nullptr;        // [img][n_tiles_fast]
    uint8_t *d_stage1 = nullptr, *d_stage2 = nullptr; // staging for host-image API
    // detection
    unsigned *d_hist = nullptr;       // [img][level][256]
    unsigned *d_corner = nullptr;     // [img][corner_total] raster-ordered FAST corners of a level: score << 24 | y << 12 | x
    int *d_corner_count = nullptr;    // [img][level]
    int *d_kp_lvl_count = nullptr;    // [img][level] keypoints kept per level (head of the level's candidate run)
    unsigned *d_cand_xy = nullptr;    // [img][cand_total]  y<<16|x
    float *d_cand_resp = nullptr;     // [img][cand_total]
    int *d_cand_count = nullptr;      // [img][level]
    unsigned *d_kp_xy = nullptr;      // [img][kcap]  x | y<<12 | level<<24
    float *d_kp_resp = nullptr, *d_kp_angle = nullptr;
    float2 *d_kp_pt = nullptr;
    float2 *d_kp_cs = nullptr;        // [img][kcap] (cos, sin) of the keypoint angle
    int *d_kp_count = nullptr;        // [img]
    unsigned *d_ovf = nullptr;        // [img] RPE_OVF_* capacity flags of the last extraction
    int last_pairs = 0, last_img2_base = 0;   // image slots of the last batch's pairs: (p, last_img2_base + p)
    int level0_slots = 0;             // image slots of the last ORB run (debug fetch of an in-place level 0)
    uint8_t *d_desc = nullptr;        // [img][kcap][32]
    // matching
    int *d_m_q = nullptr, *d_m_t = nullptr, *d_m_d = nullptr, *d_m_n = nullptr;
    unsigned long long *d_m_best = nullptr;   // L2 matcher: [pair][kcap] per train: packed (f32 dist bits << 18 | queryIdx) of its nearest query
    int *d_m_norm = nullptr;                  // L2 matcher: [img][kcap][2] { |u|^2, |u|^2 + 2 sum(u) }, u = byte - 128, of every descriptor
    unsigned *d_hm_best = nullptr, *d_hm_row = nullptr;   // Hamming matcher, small batches (<= RPE_MATCH_SPLIT_PAIRS pairs): election / own-nearest words in HBM
    unsigned long long *d_m_best2 = nullptr;  // L2 matcher: [pair][kcap] per query: the same key of its nearest train (the ratio mode's only list)
    float2 *d_pts1 = nullptr, *d_pts2 = nullptr;   // [pair][max_matches]
    // RANSAC
    unsigned short *d_subsets = nullptr;  // [M 0..max_matches][iters][5]
    double *d_nit_denom = nullptr;        // [(M,g)] log(1-(1-ep)^5) or NaN-coded flags
    int *d_nit_round = nullptr;           // [(M,g)] cvRound(num/denom), -1 => denom<DBL_MIN (return 0)
    double nit_num = 0;                   // log(1-p)
    RpeRansacState *d_rstate = nullptr;
    double2 *d_n1 = nullptr, *d_n2 = nullptr;   // K-normalised matched points [pair][max_matches]
    int *d_found = nullptr;                   // [pair] 0 none, 1 one model, n > 1: n stacked models (exactly 5 matches)
    double *d_hyp = nullptr;              // [pair][88][64] per-hypothesis record between the two solver kernels
    double *d_models = nullptr;           // [pair][CHUNK][10][9]
    int *d_nmodels = nullptr;             // [pair][MAXCHUNK]
    int *d_counts = nullptr;              // [pair][MAXCHUNK][10] inlier counts of the current chunk
    uint8_t *d_mask = nullptr;            // [pair][max_matches]
    // results
    double *d_R = nullptr, *d_t = nullptr, *d_E = nullptr;
    int *d_inliers = nullptr, *d_status = nullptr;
    double K_last[9] = {0}; bool K_valid = false;        // camera matrix resident in d_K (re-uploaded only when it changes)
    uint8_t *d_resall = nullptr; unsigned *d_ovfall = nullptr;   // whole-batch result block / flag words of a chunked host batch (created on first use)
    uint8_t *d_resblk = nullptr, *h_resblk = nullptr;   // d_R, d_t, d_inliers, d_status, d_m_n live in d_resblk; pinned host mirror
    double *d_K = nullptr;
    // profiling
    bool profiling = false;
    hipEvent_t ev[RPE_STAGE_COUNT + 1] = {};
    float stage_ms[RPE_STAGE_COUNT] = {};
    bool ev_valid = false;
    std::vector<void *> user_allocs;
};

// ---- kernel launchers (defined in the .hip files) --------------------------
void rpe_launch_pyramid(rpe_handle *h, int n_img);
void rpe_launch_fast(rpe_handle *h, int n_img);
void rpe_launch_nms(rpe_handle *h, int n_img);
void rpe_launch_select(rpe_handle *h, int n_img);
void rpe_launch_harris(rpe_handle *h, int n_img);
void rpe_launch_keypoints(rpe_handle *h, int n_img);
void rpe_launch_angle(rpe_handle *h, int n_img);
void rpe_launch_blur(rpe_handle *h, int n_img);
void rpe_launch_describe(rpe_handle *h, int n_img);
void rpe_launch_match(rpe_handle *h, int B);
void rpe_launch_match_l2(rpe_handle *h, int B);
int rpe_sift_create(rpe_handle *h);
void rpe_sift_destroy(rpe_handle *h);
int rpe_sift_run(rpe_handle *h, const uint8_t *d_a, const uint8_t *d_b, int na, int nb);
int rpe_sift_fetch(rpe_handle *h, int n_images, float *fin_host, int *counts);
int rpe_sift_fetch_gauss(rpe_handle *h, int index, float *out);
long long rpe_sift_gauss_floats(rpe_handle *h);
void rpe_launch_ransac(rpe_handle *h, int B, bool want_mask);
void rpe_launch_pose(rpe_handle *h, int B, bool set_status);

// per-stage hipEvents on the handle's stream (rpe_set_profiling / rpe_get_stage_ms)
#define MARK(h, stage) do { if ((h)->profiling) hipEventRecord((h)->ev[stage], (h)->stream); } while (0)
