// orb_kernels.hip -- ORB detectAndCompute as hand-written gfx950 kernels.
//
// Replaces cv2.ORB_create(nfeatures, 1.1, 12, fastThreshold=15, HARRIS_SCORE)
// .detectAndCompute(image, None)  (reference src/core/pose_estimator.py:85-91,:108).
// Kernels (all images of the batch per launch; the stage names are the hipEvent slots of rpe_get_stage_ms):
//   pyramid  : pyr_resize x11: INTER_LINEAR_EXACT chain, 8.8 fixed point, 128x64 tiles, source window in LDS
//   fast     : fast_nms: FAST-9/16 score + 3x3 NMS + 31-px border filter, fused, per-tile keypoint lists;
//              tiles cover the border-filtered region only ("nms" slot is empty)
//   select   : raster_corners (a level's tile lists -> one list in FAST's raster emission order) + retain_fast:
//              retainBest(2*quota) on the FAST score, replayed as the C++ runtime's nth_element + partition (cv2's ORDER)
//   harris   : 7x7 Harris response per candidate (f32, op order = oracle)
//   keypoints: retain_harris: retainBest(quota) on the Harris response, same replay; compact_keypoints: level-major lists
//   angle    : orient_describe: one wave per keypoint: patch in LDS -> intensity-centroid angle (fastAtan2) ->
//              Gaussian 7x7 on the patch (cv2's sepFilter2D f32 route, fused multiply-adds) -> 256-bit steered BRIEF ("blur" / "describe" slots are empty; the
//              whole-level blur kernel below only serves rpe_orb_debug_fetch)
// Everything is integer or mirrored-order f32, so results equal the CPU oracle bit for bit -- and, on the reference's own
// image pairs, cv2's (tests/test_reference_rows_cpu.py, tests/test_gpu_round3.py).
#include "rpe_internal.h"
#include "rpe_devmath.h"
#include "retain_best_emul.h"

#define TW 64
#define TH 64

__device__ float4 c_pattern_f[256];         // rBRIEF pattern (brief_pattern.inc) as f32 (x0, y0, x1, y1), uploaded at handle creation
__constant__ signed char c_disc[768 * 2];   // (u,v) offsets of the radius-15 disc
__constant__ int c_ndisc;
__constant__ signed char c_circ[16 * 2] = {0,3, 1,3, 2,2, 3,1, 3,0, 3,-1, 2,-2, 1,-3, 0,-3, -1,-3, -2,-2, -3,-1, -3,0, -3,1, -2,2, -1,3};

// the same disc as packed-u8 dot-product weights: item (row v = -15..15, dword j = 0..7) covers u = -16 + 4j .. +3;
// .x = 1 per in-disc byte, .y = (u + 16) per in-disc byte (0 elsewhere)
__constant__ uint2 c_discw[31 * 8];

// taps of the descriptor blur: GaussianBlur(7x7, sigma 2) as cv2's sepFilter2D f32 route holds them -- (float)(exp(-x^2 / 8) / sum),
// getGaussianKernel's normalised f64 kernel cast to f32 (computed on the host at handle creation, like the oracle does)
__constant__ float c_gauss[7];

void rpe_orb_upload_disc(const signed char *disc, int n)
{
    hipMemcpyToSymbol(HIP_SYMBOL(c_disc), disc, (size_t)n * 2);
    hipMemcpyToSymbol(HIP_SYMBOL(c_ndisc), &n, sizeof(int));
    uint2 wt[31 * 8];
    for (int i = 0; i < 31 * 8; ++i) wt[i] = make_uint2(0u, 0u);
    for (int i = 0; i < n; ++i) {
        const int u = disc[2 * i], v = disc[2 * i + 1];
        const int j = (u + 16) >> 2, b = (u + 16) & 3;
        wt[(v + 15) * 8 + j].x |= 1u << (8 * b);
        wt[(v + 15) * 8 + j].y |= (unsigned)(u + 16) << (8 * b);
    }
    hipMemcpyToSymbol(HIP_SYMBOL(c_discw), wt, sizeof(wt));
    {
        static const signed char pat[256 * 4] = {
#include "brief_pattern.inc"
        };
        std::vector<float> pf(1024);
        for (int i = 0; i < 1024; ++i) pf[i] = (float)pat[i];
        hipMemcpyToSymbol(HIP_SYMBOL(c_pattern_f), pf.data(), sizeof(float) * 1024);
    }
    {
        double v[7], sum = 0.;
        float g[7];
        for (int i = 0; i < 7; ++i) { const double x = (double)(i - 3); v[i] = exp(-0.125 * x * x); sum += v[i]; }
        for (int i = 0; i < 7; ++i) g[i] = (float)(v[i] / sum);
        hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g, sizeof(g));
    }
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an
// XCD and its L2), so workgroup b takes tile (b % 8) * ceil(n/8) + b / 8: every XCD walks one
// contiguous run of the raster-ordered tile list and neighbouring tiles (shared halo rows and
// shared 128-B lines) hit the same L2.  Only speed depends on this, never correctness.
__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int per = (n + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

// Image-granular XCD mapping for the per-keypoint / per-candidate gather kernels: workgroups are dealt round-robin over
// the 8 XCDs (b and b + 8 share one), each XCD has its own 4 MB L2, and a gather kernel whose workgroups of ONE image
// land on all 8 XCDs pulls that image's pyramid into 8 L2s (r02 PMC: 8.75 GB fetched per step by the descriptor kernel
// for 3.3 GB of pyramids).  Linear block b -> XCD group x = b % 8, position idx = b / 8 inside the group; group x owns
// images x, x + 8, ... and walks them one after the other: img = x + 8 (idx / nb), blk = idx % nb.  Speed / traffic only.
__device__ __forceinline__ bool xcd_image_block(int nb, int n_img, int &img, int &blk)
{
    const int b = blockIdx.x, x = b & 7, idx = b >> 3;
    img = x + 8 * (idx / nb);
    blk = idx - (idx / nb) * nb;
    return img < n_img;
}
static inline unsigned xcd_image_grid(int nb, int n_img) { return (unsigned)(8 * ((n_img + 7) / 8) * nb); }

// ---------------------------------------------------------------- pyramid
// Workgroup = PYR_TW x PYR_TH destination tile of level l.  The source footprint in level l-1
// (<= PYR_ROWS rows x 152 bytes, bounds derived arithmetically so the loads do not depend on
// the coefficient tables; checked on the host) is staged in LDS with aligned 16-byte loads, all of
// them in flight before the first LDS store -- 6.9 KB per workgroup at PYR_TH = 32.
// One lane = 4 destination columns x 8 consecutive destination rows.  The bilinear chain is separable in exact
// integer arithmetic: h(src row, dst col) = a0 p[o] + a1 p[o+1] (16 bits), out = (b0 h(top) + b1 h(bottom) + 2^15) >> 16.
// Reading p[o+1] and the row below unclamped is exact: the coefficient tables give weight 0 wherever OpenCV clamps
// (last source column / row).  The row loop is described where it stands.
// PYR_TW / PYR_TH (tile size), PYR_DW / PYR_ROWS (window size) and PYR_THREADS live in rpe_internal.h: the host builds the
// tile table from them
#define PYR_NCG (PYR_TW / 4)                    // column groups of 4 pixels per tile row
#define PYR_NRG (PYR_THREADS / PYR_NCG)         // row groups of 8 rows side by side in a workgroup
#define PYR_RGPL ((PYR_TH / 8) / PYR_NRG)       // row groups per lane: ty8, ty8 + PYR_NRG, ...
__global__ __launch_bounds__(PYR_THREADS) void pyr_resize_kernel(uint8_t *pyr, RpeDeviceLayout lay, const int *__restrict__ coef,
                                                                  const RpePyrTile *__restrict__ ptiles, int ntiles, int l)
{
    __shared__ __attribute__((aligned(16))) unsigned s_src[PYR_ROWS * PYR_DW + 4];   // +4: the unclamped p[o+1] of the last row's last column
    const RpeLevel &S = lay.lv[l - 1];
    const RpeLevel &D = lay.lv[l];
    const int tid = threadIdx.x;
    const int ti = xcd_tile(blockIdx.x, ntiles);
    if (ti >= ntiles) return;
    // tile origin and source-window origin come from a host table: the two 64-bit divisions that derive the window from
    // the tile (floor(x0 * S.w / D.w), floor(y0 * S.h / D.h)) cost ~250 scalar instructions per wave when done here, and
    // the scalar unit issues at the same 1-per-4-cycles cadence per SIMD as the vector ALU
    const RpePyrTile pt = ptiles[ti];
    const int x0 = pt.x0, y0 = pt.y0, a0 = pt.a0, sy0 = pt.sy0;
    uint8_t *base = pyr + (long long)blockIdx.y * lay.stride;
    const uint8_t *src = rpe_level_base(pyr, lay, blockIdx.y, l - 1);
    // packed (offset | weight << 16) per destination column / row; the device table pads the x run to a multiple of 128
    // and the y run to a multiple of 64 entries (last entry replicated) and aligns both to 16 B, so a lane fetches its
    // 4 columns with one 16-B load and 8 rows with two, without clamps
    const int *cxp = coef + D.dcoef_off, *cyp = cxp + ((D.w + 127) & ~127);
    // The kernel lives on how many tile windows a CU keeps in flight (r02 diagnostic builds: loads + LDS alone 1.12 ms, rows +
    // stores alone 1.26 ms, together 1.92 ms -- the phases of a workgroup overlap only through OTHER workgroups, and FAST
    // run beside it on a second stream gained nothing: wave slots are the contended resource).  A lane therefore takes 4
    // columns x TWO row groups of 8 rows (half the waves per window, the column constants serve 16 rows), and the tile is
    // only PYR_TH = 32 rows high: one wave per tile, 23 independent windows per CU.
    const int tx = tid % PYR_NCG, ty8 = tid / PYR_NCG;     // row groups ty8, ty8 + PYR_NRG, ...
    const int x4 = x0 + 4 * tx;
    // coefficient loads go out first, in the shadow of the window loads
    const int4 cv = *(const int4 *)(cxp + x4);
    int4 rq[2 * PYR_RGPL];
#pragma unroll
    for (int k = 0; k < 2 * PYR_RGPL; ++k) rq[k] = *(const int4 *)(cyp + y0 + (ty8 + PYR_NRG * (k >> 1)) * 8 + 4 * (k & 1));
    {   // all window loads (16 B per lane) in flight before the first LDS store (one HBM round trip per tile):
        // chunk i = tid + 128 q of the 74 x 11 chunks of the window, row i / 11, 16-B column i % 11
        constexpr int NQ = PYR_DW / 4, NCHUNK = PYR_ROWS * NQ, NLD = (NCHUNK + PYR_THREADS - 1) / PYR_THREADS;
        uint4 stage[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = min(tid + PYR_THREADS * q, NCHUNK - 1);
            const int r = i / NQ, c = i - r * NQ;
            stage[q] = *(const uint4 *)(src + min(a0 + 16 * c, S.pitch - 16) + __umul24((unsigned)min(sy0 + r, S.h - 1), (unsigned)S.pitch));   // pitch is a multiple of 16; clamped columns are never read
        }
        // pin the loads above the guarded stores: left alone the compiler sinks every load into the block of its store
        // (load, s_waitcnt vmcnt(0), ds_write, next load ...: dependent HBM round trips instead of one)
#pragma unroll
        for (int q = 0; q < NLD; ++q) asm volatile("" : "+v"(stage[q].x), "+v"(stage[q].y), "+v"(stage[q].z), "+v"(stage[q].w));
#pragma unroll
        for (int q = 0; q < NLD; ++q) { const int i = tid + PYR_THREADS * q; if (i < NCHUNK) ((uint4 *)s_src)[i] = stage[q]; }
    }
    // per-lane column constants: source offsets o_j (non-decreasing, o_3 - o_0 <= 4) -> byte selectors, packed weights
    typedef unsigned short v2u16_t __attribute__((ext_vector_type(2)));
    unsigned selp[4];              // v_perm selector of column j: bytes (p[o_j], 0, p[o_j + 1], 0) of the 8 window bytes from o_0 on
    v2u16_t apk[4];                // (256 - a1_j, a1_j)
    int bcol;                      // window byte column of o_0
    {
        const int cw[4] = {cv.x, cv.y, cv.z, cv.w};
        const int o0 = cw[0] & 0xFFFF;
        bcol = o0 - a0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned rel = (unsigned)min(max((cw[j] & 0xFFFF) - o0, 0), 4);   // columns past D.w repeat the last entry: rel stays in range
            selp[j] = 0x0c000c00u | rel | ((rel + 1u) << 16);
            const unsigned a1 = (unsigned)cw[j] >> 16;
            apk[j] = __builtin_bit_cast(v2u16_t, (256u - a1) | (a1 << 16));
        }
    }
    // per-lane row constants, packed (LDS dword index of the top source row at the dword holding p[o_0]) | (bottom-row
    // weight << 16).  Materialised here: rematerialised inside the row loop, every row's first use of a table register sits
    // behind an s_waitcnt vmcnt(0) -- which on gfx9 also waits for the previous row's global STORE (rows serialised on HBM
    // write latency)
    unsigned rc[8 * PYR_RGPL];
    {
#pragma unroll
        for (int k = 0; k < 2 * PYR_RGPL; ++k) {
            const int rv4[4] = {rq[k].x, rq[k].y, rq[k].z, rq[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rr = 4 * k + e;
                rc[rr] = (__umul24((unsigned)((rv4[e] & 0xFFFF) - sy0), PYR_DW) + (unsigned)(bcol >> 2)) | ((unsigned)rv4[e] & 0xFFFF0000u);
                asm volatile("" : "+v"(rc[rr]));
            }
        }
    }
    __syncthreads();
    if (x4 >= D.pitch) return;
    // One output row = the horizontal pass of its two source rows + the vertical blend, no branches:
    //   source row: three aligned LDS dwords from the one holding p[o_0] (the 4 columns span <= 6 bytes), two v_alignbyte to
    //   the lane's byte phase; the bottom row is the top row + one window row -- where OpenCV clamps it (last source row)
    //   its weight is 0 and the window holds finite bytes, so the unclamped read is exact;
    //   column j: v_perm_b32 -> (p[o_j], p[o_j + 1]) as two u16, v_dot2_u32_u16 with (256 - a1_j, a1_j) -> h (16 bits);
    //   out_j = (b0 h_top + b1 h_bot + 2^15) >> 16 by two 24-bit mads; the four results are byte 2 of four 24-bit sums.
    // ~38 vector instructions per row of 4 pixels, and no divergent control flow; the version with a cached bottom row and
    // separate byte extraction / multiplies took 62.  (Byte-unaligned ds_read_b64 straight at p[o_0] saves the two
    // v_alignbyte and was measured 70 % SLOWER: the LDS serialises misaligned 8-byte lanes.)
    const unsigned sh = (unsigned)bcol & 3u;
    const unsigned colmask = x4 + 3 < D.w ? 0xFFFFFFFFu : (x4 >= D.w ? 0u : (0xFFFFFFFFu >> (8 * (x4 + 4 - D.w))));   // bytes past D.w stay 0
#pragma unroll
    for (int half = 0; half < PYR_RGPL; ++half) {
        const int yb = y0 + (ty8 + PYR_NRG * half) * 8;
        uint8_t *dstp = base + D.off + __umul24((unsigned)yb, (unsigned)D.pitch) + x4;
        const int nrows = min(8, D.h - yb);
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            if (r8 < nrows) {
                const unsigned rcv = rc[8 * half + r8];
                const unsigned *rt = s_src + (rcv & 0xFFFFu), *rb = rt + PYR_DW;
                const unsigned t0 = rt[0], t1 = rt[1], t2 = rt[2], u0 = rb[0], u1 = rb[1], u2 = rb[2];
                uint2 vt, vb;
                vt.x = __builtin_amdgcn_alignbyte(t1, t0, sh); vt.y = __builtin_amdgcn_alignbyte(t2, t1, sh);
                vb.x = __builtin_amdgcn_alignbyte(u1, u0, sh); vb.y = __builtin_amdgcn_alignbyte(u2, u1, sh);
                const unsigned b1u = rcv >> 16, b0u = 256u - b1u;
                unsigned sm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned ht = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16_t, __builtin_amdgcn_perm(vt.y, vt.x, selp[j])), apk[j], 0u, false);
                    const unsigned hb = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16_t, __builtin_amdgcn_perm(vb.y, vb.x, selp[j])), apk[j], 0u, false);
                    sm[j] = __umul24(b1u, hb) + (__umul24(b0u, ht) + 32768u);
                }
                const unsigned out = __builtin_amdgcn_perm(sm[1], sm[0], 0x0c0c0602u) | __builtin_amdgcn_perm(sm[3], sm[2], 0x06020c0cu);
                *(unsigned *)dstp = out & colmask;
                dstp += D.pitch;
            }
        }
    }
}

void rpe_launch_pyramid(rpe_handle *h, int n_img)
{
    for (int l = 1; l < RPE_NLEVELS; ++l) {
        const int nt = h->pyr_tile_cnt[l];
        hipLaunchKernelGGL(pyr_resize_kernel, dim3((nt + 7) / 8 * 8, n_img), dim3(PYR_THREADS), 0, h->stream, h->d_pyr, h->lay, h->d_coef,
                           (const RpePyrTile *)(h->d_pyr_tiles + h->pyr_tile_off[l]), nt, l);
    }
}

// ------------------------------------------------------------- FAST + NMS
// Fused FAST-9/16 score + 3x3 non-maximum suppression + border filter.
// Tile = 64x64 output pixels; scores are needed on 66x66, pixels on 72x72 (halo 3+1
// rows, 4 columns => dword aligned).  ~5 KB of loads in flight per workgroup.
//  phase 1: every dword group (4 px) of the 66x72 score area: OpenCV's pair test on the
//           4 compass + 4 diagonal ring pixels with class bits (darker=1 / brighter=2):
//           a 9-arc contains one pixel of every opposite pair, so
//           (c0|c8)&(c4|c12)&(c2|c10)&(c6|c14) != 0 is necessary; survivors are appended
//           to an LDS list with one wave-aggregated LDS atomic per wave.
//  phase 2: list processed densely: 16 ring differences, window-9 min/max via
//           min3/max3, score = max(A,B)-1 (0 if not a corner) into the LDS score tile.
//  phase 3: strict 3x3 maximum on the LDS score tile, 31-px border filter
//           (KeyPointsFilter::runByImageBorder), per-tile keypoint list to HBM.
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }
typedef short short2_t __attribute__((ext_vector_type(2)));
// Bresenham circle of radius 3 (fast.cpp), compile-time offsets
static constexpr int CIRC_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static constexpr int CIRC_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

#define FS_ROWS (FAST_TH + 2)        // score rows (y0-1 .. y0+FAST_TH)
#define FAST_PROWS (FAST_TH + 8)     // pixel rows (y0-4 .. y0+FAST_TH+3)
#define FAST_THREADS (FAST_TH * 4)
#define FAST_NRP (FAST_THREADS / 18)  // rows of the 18-dword tile row that one pass of the lanes covers (14 or 7)
#define FAST_LANES (FAST_NRP * 18)
// Output: one compact list per tile of the keypoints that survive NMS and the border filter, packed
// score << 24 | y << 12 | x (level coordinates).  A strict 3x3
// maximum cannot have an 8-neighbour that is one too, so a 64x64 tile holds at most 32*32 = 1024 of them:
// RPE_FAST_TILE_CAP is never exceeded and nothing is ever dropped here.  ~0.5 % of the pixels survive, so
// the lists replace a dense NMS map (1.6 MB written + re-read per VGA image) by ~100 bytes per tile; list
// order inside a tile depends on wave timing, which nothing downstream reads (select ranks by (y, x)).
__global__ __launch_bounds__(FAST_THREADS) void fast_nms_kernel(const uint8_t *__restrict__ pyr, unsigned *__restrict__ tile_list,
                                                        int *__restrict__ tile_cnt, RpeDeviceLayout lay,
                                                        const RpeTile *__restrict__ tiles, int ntiles)
{
    // 19.5 KB of LDS per workgroup = 8 workgroups (32 waves) per CU: the pixel tile is dead after phase 2, so the keypoint
    // list of phase 3 lives in its place (4096 <= 5184 bytes); at 24.5 KB only 6 workgroups fit
    constexpr int S_IN_DW = FAST_PROWS * 18 > RPE_FAST_TILE_CAP + 256 ? FAST_PROWS * 18 : RPE_FAST_TILE_CAP + 256;
    __shared__ __attribute__((aligned(16))) unsigned s_in[S_IN_DW];                // pixels  y0-4 .. y0+FAST_TH+3, x0-4 .. x0+67 (sized for the phase-3 aliases too)
    __shared__ __attribute__((aligned(16))) unsigned s_sc[FS_ROWS * 18];   // scores  y0-1 .. y0+64, x0-4 .. x0+67
    unsigned *s_out = s_in;                                               // phase 3: the tile's keypoint list [1024]
    __shared__ unsigned short s_cand[FS_ROWS * 72];
    __shared__ int s_ncand, s_nout;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ti = xcd_tile(blockIdx.x, ntiles);
    if (ti >= ntiles) return;
    const RpeTile t = tiles[ti];
    const RpeLevel &L = lay.lv[t.level];
    const int w = L.w, hgt = L.h, pitch = L.pitch, thr = lay.fast_thr;
    const int x0 = t.tx, y0 = t.ty;
    const long long tslot = (long long)blockIdx.y * ntiles + ti;
    // tiles that cannot contain a keypoint after the border filter (the host table lists none): empty list
    const bool live = w > 2 * RPE_EDGE && hgt > 2 * RPE_EDGE && x0 < w - RPE_EDGE && x0 + 64 > RPE_EDGE &&
                      y0 < hgt - RPE_EDGE && y0 + FAST_TH > RPE_EDGE;
    if (!live) {
        if (tid == 0) tile_cnt[tslot] = 0;
        return;
    }
    const uint8_t *src = rpe_level_base(pyr, lay, blockIdx.y, t.level);
    if (tid == 0) { s_ncand = 0; s_nout = 0; }
    {   // all tile loads in flight before the first LDS store.  lane -> fixed dword column (tid % 18) and rows
        // tid / 18 + 14 q: one column clamp and one division per tile instead of one per load
        const int lc = tid % 18, lr = tid / 18;
        const int lx = min(max(x0 - 4 + 4 * lc, 0), pitch - 4);
        unsigned stage[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = min(lr + FAST_NRP * q, FAST_PROWS - 1);
            const int y = min(max(y0 - 4 + r, 0), hgt - 1);
            stage[q] = *(const unsigned *)(src + (__umul24((unsigned)y, (unsigned)pitch) + (unsigned)lx));
        }
        // pin the loads above the guarded stores (the compiler sinks a load into the block of its store: dependent round trips)
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("" : "+v"(stage[q]));
#pragma unroll
        for (int q = 0; q < 6; ++q) { const int r = lr + FAST_NRP * q; if (tid < FAST_LANES && r < FAST_PROWS) s_in[r * 18 + lc] = stage[q]; }
    }
    {   // zero the score tile with 16-B LDS stores
        uint4 *z1 = (uint4 *)s_sc;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < FS_ROWS * 18 / 4; i += FAST_THREADS) z1[i] = z;
    }
    __syncthreads();
    // ---- phase 1 (packed 16-bit SWAR: even / odd pixels of the dword group in one VGPR each)
    // lane -> fixed dword column c (18 per row) and rows r0, r0+14, ... : the x-validity mask is
    // computed once per tile and the candidate bits of the 5 rows are appended in one go
    const unsigned T1 = (unsigned)(thr + 1) * 0x00010001u, T0 = (unsigned)thr * 0x00010001u;
    const int c = tid % 18, r0 = tid / 18;
    unsigned cand_bits = 0;
    if (tid < FAST_LANES) {
        const int cl = max(c - 1, 0), cr = min(c + 1, 17);
        const int pxb = x0 - 4 + 4 * c;
        const int xlo = max(x0 - 1, RPE_EDGE - 1), xhi = min(x0 + 64, w - RPE_EDGE);   // valid px range (inclusive)
        unsigned vmask = 0xFu;
        if (pxb < xlo) vmask &= 0xFu << min(xlo - pxb, 4);
        if (pxb + 3 > xhi) vmask &= 0xFu >> min(pxb + 3 - xhi, 4);
        vmask &= 0xFu;
        // candidate flags of the 5 rows share one dword: pixel px of row it -> bit 8 px + it (the four sign bytes of a row
        // are gathered by one v_perm and dropped into place by a shift and a v_and_or: 5 instructions per row instead of
        // 14 for the compact 4-bit form; the x-validity mask is applied once, spread to bytes)
        const unsigned vmask5 = ((vmask * 0x00204081u) & 0x01010101u) * 0x1Fu;
        const unsigned *pC = s_in + (r0 + 3) * 18 + c, *pL = s_in + (r0 + 3) * 18 + cl, *pR = s_in + (r0 + 3) * 18 + cr;
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int ry = r0 + FAST_NRP * it;
            const int py = y0 - 1 + ry;
            if (vmask == 0 || ry >= FS_ROWS || py < RPE_EDGE - 1 || py >= hgt - RPE_EDGE + 1) continue;
            // input row of this score row = ry + 3; the three column pointers are per-lane constants, the rows compile-time
            // offsets from them (ds_read immediate offsets instead of six address instructions per row)
            const int ro = FAST_NRP * 18 * it;
            const unsigned cdw = pC[ro], ldw = pL[ro], rdw = pR[ro];
            const unsigned upl = pL[ro - 36], upc = pC[ro - 36], upr = pR[ro - 36];
            const unsigned dnl = pL[ro + 36], dnc = pC[ro + 36], dnr = pR[ro + 36];
            const unsigned bot = pC[ro + 54], top = pC[ro - 54];
            unsigned z[2];
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                // v_perm_b32 picks bytes (sh+par, sh+par+2) of {hi:lo} into the low bytes of two 16-bit lanes
#define PK2(hi, lo, sh) __builtin_bit_cast(short2_t, __builtin_amdgcn_perm((hi), (lo), 0x0c000c00u | (unsigned)((sh) + par) | ((unsigned)((sh) + par + 2) << 16)))
                const short2_t ce = PK2(0u, cdw, 0);
                // ring values r (not differences): with d = ce - r,
                //   min_pairs max(d_a,d_b) = ce - max_pairs min(r_a,r_b),  max_pairs min(d_a,d_b) = ce - min_pairs max(r_a,r_b)
                short2_t r[8];
                r[0] = PK2(0u, bot, 0);          // ( 0, 3)
                r[1] = PK2(0u, top, 0);          // ( 0,-3)
                r[2] = PK2(rdw, cdw, 3);         // ( 3, 0)
                r[3] = PK2(cdw, ldw, 1);         // (-3, 0)
                r[4] = PK2(dnr, dnc, 2);         // ( 2, 2)
                r[5] = PK2(upc, upl, 2);         // (-2,-2)
                r[6] = PK2(upr, upc, 2);         // ( 2,-2)
                r[7] = PK2(dnc, dnl, 2);         // (-2, 2)
#undef PK2
                // opposite pairs: (0,1) (2,3) (4,5) (6,7)
                const short2_t lo = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_min(r[0], r[1]), __builtin_elementwise_min(r[2], r[3])),
                                                              __builtin_elementwise_max(__builtin_elementwise_min(r[4], r[5]), __builtin_elementwise_min(r[6], r[7])));
                const short2_t hi = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_max(r[0], r[1]), __builtin_elementwise_max(r[2], r[3])),
                                                              __builtin_elementwise_min(__builtin_elementwise_max(r[4], r[5]), __builtin_elementwise_max(r[6], r[7])));
                // darker arc possible:   ce - lo > thr  <=>  (ce - lo) - (thr+1) >= 0
                // brighter arc possible: hi - ce > thr  <=>  (ce - hi) + thr < 0
                const unsigned u = __builtin_bit_cast(unsigned, (ce - lo) - __builtin_bit_cast(short2_t, T1));
                const unsigned q = __builtin_bit_cast(unsigned, (ce - hi) + __builtin_bit_cast(short2_t, T0));
                z[par] = ~u | q;                                   // bit 15 / 31: candidate flag of the even / odd pixel pair
            }
            const unsigned x4 = __builtin_amdgcn_perm(z[1], z[0], 0x07030501u);      // bytes (px0, px1, px2, px3), flag = bit 7
            cand_bits |= (x4 >> (7 - it)) & (0x01010101u << it);
        }
        cand_bits &= vmask5;
    }
    {   // one append per tile: wave prefix sum of the per-lane counts, one LDS atomic per wave
        const int n = __popc(cand_bits);
        const int inc = wave_inclusive_sum(n);
        const int total = __shfl(inc, 63);
        if (total) {
            int base = 0;
            if (lane == 63) base = atomicAdd(&s_ncand, total);
            int pos = __shfl(base, 63) + inc - n;
            unsigned bits = cand_bits;
            while (bits) {
                const int bpos = __ffs((int)bits) - 1;
                bits &= bits - 1;
                s_cand[pos++] = (unsigned short)(((r0 + FAST_NRP * (bpos & 7)) << 7) | (4 * c + (bpos >> 3)));
            }
        }
    }
    __syncthreads();
    // ---- phase 2
    const int ncand = s_ncand;
    const uint8_t *sb = (const uint8_t *)s_in;
    for (int i = tid; i < ncand; i += FAST_THREADS) {
        const int cc = s_cand[i];
        const int bx = cc & 127, ry = cc >> 7;
        const uint8_t *p = sb + (ry + 3) * 72 + bx;
        const int v = p[0];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = v - (int)p[CIRC_DY[k] * 72 + CIRC_DX[k]];
        int m3[16], x3[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            m3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
            x3[k] = imax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        }
        int A = -1000, Bm = 1000;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            A = max(A, imin3(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]));
            Bm = min(Bm, imax3(x3[k], x3[(k + 3) & 15], x3[(k + 6) & 15]));
        }
        const int s = max(A, -Bm);
        if (s > thr) ((uint8_t *)s_sc)[ry * 72 + bx] = (uint8_t)(s - 1);
    }
    __syncthreads();
    // ---- phase 3: NMS + border filter over the candidate list (only pixels that went through
    // phase 2 can hold a score); survivors are appended to the tile's list, one LDS atomic per wave and round.
    // The pixel tile is dead now (the barrier above ended phase 2): its LDS holds the list.
    const uint8_t *sc = (const uint8_t *)s_sc;
    for (int i0 = 0; i0 < ncand; i0 += FAST_THREADS) {                  // block-uniform trip count: the ballot sees whole waves
        const int i = i0 + tid;
        bool keep = false;
        unsigned ent = 0;
        if (i < ncand) {
            const int cc = s_cand[i];
            const int bx = cc & 127, ry = cc >> 7;
            const int px = x0 - 4 + bx, py = y0 - 1 + ry;
            // branch-free: the eight neighbour reads go out together (as a short-circuit chain they were nine dependent LDS
            // round trips with an exec-mask save / restore each); reads of halo candidates past the score tile land in
            // other LDS arrays of this kernel and are discarded by `inside`
            const bool inside = (unsigned)(bx - 4) < 64u & (unsigned)(ry - 1) < (unsigned)FAST_TH &                 // halo pixels are not outputs
                                (unsigned)(px - RPE_EDGE) < (unsigned)(w - 2 * RPE_EDGE) & (unsigned)(py - RPE_EDGE) < (unsigned)(hgt - 2 * RPE_EDGE);
            const uint8_t *q = sc + ry * 72 + bx;
            const int v = q[0];
            const int nmax = imax3(imax3(q[-1], q[1], q[-73]), imax3(q[-72], q[-71], q[71]), max((int)q[72], (int)q[73]));
            if (inside & (v != 0) & (v > nmax)) {
                keep = true;
                ent = ((unsigned)v << 24) | ((unsigned)py << 12) | (unsigned)px;
            }
        }
        const unsigned long long km = __ballot(keep);
        if (km) {                                              // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nout, __popcll(km));
            base = __shfl(base, 0);
            if (keep) s_out[base + __popcll(km & ((1ull << lane) - 1ull))] = ent;
        }
    }
    __syncthreads();
    const int nout = s_nout;                                   // <= RPE_FAST_TILE_CAP by the NMS argument above
    unsigned *dst = tile_list + tslot * RPE_FAST_TILE_CAP;
    for (int i = tid; i < nout; i += FAST_THREADS) dst[i] = s_out[i];
    if (tid == 0) tile_cnt[tslot] = nout;
}

void rpe_launch_fast(rpe_handle *h, int n_img)
{
    // tiles cover the border-filtered region only
    if (h->n_tiles_fast == 0) return;
    hipLaunchKernelGGL(fast_nms_kernel, dim3((h->n_tiles_fast + 7) / 8 * 8, n_img), dim3(FAST_THREADS), 0, h->stream,
                       h->d_pyr, h->d_tile_list, h->d_tile_cnt, h->lay, h->d_tiles_fast, h->n_tiles_fast);
}

void rpe_launch_nms(rpe_handle *h, int n_img) { (void)h; (void)n_img; }   // fused into fast_nms_kernel

// ------------------------------------------------- block-wide exclusive scan
__device__ __forceinline__ int block_excl_scan(int v, int *s_wave /*[5]*/, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_inclusive_sum(v);
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { int s = s_wave[k]; if (k < wv) base += s; }
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + inc - v;
}

// ----------------------------------------------------------------- select
// cv2's keypoint ORDER (orb.cpp computeKeyPoints): FAST emits a level's corners in raster order, then
// KeyPointsFilter::retainBest(2 * quota) on the FAST score reorders them through std::nth_element + std::partition;
// the Harris responses are computed for the survivors in THAT order and retainBest(quota) reorders them once more.
// The reference's own result rows pin this order (retain_best_emul.h, tests/test_reference_rows_cpu.py), so the
// selection is replayed move for move:
//   raster_corners  one workgroup per (level, image): the level's FAST tile lists -> ONE list in raster order
//                   (counting sort over rows in LDS, rank by x inside a row), the first ccap entries
//   retain_fast     one wave per (level, image), one lane working: retainBest(2 * quota) on the FAST score -> candidates
//   harris          (below)
//   retain_harris   one wave per (level, image): retainBest(quota) on the Harris response, in place
//   compact         one workgroup per image: level-major concatenation into the keypoint arrays
// Raster order, pass by pass:
//   pass 1  per-row count of the entries (LDS atomics)                          -> exclusive scan = row starts
//   pass 2  entry -> slot row_start[y] + (arrival order inside the row); rows that start at or beyond the
//           capacity are dropped here (their ranks are >= ccap whatever their x)
//   pass 3  rank inside the row by x (a row holds a handful of entries) -> final position; positions >= ccap
//           are truncated exactly as the oracle truncates (first ccap in raster order) and flagged.
// Arrival order inside a row depends on wave timing; the final position does not.
__global__ __launch_bounds__(256) void raster_corners_kernel(const unsigned *__restrict__ tile_list, const int *__restrict__ tile_cnt,
                                                              unsigned *__restrict__ corner, int *__restrict__ corner_count,
                                                              unsigned *__restrict__ ovf, RpeDeviceLayout lay, int ntiles, int rows_cap, int key_cap)
{
    extern __shared__ unsigned s_dyn[];
    unsigned *s_rs = s_dyn;                          // [rows_cap + 1] per-row counts, then exclusive row starts
    unsigned *s_fill = s_rs + rows_cap + 1;          // [rows_cap] arrival counters
    unsigned *s_key = s_fill + rows_cap;             // [key_cap] staged entries (score << 24 | y << 12 | x), grouped by row
    unsigned *s_toff = s_key + key_cap;              // [level tiles + 1] exclusive prefix of the tile counts
    __shared__ int s_wave[5];
    __shared__ int s_live;
    const int tid = threadIdx.x, l = blockIdx.x, img = blockIdx.y;
    const RpeLevel &L = lay.lv[l];
    if (tid == 64) s_live = 0x7FFFFFFF;
    const int nrows = L.h, nt = L.ntile, ccap = L.ccap;
    for (int i = tid; i <= nrows; i += 256) s_rs[i] = 0;
    for (int i = tid; i < nrows; i += 256) s_fill[i] = 0;
    // exclusive prefix of the level's tile counts (<= 16 tiles per lane and round)
    const int *tc = tile_cnt + (long long)img * ntiles + L.tile0;
    const unsigned *tl = tile_list + ((long long)img * ntiles + L.tile0) * RPE_FAST_TILE_CAP;
    int nraw = 0;
    for (int t0 = 0; t0 < nt; t0 += 256) {                 // block-uniform
        const int t = t0 + tid;
        const int c = t < nt ? tc[t] : 0;
        int total;
        const int ex = block_excl_scan(c, s_wave, total);
        if (t < nt) s_toff[t] = (unsigned)(nraw + ex);
        nraw += total;
    }
    if (tid == 0) s_toff[nt] = (unsigned)nraw;
    __syncthreads();
    // entry s of the concatenated lists -> (tile, index) by binary search over the prefix
    auto fetch = [&](int sidx) -> unsigned {
        int lo = 0, hi = nt;                               // s_toff[lo] <= sidx < s_toff[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_toff[mid] <= (unsigned)sidx) lo = mid; else hi = mid; }
        return tl[(long long)lo * RPE_FAST_TILE_CAP + (sidx - (int)s_toff[lo])];
    };
    // both passes over the lists take four entries per lane and round: four independent loads in flight instead of one
    // dependent L2 / HBM round trip per 256 entries
    for (int base = 0; base < nraw; base += 1024) {
        unsigned e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int sidx = base + tid + 256 * u; e4[u] = sidx < nraw ? fetch(sidx) : 0u; }   // score 0 = no entry
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e4[u] >> 24) atomicAdd(&s_rs[(e4[u] >> 12) & 0xFFFu], 1u);
    }
    __syncthreads();
    // exclusive scan of the row counts, 16 consecutive rows per lane
    int kept = 0;
    for (int r0 = 0; r0 < nrows; r0 += 4096) {             // block-uniform (one round: h <= 4095)
        const int rb = r0 + tid * 16;
        unsigned c[16]; int sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { c[k] = (rb + k < nrows) ? s_rs[rb + k] : 0u; sum += (int)c[k]; }
        int total;
        int ex = block_excl_scan(sum, s_wave, total) + kept;
        int first_out = 0x7FFFFFFF;                        // first row start at or beyond the capacity
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (rb + k < nrows) { s_rs[rb + k] = (unsigned)ex; if (ex >= ccap) first_out = min(first_out, ex); }
            ex += (int)c[k];
        }
        if (first_out != 0x7FFFFFFF) atomicMin(&s_live, first_out);
        kept += total;
    }
    if (tid == 0) s_rs[nrows] = (unsigned)kept;
    __syncthreads();
    for (int base = 0; base < nraw; base += 1024) {
        unsigned e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int sidx = base + tid + 256 * u; e4[u] = sidx < nraw ? fetch(sidx) : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned e = e4[u];
            if (e >> 24) {
                const unsigned y = (e >> 12) & 0xFFFu, rs = s_rs[y];
                if (rs < (unsigned)ccap) {
                    const unsigned slot = rs + atomicAdd(&s_fill[y], 1u);
                    if (slot < (unsigned)key_cap) s_key[slot] = e;
                }
            }
        }
    }
    __syncthreads();
    unsigned *out = corner + (long long)img * lay.corner_total + L.corner_off;
    // staged slots: every row that starts below ccap is staged whole, so the live slots are [0, first row start >= ccap)
    // (or all `kept` of them); a row holds <= w/2 keypoints, hence nlive <= ccap + w/2 <= key_cap
    const int nlive = min(min(s_live, kept), key_cap);
    for (int slot = tid; slot < nlive; slot += 256) {
        const unsigned key = s_key[slot];
        const unsigned y = (key >> 12) & 0xFFFu, x = key & 0xFFFu;
        const unsigned rs = s_rs[y], re = s_rs[y + 1];
        unsigned rank = 0;
        for (unsigned j = rs; j < re; ++j) rank += ((s_key[j] & 0xFFFu) < x) ? 1u : 0u;
        const unsigned pos = rs + rank;
        if (pos < (unsigned)ccap) out[pos] = key;
    }
    if (tid == 0) {
        corner_count[img * RPE_NLEVELS + l] = min(kept, ccap);
        if (kept > ccap) atomicOr(&ovf[img], (unsigned)RPE_OVF_ORB_CANDIDATES);
    }
}

// retainBest(2 * quota) on the FAST score.  One wave per (level, image); the list sits in LDS and the wave replays the
// runtime library's nth_element + partition on it (retain_best_emul.h): libstdc++'s partition passes have a closed form
// that the 64 lanes evaluate with ballots and popcounts (one lane alone spent 2.5 ms per 2048 images chasing dependent
// LDS round trips); the few-element steps and MSVC's fat-pivot partition run on one lane.  Two launches share the work by
// list length (lo < n0 <= cap) so that the common short lists do not reserve the LDS of the longest possible one.
struct FastScoreGT { __device__ __forceinline__ bool operator()(unsigned a, unsigned b) const { return (a >> 24) > (b >> 24); } };
struct FastScoreGE { __device__ __forceinline__ bool operator()(unsigned a, unsigned b) const { return (a >> 24) >= (b >> 24); } };

__global__ __launch_bounds__(64) void retain_fast_kernel(const unsigned *__restrict__ corner, const int *__restrict__ corner_count,
                                                          unsigned *__restrict__ cand_xy, int *__restrict__ cand_count,
                                                          unsigned *__restrict__ ovf, RpeDeviceLayout lay, int lo, int cap, int n_img)
{
    extern __shared__ unsigned s_a[];                          // [cap] elements, then [cap + 2] u16 stopper positions (wave_pair_swap)
    __shared__ unsigned long long s_mask[2 * 128];           // stopper masks of <= 8192 elements
    __shared__ int s_n1;
    const int lane = threadIdx.x, l = blockIdx.x;
    const RpeLevel &L = lay.lv[l];
    // the long-list launch runs a small grid whose workgroups walk over the images: on ordinary images it has nothing to do,
    // and 10 000 empty workgroups that each reserve 26 KB of LDS took 64 us
#pragma unroll 1
    for (int img = blockIdx.y; img < n_img; img += gridDim.y) {
    __syncthreads();
    const int n0 = corner_count[img * RPE_NLEVELS + l];
    const int n_points = 2 * L.quota;
    const bool active = n0 > n_points;                       // retainBest does nothing otherwise (the list stays in raster order)
    // the pass-through lists belong to the first launch (lo == 0)
    if (active ? !(n0 > lo && n0 <= cap) : lo != 0) continue;
    const unsigned *in = corner + (long long)img * lay.corner_total + L.corner_off;
    unsigned *out = cand_xy + (long long)img * lay.cand_total + L.cand_off;
    int n1 = n0;
    if (active) {
        for (int i = lane; i < n0; i += 64) s_a[i] = in[i];
        __syncthreads();
        if (lay.stl == rb::RT_LIBSTDCXX)                     // the partition passes as ballot / popcount sweeps of the whole wave
            n1 = rb::wave_retain_best_gnu(s_a, n0, n_points, FastScoreGT(), FastScoreGE(), s_mask, (unsigned short *)(s_a + cap), &s_n1);
        else {                                               // MSVC's fat-pivot partition: one lane, move for move
            if (lane == 0) s_n1 = rb::retain_best(s_a, n0, n_points, lay.stl, FastScoreGT(), FastScoreGE());
            __syncthreads();
            n1 = s_n1;
        }
    }
    const int nw = min(n1, L.kcap2);
    for (int i = lane; i < nw; i += 64) {
        const unsigned e = active ? s_a[i] : in[i];
        out[i] = (((e >> 12) & 0xFFFu) << 16) | (e & 0xFFFu);
    }
    if (lane == 0) {
        cand_count[img * RPE_NLEVELS + l] = nw;
        if (n1 > L.kcap2) atomicOr(&ovf[img], (unsigned)RPE_OVF_ORB_CANDIDATES);
    }
    }
}

#define RPE_RETAIN_TIER 2048         // list length served by the small-LDS launch of the two retain kernels

void rpe_launch_select(rpe_handle *h, int n_img)
{
    // LDS: row starts + fill counters + staged entries (capacity: largest ccap + one full row of keypoints) + tile prefix
    int rows_cap = 0, ccap_max = 0, nt_max = 0, wmax = 0, nlev_big = 0;
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        rows_cap = std::max(rows_cap, h->lay.lv[l].h); ccap_max = std::max(ccap_max, h->lay.lv[l].ccap);
        nt_max = std::max(nt_max, h->lay.lv[l].ntile); wmax = std::max(wmax, h->lay.lv[l].w);
        if (h->lay.lv[l].ccap > RPE_RETAIN_TIER) nlev_big = l + 1;       // capacities shrink with the level
    }
    const int key_cap = ccap_max + wmax / 2 + 64;
    const size_t lds = sizeof(unsigned) * ((size_t)rows_cap + 1 + rows_cap + key_cap + nt_max + 1);
    // images beyond ~3000 px need more than the default 64 KB of dynamic LDS per workgroup (gfx950 has 160 KB per CU)
    if (lds > 65536) hipFuncSetAttribute((const void *)raster_corners_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(raster_corners_kernel, dim3(RPE_NLEVELS, n_img), dim3(256), lds, h->stream,
                       (const unsigned *)h->d_tile_list, (const int *)h->d_tile_cnt,
                       h->d_corner, h->d_corner_count, h->d_ovf, h->lay, h->n_tiles_fast, rows_cap, key_cap);
    auto lds_of = [](int cap) { return sizeof(unsigned) * (size_t)cap + sizeof(unsigned short) * ((size_t)cap + 4); };
    hipLaunchKernelGGL(retain_fast_kernel, dim3(RPE_NLEVELS, n_img), dim3(64), lds_of(std::min(ccap_max, RPE_RETAIN_TIER)), h->stream,
                       (const unsigned *)h->d_corner, (const int *)h->d_corner_count, h->d_cand_xy, h->d_cand_count, h->d_ovf, h->lay,
                       0, std::min(ccap_max, RPE_RETAIN_TIER), n_img);
    if (nlev_big > 0)
        hipLaunchKernelGGL(retain_fast_kernel, dim3(nlev_big, std::min(n_img, 512)), dim3(64), lds_of(ccap_max), h->stream,
                           (const unsigned *)h->d_corner, (const int *)h->d_corner_count, h->d_cand_xy, h->d_cand_count, h->d_ovf, h->lay,
                           RPE_RETAIN_TIER, ccap_max, n_img);
}

// ----------------------------------------------------------------- harris
// orb.cpp HarrisResponses: 7x7 block of 3x3 Sobel-like derivatives, f32 response.
__global__ __launch_bounds__(256) void harris_kernel(const uint8_t *__restrict__ pyr, const unsigned *__restrict__ cand_xy,
                                                      const int *__restrict__ cand_count, float *__restrict__ cand_resp,
                                                      RpeDeviceLayout lay, int nb, int n_img)
{
    // dense lane -> candidate mapping over the per-level counts: the slot arrays are sized 4*quota + 256 per level but
    // hold ~2*quota entries, so slot-indexed lanes were two thirds idle
    int img, blk;
    if (!xcd_image_block(nb, n_img, img, blk)) return;
    const int dsel = blk * 256 + threadIdx.x;
    int l = -1, ci = 0, acc = 0;
#pragma unroll
    for (int k = 0; k < RPE_NLEVELS; ++k) {
        const int n = cand_count[img * RPE_NLEVELS + k];
        if (l < 0 && dsel < acc + n) { l = k; ci = dsel - acc; }
        acc += n;
    }
    if (l < 0) return;
    const RpeLevel &L = lay.lv[l];
    const int c = L.cand_off + ci;
    unsigned xy = cand_xy[(long long)img * lay.cand_total + c];
    const int x0 = xy & 0xFFFF, y0 = xy >> 16, pitch = L.pitch;
    // 9x9 patch (7x7 block of 3x3 derivatives) as 9 rows of THREE ALIGNED DWORDS each (the 9 bytes x0-4 .. x0+4 start at
    // byte (x0 & 3) of the dword run beginning at (x0 - 4) & ~3): 27 dword loads per candidate, all in flight at once,
    // instead of 81 dependent byte loads -- the kernel was bound by the number of memory instructions, not by bytes
    // (0.05 G vector instructions in 0.5 ms).  Reads stay inside the row: x0 >= 31 and x0 + 7 < w - 24 <= pitch.
    const int sh = x0 & 3;
    const uint8_t *p0 = rpe_level_base(pyr, lay, img, l) + (long long)(y0 - 4) * pitch + ((x0 - 4) & ~3);
    unsigned w[9][3];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const unsigned *pr = (const unsigned *)(p0 + r * pitch);
        w[r][0] = pr[0]; w[r][1] = pr[1]; w[r][2] = pr[2];
    }
    int a = 0, b = 0, cc = 0;
    int rowm[9], row0[9], rowp[9];
    auto unpack = [&](int r, int (&out)[9]) {
        const unsigned q0 = __builtin_amdgcn_alignbyte(w[r][1], w[r][0], sh), q1 = __builtin_amdgcn_alignbyte(w[r][2], w[r][1], sh);
        const unsigned q2 = w[r][2] >> (8 * sh);
#pragma unroll
        for (int k = 0; k < 4; ++k) { out[k] = (int)((q0 >> (8 * k)) & 255u); out[4 + k] = (int)((q1 >> (8 * k)) & 255u); }
        out[8] = (int)(q2 & 255u);
    };
    unpack(0, rowm); unpack(1, row0);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        unpack(r + 2, rowp);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            int Ix = (row0[k + 1] - row0[k - 1]) * 2 + (rowm[k + 1] - rowm[k - 1]) + (rowp[k + 1] - rowp[k - 1]);
            int Iy = (rowp[k] - rowm[k]) * 2 + (rowp[k - 1] - rowm[k - 1]) + (rowp[k + 1] - rowm[k + 1]);
            a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) { rowm[k] = row0[k]; row0[k] = rowp[k]; }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    float fa = (float)a, fb = (float)b, fc = (float)cc;
    float t1 = fa * fb, t2 = fc * fc, t3 = t1 - t2;
    float s = fa + fb;
    float t4 = (0.04f * s) * s;
    cand_resp[(long long)img * lay.cand_total + c] = (t3 - t4) * scale_sq_sq;
}

void rpe_launch_harris(rpe_handle *h, int n_img)
{
    const int nb = (h->lay.cand_total + 255) / 256;
    hipLaunchKernelGGL(harris_kernel, dim3(xcd_image_grid(nb, n_img)), dim3(256), 0, h->stream,
                       h->d_pyr, h->d_cand_xy, h->d_cand_count, h->d_cand_resp, h->lay, nb, n_img);
}

// -------------------------------------------------------------- keypoints
// KeyPointsFilter::retainBest(quota) on the Harris response, per level, replayed on the candidates in the order the
// first retainBest left them (see "select" above): element = (f32 response bits << 32 | y << 16 | x), one lane, in LDS;
// the survivors go back to the head of the level's candidate run, in order.  compact_keypoints then concatenates the
// levels into the level-major keypoint arrays the descriptor kernel and the matcher read.
struct HarrisGT { __device__ __forceinline__ bool operator()(unsigned long long a, unsigned long long b) const { return __uint_as_float((unsigned)(a >> 32)) > __uint_as_float((unsigned)(b >> 32)); } };
struct HarrisGE { __device__ __forceinline__ bool operator()(unsigned long long a, unsigned long long b) const { return __uint_as_float((unsigned)(a >> 32)) >= __uint_as_float((unsigned)(b >> 32)); } };

__global__ __launch_bounds__(64) void retain_harris_kernel(unsigned *__restrict__ cand_xy, float *__restrict__ cand_resp,
                                                            const int *__restrict__ cand_count, int *__restrict__ kp_lvl_count,
                                                            RpeDeviceLayout lay, int lo, int cap)
{
    extern __shared__ unsigned long long s_e[];               // [cap] elements, then [cap + 2] u16 stopper positions
    __shared__ unsigned long long s_mask[2 * 128];
    __shared__ int s_n2;
    const int lane = threadIdx.x, l = blockIdx.x, img = blockIdx.y;
    const RpeLevel &L = lay.lv[l];
    const int n1 = cand_count[img * RPE_NLEVELS + l];
    const int q = L.quota;
    const bool active = n1 > q;
    if (active ? !(n1 > lo && n1 <= cap) : lo != 0) return;
    if (!active) { if (lane == 0) kp_lvl_count[img * RPE_NLEVELS + l] = n1; return; }
    unsigned *xy = cand_xy + (long long)img * lay.cand_total + L.cand_off;
    float *resp = cand_resp + (long long)img * lay.cand_total + L.cand_off;
    for (int i = lane; i < n1; i += 64) s_e[i] = ((unsigned long long)__float_as_uint(resp[i]) << 32) | xy[i];
    __syncthreads();
    int n2;
    if (lay.stl == rb::RT_LIBSTDCXX)
        n2 = rb::wave_retain_best_gnu(s_e, n1, q, HarrisGT(), HarrisGE(), s_mask, (unsigned short *)(s_e + cap), &s_n2);
    else {
        if (lane == 0) s_n2 = rb::retain_best(s_e, n1, q, lay.stl, HarrisGT(), HarrisGE());
        __syncthreads();
        n2 = s_n2;
    }
    for (int i = lane; i < n2; i += 64) { const unsigned long long e = s_e[i]; xy[i] = (unsigned)e; resp[i] = __uint_as_float((unsigned)(e >> 32)); }
    if (lane == 0) kp_lvl_count[img * RPE_NLEVELS + l] = n2;
}

__global__ __launch_bounds__(256) void compact_keypoints_kernel(const unsigned *__restrict__ cand_xy, const float *__restrict__ cand_resp,
                                                                 const int *__restrict__ kp_lvl_count,
                                                                 unsigned *__restrict__ kp_xy, float *__restrict__ kp_resp,
                                                                 float2 *__restrict__ kp_pt, int *__restrict__ kp_count,
                                                                 unsigned *__restrict__ ovf, RpeDeviceLayout lay)
{
    const int img = blockIdx.x, kcap = lay.kcap;
    int offset = 0;
#pragma unroll 1
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        const RpeLevel &L = lay.lv[l];
        const int n = kp_lvl_count[img * RPE_NLEVELS + l];
        const unsigned *xy = cand_xy + (long long)img * lay.cand_total + L.cand_off;
        const float *resp = cand_resp + (long long)img * lay.cand_total + L.cand_off;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int o = offset + i;
            if (o < kcap) {
                const unsigned p = xy[i];
                const int x = p & 0xFFFF, y = p >> 16;
                const long long g = (long long)img * kcap + o;
                kp_xy[g] = (unsigned)x | ((unsigned)y << 12) | ((unsigned)l << 24);
                kp_resp[g] = resp[i];
                kp_pt[g] = make_float2((float)x * L.scale, (float)y * L.scale);
            }
        }
        offset += n;
    }
    if (threadIdx.x == 0) {
        kp_count[img] = min(offset, kcap);
        if (offset > kcap) atomicOr(&ovf[img], (unsigned)RPE_OVF_ORB_KEYPOINTS);
    }
}

void rpe_launch_keypoints(rpe_handle *h, int n_img)
{
    int kcap2_max = 0, nlev_big = 0;
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        kcap2_max = std::max(kcap2_max, h->lay.lv[l].kcap2);
        if (h->lay.lv[l].kcap2 > RPE_RETAIN_TIER / 2) nlev_big = l + 1;
    }
    const int tier = std::min(kcap2_max, RPE_RETAIN_TIER / 2);
    auto lds_of = [](int cap) { return sizeof(unsigned long long) * (size_t)cap + sizeof(unsigned short) * ((size_t)cap + 4); };
    hipLaunchKernelGGL(retain_harris_kernel, dim3(RPE_NLEVELS, n_img), dim3(64), lds_of(tier), h->stream,
                       h->d_cand_xy, h->d_cand_resp, (const int *)h->d_cand_count, h->d_kp_lvl_count, h->lay, 0, tier);
    if (nlev_big > 0)
        hipLaunchKernelGGL(retain_harris_kernel, dim3(nlev_big, n_img), dim3(64), lds_of(kcap2_max), h->stream,
                           h->d_cand_xy, h->d_cand_resp, (const int *)h->d_cand_count, h->d_kp_lvl_count, h->lay, RPE_RETAIN_TIER / 2, kcap2_max);
    hipLaunchKernelGGL(compact_keypoints_kernel, dim3(n_img), dim3(256), 0, h->stream,
                       (const unsigned *)h->d_cand_xy, (const float *)h->d_cand_resp, (const int *)h->d_kp_lvl_count,
                       h->d_kp_xy, h->d_kp_resp, h->d_kp_pt, h->d_kp_count, h->d_ovf, h->lay);
}

// ------------------------------------------------- orientation + descriptor
// Fused per-keypoint kernel (replaces the separate ICAngles, whole-pyramid GaussianBlur
// and rBRIEF launches of the first version): one wave per keypoint, KP_PER_WG (= 1) keypoints per
// workgroup.  The 45 x 48-byte raw patch around the keypoint is staged in LDS once and
// feeds (1) the intensity-centroid moments over the radius-15 disc -> fastAtan2 angle,
// (2) the horizontal pass of the fixed-point 7x7 Gaussian on the rows the descriptor can
// touch, (3) the vertical pass evaluated only at the 512 steered sampling points.
// Integer results are identical to blurring the whole level (the patch never reaches
// the image border: keypoints are >= 31 px inside, the footprint is 22 px).
// keypoints (waves) per workgroup.  The phases of a keypoint (patch fetch 0.88 ms, moments + MFMA blur 0.36 ms, steered
// sampling 0.65 ms when run alone -- diagnostic builds) overlap only through OTHER waves, and barriers that tie four
// keypoints together cost more than the shared angle / sincos evaluation saves: 4 per workgroup 2.25 ms, 2: 2.25, 1: 2.17.
#ifndef KP_PER_WG
#define KP_PER_WG 1
#endif
#define KP_R 22
#define KP_ROWS 45
#define KP_RAW_DW 12                 // 48 bytes per raw row
#define KP_HCOLS 40                  // horizontally blurred columns: x = x0 - 19 + j (the descriptor reaches |dx| <= 18: columns 1..37)
#define KP_HSTRIDE 49                // f32 per column of the blurred buffer (45 rows + padding; odd, so neighbouring columns start in different banks)
__global__ __launch_bounds__(64 * KP_PER_WG) void orient_describe_kernel(const uint8_t *__restrict__ pyr, const unsigned *__restrict__ kp_xy,
                                                               const float2 *__restrict__ kp_pt, const int *__restrict__ kp_count,
                                                               float *__restrict__ kp_angle, uint8_t *__restrict__ desc,
                                                               RpeDeviceLayout lay, int nb, int n_img)
{
    __shared__ __attribute__((aligned(16))) unsigned s_raw[KP_PER_WG][KP_ROWS * KP_RAW_DW + 4];   // 16-B aligned rows of 48 B (+ one row pass over-read)
    // horizontally blurred patch, COLUMN-major f32 [column][row]: the 7 vertical taps of a steered sample are contiguous
    __shared__ float s_hb[KP_PER_WG][KP_HCOLS * KP_HSTRIDE];
#if KP_PER_WG > 1
    __shared__ int s_m[KP_PER_WG][2];              // (m01, m10) of the workgroup's keypoints
    __shared__ float s_ab[KP_PER_WG][2];           // (cos, sin) of their angles
#endif
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int img, blk;
    if (!xcd_image_block(nb, n_img, img, blk)) return;
    const int k = blk * KP_PER_WG + wv;
    const int nkp = kp_count[img];
    if (blk * KP_PER_WG >= nkp) return;                               // workgroup-uniform: the grid is sized for the keypoint capacity
    const bool active = k < nkp;
    const long long g = (long long)img * lay.kcap + (active ? k : 0);
    const unsigned p = kp_xy[g];
    const int x0 = p & 0xFFF, y0 = (p >> 12) & 0xFFF, l = p >> 24;
    const RpeLevel &L = lay.lv[l];
    const int pitch = L.pitch;
    const int xal = (x0 - KP_R) & ~3, off0 = (x0 - KP_R) - xal;     // off0 in 0..3
    unsigned *raw = s_raw[wv];
    float *hb = s_hb[wv];
    if (active) {
        const uint8_t *src = rpe_level_base(pyr, lay, img, l) + (long long)(y0 - KP_R) * pitch + xal;
        // lane -> fixed dword column (lane % 12) and rows lane / 12 + 5 q (60 lanes x 9 loads = 45 x 12 dwords)
        const int lc = lane % KP_RAW_DW, lr = lane / KP_RAW_DW;
        unsigned stage[9];
        const uint8_t *col = src + 4 * lc;
#pragma unroll
        for (int q = 0; q < 9; ++q) stage[q] = *(const unsigned *)(col + (long long)min(lr + 5 * q, KP_ROWS - 1) * pitch);
        if (lane < 60) {
#pragma unroll
            for (int q = 0; q < 9; ++q) raw[(lr + 5 * q) * KP_RAW_DW + lc] = stage[q];
        }
    }
    __syncthreads();
    int km01 = 0, km10 = 0;
    if (active) {
        // ---- orb.cpp ICAngles: integer moments over the disc, reduced with wave shuffles
        // integer sums, so any summation order gives the oracle's moments: 4 disc pixels per packed-u8 dot product
        int m10 = 0, m01 = 0;
        {
            const unsigned *rw = raw + (KP_R - 15) * KP_RAW_DW;                  // patch row of v = -15
            const int bo = off0 + KP_R - 16;                                        // byte column of u = -16 (>= 6)
            const int sh = bo & 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int it = lane + 64 * q;                                       // item = (v + 15) * 8 + j
                if (it < 31 * 8) {
                    const int vr = it >> 3, j = it & 7;
                    const uint2 wt = c_discw[it];
                    const unsigned *pw = rw + vr * KP_RAW_DW + ((bo >> 2) + j);
                    const unsigned px = __builtin_amdgcn_alignbyte(pw[1], pw[0], sh);
                    const int s1 = (int)__builtin_amdgcn_udot4(px, wt.x, 0u, false);
                    const int su = (int)__builtin_amdgcn_udot4(px, wt.y, 0u, false);
                    m10 += su - 16 * s1;
                    m01 += (vr - 15) * s1;
                }
            }
        }
        // wave sums by DPP row shifts / broadcasts (total in lane 63, read back as a scalar): 12 + 2 instructions; six
        // rounds of __shfl_xor were 36 vector + 12 LDS (ds_bpermute) instructions
        m10 = wave_sum(m10);
        m01 = wave_sum(m01);
        // fastAtan2 and the deterministic sincos are ~100 vector instructions on wave-uniform values: the four keypoints
        // of the workgroup get them from four LANES of wave 0 after the barrier below instead of from four waves
#if KP_PER_WG > 1
        if (lane == 0) { s_m[wv][0] = m01; s_m[wv][1] = m10; }
#else
        km01 = m01; km10 = m10;
#endif
        // ---- horizontal pass of the descriptor blur.  cv2 blurs every pyramid level with GaussianBlur(7x7, sigma 2) before
        // computeOrbDescriptors; on a pyramid SUB-matrix that call takes sepFilter2D's f32 route (filter.simd.hpp
        // RowFilter<uchar, float>, SymmColumnFilter<Cast<float, uchar>>), whose AVX2-dispatched build fuses s += f * x: the
        // reference's own result rows single this out against every fixed-point variant (tests/test_reference_rows_cpu.py).
        // Row pass: s = g0 p[x-3]; s = fma(g_k, p[x-3+k], s), k = 1..6, in that order.  hbuf column j <-> x = x0 - 19 + j.
        // Item = (patch row, group of 8 columns): 45 x 5 items, 20 raw bytes each, shared by the item's 8 outputs.
        {
            const float g0 = c_gauss[0], g1 = c_gauss[1], g2 = c_gauss[2], g3 = c_gauss[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int it = lane + 64 * q;
                if (it < KP_ROWS * 5) {
                    const int row = it / 5, cg = it - 5 * row;
                    const unsigned *pw = raw + row * KP_RAW_DW + 2 * cg;
                    const unsigned d0 = pw[0], d1 = pw[1], d2 = pw[2], d3 = pw[3], d4 = pw[4];
                    const unsigned q0 = __builtin_amdgcn_alignbyte(d1, d0, off0), q1 = __builtin_amdgcn_alignbyte(d2, d1, off0),
                                   q2 = __builtin_amdgcn_alignbyte(d3, d2, off0), q3 = __builtin_amdgcn_alignbyte(d4, d3, off0);
                    // two outputs per v_pk_mul_f32 / v_pk_fma_f32 (each half an IEEE fma of its own: same bits as the scalar
                    // form).  A packed operand is an aligned register pair, so the converted pixels are kept twice: pairs
                    // starting at even bytes (fe) and at odd bytes (fo); output pair (o, o + 1), tap k reads pair o + k.
                    typedef float v2f_t __attribute__((ext_vector_type(2)));
                    const unsigned qq[4] = {q0, q1, q2, q3};
                    v2f_t fe[8], fo[7];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned w = qq[e >> 1] >> (16 * (e & 1));
                        fe[e] = (v2f_t){(float)(w & 255u), (float)((w >> 8) & 255u)};
                    }
#pragma unroll
                    for (int e = 0; e < 7; ++e) fo[e] = (v2f_t){fe[e].y, fe[e + 1].x};
                    float *dst = hb + (8 * cg) * KP_HSTRIDE + row;
                    const v2f_t G0 = {g0, g0}, G1 = {g1, g1}, G2 = {g2, g2}, G3 = {g3, g3};
#pragma unroll
                    for (int o = 0; o < 8; o += 2) {
                        // pair j = o + k: even j -> fe[j / 2], odd j -> fo[(j - 1) / 2]
                        auto P = [&](int j) -> v2f_t { return (j & 1) ? fo[(j - 1) >> 1] : fe[j >> 1]; };
                        v2f_t sacc = G0 * P(o);
                        sacc = __builtin_elementwise_fma(G1, P(o + 1), sacc);
                        sacc = __builtin_elementwise_fma(G2, P(o + 2), sacc);
                        sacc = __builtin_elementwise_fma(G3, P(o + 3), sacc);
                        sacc = __builtin_elementwise_fma(G2, P(o + 4), sacc);      // the kernel is symmetric: g4 = g2, g5 = g1, g6 = g0 (same f32 values)
                        sacc = __builtin_elementwise_fma(G1, P(o + 5), sacc);
                        sacc = __builtin_elementwise_fma(G0, P(o + 6), sacc);
                        dst[o * KP_HSTRIDE] = sacc.x;
                        dst[(o + 1) * KP_HSTRIDE] = sacc.y;
                    }
                }
            }
        }
    }
    __syncthreads();
#if KP_PER_WG > 1
    if (wv == 0 && lane < KP_PER_WG && blk * KP_PER_WG + lane < nkp) {
        const float angle = fast_atan2_deg((float)s_m[lane][0], (float)s_m[lane][1]);
        kp_angle[(long long)img * lay.kcap + blk * KP_PER_WG + lane] = angle;
        const float ang = angle * (float)(3.141592653589793238462643383279502884 / 180.0);
        double sn, cs;
        det_sincos((double)ang, sn, cs);
        s_ab[lane][0] = (float)cs; s_ab[lane][1] = (float)sn;
    }
    __syncthreads();
    if (!active) return;
    const float a = s_ab[wv][0], b = s_ab[wv][1];
#else
    // one keypoint per workgroup: the wave evaluates its own angle (uniform values), no exchange and no barrier
    if (!active) return;
    float a, b;
    {
        const float angle = fast_atan2_deg((float)km01, (float)km10);
        if (lane == 0) kp_angle[g] = angle;
        const float ang = angle * (float)(3.141592653589793238462643383279502884 / 180.0);
        double sn, cs;
        det_sincos((double)ang, sn, cs);
        a = (float)cs; b = (float)sn;
    }
#endif
    // ---- orb.cpp computeOrbDescriptors: lane = 4 consecutive bit tests, vertical pass at the samples
    const float2 pt = kp_pt[g];
    const float sc = 1.f / L.scale;
    const int cx = __float2int_rn(pt.x * sc), cy = __float2int_rn(pt.y * sc);
    const int dxo = cx - x0 + 19, dyo = cy - y0 + KP_R - 3;       // (cx,cy) == (x0,y0) in practice
    const float g3 = c_gauss[3], g4 = c_gauss[4], g5 = c_gauss[5], g6 = c_gauss[6];
    unsigned nib = 0;
#pragma unroll
    for (int bit = 0; bit < 4; ++bit) {
        const float4 pf = c_pattern_f[lane * 4 + bit];
        const float p0 = pf.x, p1 = pf.y, p2 = pf.z, p3 = pf.w;
        float fx0 = p0 * a - p1 * b, fy0 = p0 * b + p1 * a;
        float fx1 = p2 * a - p3 * b, fy1 = p2 * b + p3 * a;
        float t01[2];
        const int ixs[2] = {__float2int_rn(fx0), __float2int_rn(fx1)}, iys[2] = {__float2int_rn(fy0), __float2int_rn(fy1)};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // vertical pass at the sample only: the 7 taps = rows iy + dyo .. + 6 of column ix + dxo, contiguous f32;
            // s = g3 c; s = fma(g_{3+k}, t[3+k] + t[3-k], s), k = 1..3 (SymmColumnFilter), then cvRound (saturate_cast<uchar>)
            const float *t = hb + (ixs[e] + dxo) * KP_HSTRIDE + (iys[e] + dyo);
            float sacc = g3 * t[3];
            sacc = __builtin_fmaf(g4, t[4] + t[2], sacc);
            sacc = __builtin_fmaf(g5, t[5] + t[1], sacc);
            sacc = __builtin_fmaf(g6, t[6] + t[0], sacc);
            t01[e] = __builtin_rintf(sacc);                           // 0 <= value <= 255 (convex combination of bytes): no saturation to apply
        }
        nib |= (unsigned)(t01[0] < t01[1]) << bit;
    }
    // 8 nibbles -> one dword (lanes 8m .. 8m+7), written by lane 8m
    unsigned v = nib << (4 * (lane & 7));
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);       // quad_perm [1,0,3,2]: lane ^ 1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);       // quad_perm [2,3,0,1]: lane ^ 2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);      // row_half_mirror: the other quad of the 8
    if ((lane & 7) == 0) *(unsigned *)(desc + g * 32 + (lane >> 3) * 4) = v;
}

void rpe_launch_angle(rpe_handle *h, int n_img)
{
    const int nb = (h->lay.kcap + KP_PER_WG - 1) / KP_PER_WG;
    hipLaunchKernelGGL(orient_describe_kernel, dim3(xcd_image_grid(nb, n_img)), dim3(64 * KP_PER_WG), 0, h->stream,
                       h->d_pyr, h->d_kp_xy, h->d_kp_pt, h->d_kp_count, h->d_kp_angle, h->d_desc, h->lay, nb, n_img);
}

// ------------------------------------------------------------------- blur
// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) as ORB runs it (in place on a pyramid sub-matrix => cv2's
// sepFilter2D f32 route, see orient_describe_kernel): row pass s = g0 p[x-3], s = fma(g_k, p[x-3+k], s); column pass
// s = g3 c, s = fma(g_{3+k}, r[y+k] + r[y-k], s); result cvRound(s).  Whole levels, for rpe_orb_debug_fetch only.
__device__ __forceinline__ int refl101(int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * n - 2 - p : p; }

__global__ __launch_bounds__(256) void blur_kernel(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ dst,
                                                    RpeDeviceLayout lay, const RpeTile *__restrict__ tiles, int src_img)
{
    // 64x64 tile; input 70 rows x 72 bytes (x0-4 .. x0+67) loaded as aligned dwords; rows
    // are reflected at load time, the <=3 reflected columns per side are patched in LDS.
    __shared__ unsigned s_in[(TH + 6) * 18];
    __shared__ float4 s_h[(TH + 6) * 16];          // horizontal pass: 4 x f32 per entry
    const int tid = threadIdx.x;
    const RpeTile t = tiles[blockIdx.x];
    const RpeLevel &L = lay.lv[t.level];
    const int w = L.w, hgt = L.h, pitch = L.pitch;
    const int x0 = t.tx, y0 = t.ty;
    const long long ibase = (long long)blockIdx.y * lay.stride + L.off;                 // destination: image slot blockIdx.y
    const uint8_t *src = pyr + (long long)(src_img + blockIdx.y) * lay.stride + L.off;
    for (int i = tid; i < (TH + 6) * 18; i += 256) {
        int r = i / 18, c = i - r * 18;
        int y = refl101(y0 - 3 + r, hgt);
        y = min(max(y, 0), hgt - 1);
        int x = min(max(x0 - 4 + 4 * c, 0), pitch - 4);
        s_in[i] = *(const unsigned *)(src + (long long)y * pitch + x);
    }
    __syncthreads();
    uint8_t *sb = (uint8_t *)s_in;
    if (x0 == 0) {               // columns -1,-2,-3 <- 1,2,3
        for (int i = tid; i < (TH + 6) * 3; i += 256) { int r = i / 3, k = i - r * 3 + 1; sb[r * 72 + 4 - k] = sb[r * 72 + 4 + k]; }
    }
    if (x0 + 64 + 3 >= w && x0 < w) {   // columns w, w+1, w+2 <- w-2, w-3, w-4
        for (int i = tid; i < (TH + 6) * 3; i += 256) {
            int r = i / 3, k = i - r * 3;
            int cd = w + k - x0 + 4, cs = w - 2 - k - x0 + 4;
            if (cd < 72 && cs >= 0) sb[r * 72 + cd] = sb[r * 72 + cs];
        }
    }
    __syncthreads();
    const float g0 = c_gauss[0], g1 = c_gauss[1], g2 = c_gauss[2], g3 = c_gauss[3];
    for (int i = tid; i < (TH + 6) * 16; i += 256) {
        int r = i >> 4, c = i & 15;
        unsigned a = s_in[r * 18 + c], b = s_in[r * 18 + c + 1], d = s_in[r * 18 + c + 2];
        float p[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { p[k] = (float)((a >> (8 * k)) & 255u); p[4 + k] = (float)((b >> (8 * k)) & 255u); p[8 + k] = (float)((d >> (8 * k)) & 255u); }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sacc = g0 * p[j + 1];
            sacc = __builtin_fmaf(g1, p[j + 2], sacc); sacc = __builtin_fmaf(g2, p[j + 3], sacc); sacc = __builtin_fmaf(g3, p[j + 4], sacc);
            sacc = __builtin_fmaf(g2, p[j + 5], sacc); sacc = __builtin_fmaf(g1, p[j + 6], sacc); sacc = __builtin_fmaf(g0, p[j + 7], sacc);
            o[j] = sacc;
        }
        s_h[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    const int tx = tid & 15, tyb = tid >> 4;
#pragma unroll
    for (int rr = 0; rr < TH / 16; ++rr) {
        const int ty = tyb + 16 * rr;
        float4 r[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) r[k] = s_h[(ty + k) * 16 + tx];
        auto col = [&](float c, float a1, float b1, float a2, float b2, float a3, float b3) -> unsigned {
            float sacc = g3 * c;
            sacc = __builtin_fmaf(g2, a1 + b1, sacc);
            sacc = __builtin_fmaf(g1, a2 + b2, sacc);
            sacc = __builtin_fmaf(g0, a3 + b3, sacc);
            return (unsigned)min(max(__float2int_rn(sacc), 0), 255);
        };
        unsigned out = col(r[3].x, r[4].x, r[2].x, r[5].x, r[1].x, r[6].x, r[0].x) |
                       (col(r[3].y, r[4].y, r[2].y, r[5].y, r[1].y, r[6].y, r[0].y) << 8) |
                       (col(r[3].z, r[4].z, r[2].z, r[5].z, r[1].z, r[6].z, r[0].z) << 16) |
                       (col(r[3].w, r[4].w, r[2].w, r[5].w, r[1].w, r[6].w, r[0].w) << 24);
        int px = x0 + 4 * tx, py = y0 + ty;
        if (py < hgt && px < pitch) *(unsigned *)(dst + ibase + (long long)py * pitch + px) = out;
    }
}

// whole-level blur of ONE image of the last run into the one-image debug buffer (rpe_orb_debug_fetch which = 3)
void rpe_launch_blur(rpe_handle *h, int img)
{
    hipLaunchKernelGGL(blur_kernel, dim3(h->n_tiles_full, 1), dim3(256), 0, h->stream,
                       h->d_pyr, h->d_bufA, h->lay, h->d_tiles_full, img);
}

// --------------------------------------------------------------- describe
void rpe_launch_describe(rpe_handle *h, int n_img) { (void)h; (void)n_img; }   // fused into orient_describe_kernel
