// sift_kernels.hip -- SIFT detectAndCompute on gfx950 (BASELINE config 3).
//
// Replaces cv2.SIFT_create().detectAndCompute(image, None) (reference
// src/core/pose_estimator.py:93-94, :108).  Algorithm = OpenCV 4.x sift.dispatch.cpp /
// sift.simd.hpp as restated in oracle/sift_oracle.c; every f32 operation keeps the oracle's
// order (contraction off, deterministic exp/sincos, fixed partial-sum trees), so results
// compare bit for bit.  Kernel groups (all images of the batch per launch):
//   upsample  : u8 -> f32, 2x INTER_LINEAR in exact integer arithmetic (weights 1/4, 3/4), 4 x 2 outputs per lane
//   blur      : fused separable Gaussian per 64 x 32 tile (window + row-pass plane in LDS, packed-f32 taps read as
//               operand pairs by two-offset LDS reads, 16-byte stores), reflect-101; an unfused row/column pair
//               remains as the fallback for other tap counts
//   halve     : INTER_NEAREST octave decimation
//   (DoG)     : never stored: layer l = G[l+1] - G[l] is formed where it is consumed (extrema scan, adjust)
//   extrema   : tiled 26-neighbour test (rolling LDS layers with halo columns, 3x3x3 max/min from LDS reads) -> 1-bit hit
//               mask + one counter per (octave, layer, row) band; parallel band scan; wave-per-row emit = raster-ordered seeds
//   adjust    : one lane per seed: adjustLocalExtrema (contrast / edge tests); survivors appended by integer atomics
//   select    : under a cap only the nfeatures * 5/4 + 256 strongest survivors (radix select on |contrast|) get an orientation;
//               exact because retainBest's threshold lies inside them (a second round covers the case that it does not)
//   orient    : one wave per selected survivor: 36-bin orientation histogram in 8 interleaved partials per bin (per-slot order
//               kept in registers by DPP row shifts), Gaussian weights from a per-keypoint table, peaks -> raw keypoints
//   sort      : response prefilter, bitonic sort of the raw keypoints by KeyPoint_LessThan (one workgroup / image)
//   finalize  : duplicate removal, retainBest(nfeatures) by radix select, ordered compaction
//   describe  : one wave per keypoint: the sample positions kept in a row of the square are an interval (one lane per row finds
//               it), dense batches of 64 samples, 4x4x8 trilinear histogram in 8 interleaved partials, lane = (slot, corner)
//   march     : (RPE_SIFT_MARCH=1) levels 1-3 / 4-5 of the large octaves in one pass over the source level each, rings of
//               row-filtered rows in LDS; bit-identical to the tile kernels
// Kernels whose lanes append through a returning atomic on a per-image counter (adjust, orient) put the image index on
// the fastest grid axis, so that workgroups in flight together hit different counters.
#include "rpe_internal.h"
#include "rpe_devmath.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <utility>

// lanes of one wave exchanging data through LDS: the hardware runs a wave's LDS instructions in order, so all that is
// needed is that the compiler keeps them in program order and does not move them across the point
#define S_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define S_NOL 3
#define S_NG 6
#define S_ND 5
#define S_BORDER 5
#define S_BINS 36

__constant__ float c_skern[6][32];     // [0] initial blur, [1..5] incremental octave blurs
__constant__ int c_sks[6];

struct SiftXTile { int o, x0, y0; };     // SX_TW x SX_TH tile of the extrema scan

struct SiftDev {                        // passed by value to kernels
    int noct;
    int w[12], h[12];
    long long goff[12], doff[12];       // float offsets inside the per-image gaussian / DoG buffers
    long long gstride, dstride, tstride; // floats per image
    int seed_cap, raw_cap, kcap, nfeatures, nbands;
    // extrema scan: a band = one row of one (octave, layer 1..3): index band0[o] + (l-1)*(h-10) + (r-5), i.e. the
    // oracle's enumeration order (octave, layer, row); hits live in a 1-bit-per-pixel mask [3][h][wpr] of 64-bit words
    int band0[12], wpr[12];
    long long bmoff[12], bmstride;
};

struct RpeSiftState {
    SiftDev dv;
    float *d_gauss = nullptr, *d_dog = nullptr, *d_tmp = nullptr;
    SiftXTile *d_xtiles = nullptr; int n_xtiles = 0;
    unsigned long long *d_xmask = nullptr;                 // [img][bmstride]
    int *d_band_cnt = nullptr, *d_band_off = nullptr;     // [img][nbands]
    unsigned *d_seeds = nullptr; int *d_nseeds = nullptr; // [img][seed_cap], [img]
    float *d_raw = nullptr;                                // [img][raw_cap][6]: x y size angle response octave(bits)
    int *d_nraw = nullptr, *d_overflow = nullptr, *d_ncand = nullptr;
    float *d_surv = nullptr; int *d_nsurv = nullptr;        // [img][seed_cap][SURV_W] refined seeds, [img]
    unsigned *d_sel = nullptr; int *d_nsel = nullptr;       // [img][seed_cap] survivors that get an orientation (sift_select_kernel), [img][4] = {count, cut, redo, -}
    int sel_k_override = 0;
    unsigned long long *d_k0 = nullptr, *d_k1 = nullptr; unsigned *d_sidx = nullptr; // sort keys [img][raw_pad]
    int raw_pad = 0;
    int group = 0, group_octaves = 1;                     // image-major schedule (rpe_sift_run): images per group, octaves inside it
    bool fused_all = false;                                // every blur radius has a fused instantiation
    bool march = false;                                    // levels 1-3 / 4-5 of the large octaves by sift_march_kernel
    int xtile_oct_end[12] = {0};                           // tiles of octaves 0 .. o end here in d_xtiles
    int ks[6] = {0, 0, 0, 0, 0, 0};                      // tap counts of c_skern
    float *d_fin = nullptr;                                // [img][kcap][6] un-halved keypoints in sorted order
};

// ------------------------------------------------------------------ image ops
// 2x INTER_LINEAR of the u8 image into f32 (oracle: sift_oracle.c upsample; cv2.resize semantics: source position
// (x + 0.5) / 2 - 0.5, clamped at the borders).  The interpolation weights are 0.25 / 0.75 and the pixels are integers
// below 256, so every product and sum of a*(1-f) + b*f is exact in f32 and any evaluation order gives the oracle's bits:
// the kernel works in integers, out = (wy0 (wx0 s00 + wx1 s01) + wy1 (wx0 s10 + wx1 s11)) / 16 with weights 1 and 3.
// A lane produces the 4 x 2 outputs x = 4t .. 4t+3, y = 2r+1, 2r+2 from source rows r, r+1 and columns 2t-1 .. 2t+2
// (clamping the indices reproduces the border rule: a clamped pair has two equal values, and (1 v + 3 v) / 4 = v).
__global__ __launch_bounds__(256) void sift_upsample_kernel(const uint8_t *__restrict__ img, int W, int H, size_t img_stride,
                                                             float *__restrict__ dst, long long dstride)
{
    const int bw = 2 * W, bh = 2 * H;
    const int t = blockIdx.x * 256 + threadIdx.x, r = (int)blockIdx.y - 1;       // r = -1 .. H-1
    if (4 * t >= bw) return;
    const uint8_t *s = img + (size_t)blockIdx.z * img_stride;
    const uint8_t *ra = s + (size_t)max(r, 0) * W, *rb = s + (size_t)min(r + 1, H - 1) * W;
    const int c0 = max(2 * t - 1, 0), c1 = min(2 * t, W - 1), c2 = min(2 * t + 1, W - 1), c3 = min(2 * t + 2, W - 1);
    const unsigned a0 = ra[c0], a1 = ra[c1], a2 = ra[c2], a3 = ra[c3];
    const unsigned b0 = rb[c0], b1 = rb[c1], b2 = rb[c2], b3 = rb[c3];
    // horizontal: x = 4t: (c0, c1) weights (1, 3); 4t+1: (c1, c2) (3, 1); 4t+2: (c1, c2) (1, 3); 4t+3: (c2, c3) (3, 1)
    const unsigned ha[4] = {a0 + 3 * a1, 3 * a1 + a2, a1 + 3 * a2, 3 * a2 + a3};
    const unsigned hb[4] = {b0 + 3 * b1, 3 * b1 + b2, b1 + 3 * b2, 3 * b2 + b3};
    typedef float f4_t __attribute__((ext_vector_type(4)));
    f4_t o1, o2;                                                                   // rows 2r+1 (weights 3, 1) and 2r+2 (1, 3)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o1[j] = (float)(3 * ha[j] + hb[j]) * 0.0625f;
        o2[j] = (float)(ha[j] + 3 * hb[j]) * 0.0625f;
    }
    float *d = dst + (long long)blockIdx.z * dstride + 4 * t;
    const int y1 = 2 * r + 1, y2 = 2 * r + 2;
    if (4 * t + 3 < bw) {
        if (y1 >= 0) *(f4_t *)(d + (size_t)y1 * bw) = o1;
        if (y2 < bh) *(f4_t *)(d + (size_t)y2 * bw) = o2;
    } else {
        for (int j = 0; j < 4 && 4 * t + j < bw; ++j) {
            if (y1 >= 0) d[(size_t)y1 * bw + j] = o1[j];
            if (y2 < bh) d[(size_t)y2 * bw + j] = o2[j];
        }
    }
}

__device__ __forceinline__ int s_refl(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; if (p >= n) p = 2 * n - 2 - p; }
    return p;
}

// row pass: one workgroup = 256 consecutive pixels of one row, taps staged in an LDS segment
__global__ __launch_bounds__(256) void sift_blur_row_kernel(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                             long long dstride, int w, int h, int kid)
{
    __shared__ float seg[256 + 32];
    const int ks = c_sks[kid], r = ks >> 1;
    const int x0 = blockIdx.x * 256, y = blockIdx.y;
    const float *s = src + (long long)blockIdx.z * sstride + (size_t)y * w;
    for (int i = threadIdx.x; i < 256 + 2 * r; i += 256) seg[i] = s[s_refl(x0 + i - r, w)];
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= w) return;
    float acc = c_skern[kid][0] * seg[threadIdx.x];                      // cv2's RowVec_32f order: first product, then fused taps
    for (int i = 1; i < ks; ++i) acc = __builtin_fmaf(c_skern[kid][i], seg[threadIdx.x + i], acc);
    dst[(long long)blockIdx.z * dstride + (size_t)y * w + x] = acc;
}

__global__ __launch_bounds__(256) void sift_blur_col_kernel(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                             long long dstride, int w, int h, int kid)
{
    const int ks = c_sks[kid], r = ks >> 1;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float *s = src + (long long)blockIdx.z * sstride;
    float acc = c_skern[kid][r] * s[(size_t)y * w + x];                  // cv2's SymmColumnVec_32f order: centre, then fused symmetric pairs
    for (int j = 1; j <= r; ++j) acc = __builtin_fmaf(c_skern[kid][r + j], s[(size_t)s_refl(y + j, h) * w + x] + s[(size_t)s_refl(y - j, h) * w + x], acc);
    dst[(long long)blockIdx.z * dstride + (size_t)y * w + x] = acc;
}

// Fused separable Gaussian: one workgroup = 64 x TH output pixels of one pyramid level.
// The (64+2R) x (TH+2R) source window (reflect-101 resolved at load time) is staged in LDS once, the
// row pass runs on all TH+2R window rows into a second LDS plane, the column pass reads that plane
// and writes G[i]: the level is read once and written once (the unfused version went through HBM
// between the passes).  Each lane accumulates 8 outputs from a register window of 8+2R values; tap
// order and accumulation order are cv2's sepFilter2D f32 route on its AVX2 build, as the oracle restates it
// (row: first product, then fma(k[i], v[i], acc), i ascending; column: centre, then fma(k[r+j], v[+j] + v[-j], acc)),
// so the f32 results are bit-identical.  (Rounds 1-2 summed every tap of both passes unfused: twice the
// vector instructions of this issue-limited kernel.)
__device__ __forceinline__ void asm_pin(float __attribute__((ext_vector_type(2))) &p) { asm volatile("" : "+v"(p)); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// P[m] = (lds[addr + 4*m*S], lds[addr + 4*(m+4)*S]) for every m of the sequence, S = 1 (ds_read2_b32) or 64 dwords
// (ds_read2st64_b32); all reads are issued before the one wait.  Inline asm because the offsets must be chosen per
// instruction; the compiler's own LDS wait counting does not see these, hence the explicit s_waitcnt and the pins.
__device__ __forceinline__ unsigned lds_addr(const float *p)     // byte offset inside the workgroup's LDS allocation
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
template <bool ST64, int M>
__device__ __forceinline__ void lds_pair(f32x2 &p, unsigned addr)
{
    if (ST64) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p) : "v"(addr), "n"(M), "n"(M + 4) : "memory");
    else      asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p) : "v"(addr), "n"(M), "n"(M + 4) : "memory");
}
template <int... Ms>
__device__ __forceinline__ void lds_pairs_sw(f32x2 *P, const unsigned *base4, std::integer_sequence<int, Ms...>)   // base by (m & 3)
{
    (lds_pair<true, Ms>(P[Ms], base4[Ms & 3]), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (asm_pin(P[Ms]), ...);
}
template <bool ST64, int... Ms>
__device__ __forceinline__ void lds_pairs(f32x2 *P, unsigned addr, std::integer_sequence<int, Ms...>)
{
    (lds_pair<ST64, Ms>(P[Ms], addr), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (asm_pin(P[Ms]), ...);
}
// 2x upsampled pixel (x, y) of a u8 image, as sift_upsample_kernel makes it (exact integer form, clamped source indices)
__device__ __forceinline__ float sift_up_at(const uint8_t *__restrict__ s8, int W, int H, int x, int y)
{
    const int t = (x - 1) >> 1, r = (y - 1) >> 1;                       // x odd: columns (t, t+1) weights (3, 1); even: (1, 3)
    const uint8_t *ra = s8 + (size_t)max(r, 0) * W, *rb = s8 + (size_t)min(r + 1, H - 1) * W;
    const int ca = max(t, 0), cb = min(t + 1, W - 1);
    const unsigned wx = (x & 1) ? 3u : 1u, wy = (y & 1) ? 3u : 1u;
    const unsigned ha = wx * ra[ca] + (4u - wx) * ra[cb], hb = wx * rb[ca] + (4u - wx) * rb[cb];
    return (float)(wy * ha + (4u - wy) * hb) * 0.0625f;
}

// UPS: the source level is the 2x upsampled u8 input image itself, formed while the window is loaded (first blur of the
// pyramid: the upsampled f32 image -- 33 MB per HD image written and read back -- never exists); u8a / u8b = the two
// image batches of the launch group (image index < na: first batch), W x H their size, w = 2 W, h = 2 H.
// dec (optional): level 0 of the next octave = this level at even x, even y (INTER_NEAREST halving), w2 x h2, written from
// the same registers (the separate decimation pass read the level back through 64-byte lines it used half of).
template <int R, int TH, bool UPS>
__global__ __launch_bounds__(256) void sift_blur_fused_kernel(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                               long long dstride, int w, int h, int kid, int tcols, int ntiles,
                                                               const uint8_t *__restrict__ u8a, const uint8_t *__restrict__ u8b, int na, int W, int H,
                                                               float *__restrict__ dec, int w2, int h2)
{
    // TH = tile height: the LDS footprint (WINH x (SP + 64) floats) decides how many workgroups a CU holds; at R >= 8 a
    // 64-row tile leaves 2 per CU and the kernel waits on its own window loads, a 32-row tile fits 4 (and 6 at R = 5, 6,
    // where it measured 1.5 % faster than 64 rows despite the taller halo)
    constexpr int WIN = 64 + 2 * R, WINH = TH + 2 * R, KS = 2 * R + 1;
    constexpr int HAL = (R + 3) & ~3, NCH = (64 + 2 * HAL) / 4;            // interior fast path: 16-B chunks from x0 - HAL on
    // a window row = [WIN values][PADF spare]: the fast path stores whole chunks, and the HAL - R columns a chunk row has on
    // either side of the window fall into the spare floats (of the row before / of the row itself); odd stride for the banks
    constexpr int PADF = HAL - R, SP = (WIN + PADF) | 1;
    __shared__ __attribute__((aligned(16))) float s_win[PADF + WINH * SP];
    float *const s_src = s_win + PADF;
    __shared__ float s_tmp[WINH * 64];
    const int tid = threadIdx.x;
    // XCD-aware order: each XCD walks a contiguous raster run of tiles, so neighbouring windows (2R-wide shared halos)
    // meet in one L2 instead of being fetched by two
    const int ti = ((blockIdx.x & 7) * ((ntiles + 7) >> 3)) + (blockIdx.x >> 3);
    if (ti >= ntiles) return;
    const int x0 = (ti % tcols) * 64, y0 = (ti / tcols) * TH;
    const float *s = src + (long long)blockIdx.y * sstride;
    const uint8_t *s8 = nullptr;
    if constexpr (UPS) s8 = (int)blockIdx.y < na ? u8a + (size_t)blockIdx.y * W * H : u8b + (size_t)((int)blockIdx.y - na) * W * H;
    if (x0 >= HAL && x0 + 64 + HAL <= w && y0 >= R && y0 + TH + R <= h) {
        // Interior tile (91 % of the tiles of a 3840 x 2160 octave): no reflection anywhere, so the window is fetched as
        // global_load_dwordx4 chunks (4-byte aligned is all the hardware asks of a multi-dword load).  A lane keeps its
        // chunk column and walks down the rows (RPP rows per pass), so addresses and LDS offsets advance by wave-uniform
        // constants: ~25 vector instructions per lane for the whole window (per-chunk index arithmetic and guarded
        // element stores were ~100, a fifth of this issue-bound kernel).
        constexpr int RPP = 256 / NCH, NQ = (WINH + RPP - 1) / RPP;
        typedef float f4_t __attribute__((ext_vector_type(4)));
        f4_t stage[NQ];
        // lanes past the last chunk of a pass and rows past the window repeat the last valid chunk / row: same address,
        // same data, so their loads and stores are harmless duplicates and nothing is predicated
        const int t = min(tid, RPP * NCH - 1);
        const int lr = t / NCH, c4 = t - lr * NCH;
        const float *a = s + (size_t)(y0 - R + lr) * w + (x0 - HAL + 4 * c4);
        float *d = s_src + lr * SP + 4 * c4 - PADF;
        constexpr int LASTQ = NQ - 1, LAST_FULL = RPP * NQ <= WINH;
        const int lrl = LAST_FULL ? lr : min(lr, WINH - 1 - RPP * LASTQ);   // row of the last pass, clamped into the window
        if constexpr (UPS) {
            // chunk = upsampled pixels x = 4t .. 4t+3 of row y: source columns 2t-1 .. 2t+2 of rows r, r+1 (r = (y-1) >> 1),
            // horizontal weights (1,3) (3,1) (1,3) (3,1), vertical (3,1) for odd y and (1,3) for even y; integers, exact
            const int tq = (x0 - HAL + 4 * c4) >> 2;
            const int cm = 2 * tq - 1, cp = min(2 * tq + 2, W - 1);              // interior tile: cm >= 0
            unsigned bytes[NQ][2][4];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int wy = y0 - R + (q < LASTQ ? lr + RPP * q : lrl + RPP * q);
                const int r = (wy - 1) >> 1;
                const uint8_t *ra = s8 + (size_t)max(r, 0) * W, *rb = s8 + (size_t)min(r + 1, H - 1) * W;
                bytes[q][0][0] = ra[cm]; bytes[q][0][1] = ra[2 * tq]; bytes[q][0][2] = ra[2 * tq + 1]; bytes[q][0][3] = ra[cp];
                bytes[q][1][0] = rb[cm]; bytes[q][1][1] = rb[2 * tq]; bytes[q][1][2] = rb[2 * tq + 1]; bytes[q][1][3] = rb[cp];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int wy = y0 - R + (q < LASTQ ? lr + RPP * q : lrl + RPP * q);
                const unsigned wa = (wy & 1) ? 3u : 1u, wb = 4u - wa;
                const unsigned *a4 = bytes[q][0], *b4 = bytes[q][1];
                const unsigned ha[4] = {a4[0] + 3 * a4[1], 3 * a4[1] + a4[2], a4[1] + 3 * a4[2], 3 * a4[2] + a4[3]};
                const unsigned hb[4] = {b4[0] + 3 * b4[1], 3 * b4[1] + b4[2], b4[1] + 3 * b4[2], 3 * b4[2] + b4[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) stage[q][e] = (float)(wa * ha[e] + wb * hb[e]) * 0.0625f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float *aq = q < LASTQ ? a + (size_t)(RPP * q) * w : a + (size_t)(RPP * q + lrl - lr) * w;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage[q]) : "v"(aq) : "memory");   // all in flight together
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            asm volatile("" : "+v"(stage[q]));
            float *dq = q < LASTQ ? d + RPP * q * SP : d + (RPP * q + lrl - lr) * SP;
#pragma unroll
            for (int e = 0; e < 4; ++e) dq[e] = stage[q][e];
        }
    } else if (w > R && h > R) {
        // window load, all requests in flight before the first LDS store: with one reflection being
        // enough (n > R; coordinates past n-1+R only feed outputs outside the image and are clamped)
        // the addresses are branch-free, so the loads are issued back to back instead of one
        // HBM round trip per element
        constexpr int NLD = (WINH * WIN + 255) / 256;
        float stage[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = min(tid + 256 * q, WINH * WIN - 1);
            const int r = i / WIN, c = i - r * WIN;
            int yy = min(max(y0 + r - R, -R), h - 1 + R), xx = min(max(x0 + c - R, -R), w - 1 + R);
            yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
            xx = xx < 0 ? -xx : xx; xx = xx >= w ? 2 * w - 2 - xx : xx;
            if constexpr (UPS) stage[q] = sift_up_at(s8, W, H, xx, yy); else stage[q] = s[(size_t)yy * w + xx];
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = tid + 256 * q;
            const int r = i / WIN, c = i - r * WIN;
            if (i < WINH * WIN) s_src[r * SP + c] = stage[q];
        }
    } else {
        for (int i = tid; i < WINH * WIN; i += 256) {
            const int r = i / WIN, c = i - r * WIN;
            const int yy = s_refl(y0 + r - R, h), xx = s_refl(x0 + c - R, w);
            if constexpr (UPS) s_src[r * SP + c] = sift_up_at(s8, W, H, xx, yy); else s_src[r * SP + c] = s[(size_t)yy * w + xx];
        }
    }
    float k[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) k[i] = c_skern[kid][i];
    __syncthreads();
    // Both passes: a lane produces 8 outputs as 4 packed pairs (j, j+4); the tap window is held as
    // pairs P[m] = (v[m], v[m+4]) so that every multiply, add and fused multiply-add is a v_pk_*_f32 (two IEEE f32
    // operations per instruction, same rounding as the scalar ones).  The pairs come
    // straight out of LDS: ds_read2_b32 / ds_read2st64_b32 take two independent offsets, so one
    // instruction returns (v[m], v[m+4]) in an aligned register pair (left to the compiler the loads
    // pair up as (v[m], v[m+1]) and 40 v_mov per item rebuild the operands: 16 % of the kernel).
    // The first product starts the sum: taps and pixels are >= 0, so 0 + k*v == k*v bit for bit.
    // row pass: item = (window row, group of 8 columns).  Lane -> item so that a 32-lane half holds 8 rows x 4 groups:
    // its read addresses r * SP + 8 g + m (SP odd) then fall on 32 different banks (rows x 8 groups would alias g and
    // g + 4).  The row-pass plane is stored with the column XORed by (row & 3): a plain [row][64] store would put all 32
    // lanes of a half on 4 banks (8-way); swizzled it is 2-way, which a store does not feel.
    for (int it = tid; it < (WINH + 7) / 8 * 64; it += 256) {
        const int g = ((it >> 3) & 4) | (it & 3), r = (it >> 6) * 8 + ((it >> 2) & 7);
        if (r >= WINH) continue;
        f32x2 P[4 + 2 * R];
        lds_pairs<false>(P, lds_addr(s_src + r * SP + 8 * g), std::make_integer_sequence<int, 4 + 2 * R>());
        float *o = s_tmp + r * 64 + 8 * g;
        float *const o0 = o + (0 ^ (r & 3)), *const o1 = o + (1 ^ (r & 3)), *const o2 = o + (2 ^ (r & 3)), *const o3 = o + (3 ^ (r & 3));
        float *const oj[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 acc = f32x2{k[0], k[0]} * P[j];
#pragma unroll
            for (int i = 1; i < KS; ++i) { const f32x2 kk = {k[i], k[i]}; acc = __builtin_elementwise_fma(kk, P[j + i], acc); }      // RowVec_32f: one v_pk_fma_f32 per tap pair
            oj[j][0] = acc.x;
            oj[j][4] = acc.y;
        }
    }
    __syncthreads();
    // column pass: a wave owns 8-row groups of the tile, a lane one column.  Inside the image the wave turns its 8 x 64
    // results through LDS (the window plane is dead after the row pass; each wave uses its own 2 KB of it, no barrier)
    // and writes them as two 16-byte stores per lane = four 256-byte rows per instruction instead of 8 dword stores
    // (measured: the same time -- the stores of this kernel cost by the byte, a third of its time -- but fewer
    // instructions).
    const int lane = tid & 63;
    const int x = x0 + lane;
    const long long ob = (long long)blockIdx.y * dstride;
    typedef float f4_t __attribute__((ext_vector_type(4)));
    for (int g = __builtin_amdgcn_readfirstlane(tid >> 6); g < TH / 8; g += 4) {
        f32x2 P[4 + 2 * R];
        // rows 8 g + m and 8 g + m + 4 share (row & 3) = m & 3: one base address per residue undoes the swizzle
        const unsigned cb = lds_addr(s_tmp + (8 * g) * 64);
        const unsigned base4[4] = {cb + 4 * (lane ^ 0), cb + 4 * (lane ^ 1), cb + 4 * (lane ^ 2), cb + 4 * (lane ^ 3)};
        lds_pairs_sw(P, base4, std::make_integer_sequence<int, 4 + 2 * R>());
        const int yb = y0 + 8 * g;
        float *pd = dst + ob + (size_t)yb * w + x0;
        const int rows_left = h - yb;
        const bool whole = rows_left >= 8 && x0 + 64 <= w;      // wave-uniform
        float *so = s_win + (8 * g) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // SymmColumnVec_32f: centre tap, then one v_pk_add_f32 + one v_pk_fma_f32 per symmetric tap pair
            f32x2 acc = f32x2{k[R], k[R]} * P[j + R];
#pragma unroll
            for (int t = 1; t <= R; ++t) { const f32x2 kk = {k[R + t], k[R + t]}; acc = __builtin_elementwise_fma(kk, P[j + R + t] + P[j + R - t], acc); }
            if (whole) {
                so[j * 64 + lane] = acc.x;
                so[(j + 4) * 64 + lane] = acc.y;
            } else {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int yl = j + 4 * hh;
                    if (yl < rows_left && x < w) {
                        (pd + (size_t)yl * w)[lane] = hh ? acc.y : acc.x;
                        if (dec && !((yb + yl) & 1) && !(x & 1) && ((yb + yl) >> 1) < h2 && (x >> 1) < w2)
                            dec[(long long)blockIdx.y * dstride + (size_t)((yb + yl) >> 1) * w2 + (x >> 1)] = hh ? acc.y : acc.x;
                    }
                }
            }
        }
        if (whole) {
            const int rr = lane >> 4, c4 = (lane & 15) * 4;
            const unsigned voff = (unsigned)(rr * w + c4) * 4u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f4_t v = *(const f4_t *)(so + (4 * t + rr) * 64 + c4);
                // scalar row base + 32-bit lane offset
                asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(voff), "v"(v), "s"(pd + (size_t)(4 * t) * w) : "memory");
                if (dec && !(rr & 1)) {                         // rows yb + 4t + rr even (yb is a multiple of 8), columns c4, c4 + 2
                    const int y2 = (yb + 4 * t + rr) >> 1, x2 = (x0 + c4) >> 1;
                    float *q = dec + (long long)blockIdx.y * dstride + (size_t)y2 * w2 + x2;
                    if (y2 < h2) { if (x2 < w2) q[0] = v[0]; if (x2 + 1 < w2) q[1] = v[2]; }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ marching pyramid
// NL consecutive levels of one octave in ONE pass over the source level: a workgroup owns a strip of MARCH_SW columns and
// marches down the image 8 rows per step.  Level j keeps a ring of its last 2 R_j + 8 row-filtered rows in LDS; its column
// filter produces the 8 rows that lie R_j rows above the newest one, which are stored and are at once the source rows of
// level j + 1's row filter.  The source level is read once (the tile kernels read every level back), every row filter runs
// once per row (the 32-row tiles ran it on 32 + 2R rows), and there are no window loads.
// Arithmetic and its order are those of sift_blur_fused_kernel, bit for bit.  Borders: the column filter adds its taps as
// symmetric pairs, so running the march over a source extended by reflect-101 above and below (P = sum of the radii rows)
// yields every level's own reflect-101 extension exactly -- no special rows.  The row filter accumulates left to right and
// is not symmetric: at the image's left / right edge a level's out-of-image columns are written as copies of its mirror
// columns (the column-filter lane of such a column reads the ring at the mirrored column).
#ifndef MARCH_SW
#define MARCH_SW 128
#endif
#define MARCH_NT (MARCH_SW >= 128 ? 256 : 128)     // threads per workgroup: one lane per column of the widest level
// (lds[addr + 4 TS i], lds[addr + 4 TS (i + 4)]) for every i of the sequence: each half is its own ds_read_b32 at a static
// offset, so it lands in its half of the register pair (left to the compiler the two uses of a value share one load and
// ~100 v_mov per step rebuild the packed operands)
template <int TS, int I>
__device__ __forceinline__ void lds_colpair(f32x2 &p, unsigned addr)
{
    float a, b;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(4 * TS * I) : "memory");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(4 * TS * (I + 4)) : "memory");
    p = f32x2{a, b};
}
template <int TS, int... Is>
__device__ __forceinline__ void lds_colpairs(f32x2 *P, unsigned addr, std::integer_sequence<int, Is...>)
{
    (lds_colpair<TS, Is>(P[Is], addr), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (asm_pin(P[Is]), ...);
}
template <int NL, int R0, int R1, int R2>
struct MarchGeo {
    static constexpr int R(int j) { return j == 0 ? R0 : j == 1 ? R1 : R2; }
    static constexpr int P = R0 + R1 + (NL > 2 ? R2 : 0);
    static constexpr int H(int j) { int a = 0; for (int m = j + 1; m < NL; ++m) a += R(m); return a; }     // halo columns of level j on either side
    static constexpr int L(int j) { int a = 0; for (int m = 0; m <= j; ++m) a += R(m); return a; }         // rows level j lags behind the source
    static constexpr int W(int j) { return MARCH_SW + 2 * H(j); }
    static constexpr int NG(int j) { return (W(j) + 7) / 8; }
    static constexpr int SRCW(int j) { return 8 * NG(j) + 2 * R(j); }      // columns of level j's source rows that are read
    static constexpr int SS(int j) { return SRCW(j) | 1; }
    static constexpr int TS(int j) { return (8 * NG(j)) | 1; }
    static constexpr int D(int j) { return 2 * R(j) + 8; }
    static constexpr int OFF_SRC(int j) { int a = 0; for (int m = 0; m < j; ++m) a += 8 * SS(m) + D(m) * TS(m); return a; }
    static constexpr int OFF_T(int j) { return OFF_SRC(j) + 8 * SS(j); }
    static constexpr int TOTAL = OFF_SRC(NL);
    // waves have roles: level j owns WV(j) waves (one lane per row-filter item / per column)
    static constexpr int WV(int j) { return (NG(j) + 7) / 8; }
    static constexpr int WBASE(int j) { int a = 0; for (int m = 0; m < j; ++m) a += WV(m); return a; }
    static constexpr int NT = 64 * WBASE(NL);
};

template <class G, int J>
__device__ __forceinline__ void march_rowfilter(float *lds, int kid, int tid)
{
    constexpr int R = G::R(J), KS = 2 * R + 1, NGJ = G::NG(J), SSJ = G::SS(J), TSJ = G::TS(J);
    float k[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) k[i] = c_skern[kid][i];
    const float *srcb = lds + G::OFF_SRC(J);
    float *win = lds + G::OFF_T(J) + 2 * R * TSJ;              // the 8 new rows go behind the 2R rows carried over
    {
        const int it = tid;                                      // tid = lane index inside the level's waves
        const int r = (it >> 2) & 7, g = 8 * (it >> 6) + (((it >> 3) & 4) | (it & 3));
        if (g >= NGJ) return;
        f32x2 P[4 + 2 * R];
        lds_pairs<false>(P, lds_addr(srcb + r * SSJ + 8 * g), std::make_integer_sequence<int, 4 + 2 * R>());
        float *o = win + r * TSJ + 8 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 acc = f32x2{k[0], k[0]} * P[j];
#pragma unroll
            for (int i = 1; i < KS; ++i) { const f32x2 kk = {k[i], k[i]}; acc = __builtin_elementwise_fma(kk, P[j + i], acc); }
            o[j] = acc.x;
            o[j + 4] = acc.y;
        }
    }
}

template <class G, int J, int NL>
__device__ __forceinline__ void march_colfilter(float *lds, int kid, int tid, int s, int x0, int w, int h, float *__restrict__ plane,
                                                float *__restrict__ decp, int w2, int h2)
{
    constexpr int R = G::R(J), WJ = G::W(J), HJ = G::H(J), TSJ = G::TS(J);
    const int c = x0 - HJ + tid;                                  // absolute column of this lane
    if (tid >= WJ || c < 0 || c >= w) return;                     // out-of-image columns are written by the lane of their mirror column
    float k[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) k[i] = c_skern[kid][R + i];
    // the window of this column: rows 0 .. 2R-1 carried over from the previous step, rows 2R .. 2R+7 new; static addresses
    float *win = lds + G::OFF_T(J) + tid;
    f32x2 P[4 + 2 * R];
    lds_colpairs<TSJ>(P, lds_addr(win), std::make_integer_sequence<int, 4 + 2 * R>());
    // the last 2R rows move up for the next step (this lane is the only one that touches its column)
#pragma unroll
    for (int i = 0; i < 2 * R; ++i) win[i * TSJ] = i + 8 < 4 + 2 * R ? P[i + 8].x : P[i + 4].y;
    f32x2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[j] = f32x2{k[0], k[0]} * P[j + R];
#pragma unroll
        for (int t = 1; t <= R; ++t) { const f32x2 kk = {k[t], k[t]}; acc[j] = __builtin_elementwise_fma(kk, P[j + R + t] + P[j + R - t], acc[j]); }
    }
    if constexpr (J + 1 < NL) {                                   // source rows of the next level (this level's column range)
        constexpr int SSN = G::SS(J + 1 < NL ? J + 1 : J);
        float *nb = lds + G::OFF_SRC(J + 1 < NL ? J + 1 : J);
#pragma unroll
        for (int j = 0; j < 4; ++j) { nb[j * SSN + tid] = acc[j].x; nb[(j + 4) * SSN + tid] = acc[j].y; }
        // reflect-101 of this level at the image's left / right edge: the mirror columns inside the strip's range
        const int ml = -c - (x0 - HJ), mr = 2 * (w - 1) - c - (x0 - HJ);
        if (c > 0 && ml >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { nb[j * SSN + ml] = acc[j].x; nb[(j + 4) * SSN + ml] = acc[j].y; }
        }
        if (c < w - 1 && mr < WJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { nb[j * SSN + mr] = acc[j].x; nb[(j + 4) * SSN + mr] = acc[j].y; }
        }
    }
    const int y0 = -G::P + 8 * s - G::L(J);                       // first of the 8 rows produced in this step
#ifdef MARCH_DIAG_NOSTORE
    if (acc[0].x == 12345.678f)
#endif
    if (tid >= HJ && tid < HJ + MARCH_SW && y0 + 7 >= 0 && y0 < h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int y = y0 + j + 4 * hh;
                const float v = hh ? acc[j].y : acc[j].x;
                if (y >= 0 && y < h) {
                    (plane + (size_t)y * w)[c] = v;
                    if (decp && !(y & 1) && !(c & 1) && (y >> 1) < h2 && (c >> 1) < w2) decp[(size_t)(y >> 1) * w2 + (c >> 1)] = v;
                }
            }
        }
    }
}

template <int NL, int R0, int R1, int R2>
__global__ __launch_bounds__(64 * (MarchGeo<NL, R0, R1, R2>::WBASE(NL))) void sift_march_kernel(
    const float *__restrict__ src, long long sstride, float *__restrict__ dst, long long dstride,
    long long pn, int w, int h, int kid0, float *__restrict__ dec, int dec_level, int w2, int h2)
{
    // All levels work in the same phase on different steps: in iteration t level j is at step t - j (its source rows were
    // written one iteration earlier), so an iteration has two phases and two barriers whatever NL is -- every level's row
    // filter, then every level's column filter -- and every wave has work in both.
    typedef MarchGeo<NL, R0, R1, R2> G;
    constexpr int NT = G::NT;
    __shared__ __attribute__((aligned(16))) float lds[G::TOTAL];
    const int tid = threadIdx.x, x0 = blockIdx.x * MARCH_SW;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // the wave's role is a scalar
    const float *s = src + (long long)blockIdx.y * sstride;
    float *d = dst + (long long)blockIdx.y * dstride;
    float *decp = dec ? dec + (long long)blockIdx.y * dstride : nullptr;
    // this lane's share of the 8 x SRCW(0) source values of a step: fixed (row, column) slots, only the row base moves
    constexpr int SW0 = G::SRCW(0), NLD = (8 * SW0 + NT - 1) / NT;
    int lrow[NLD], lcol[NLD], loff[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int idx = min(tid + NT * q, 8 * SW0 - 1);
        const int r = idx / SW0, cc = idx - r * SW0;
        int c = x0 - G::P + cc;
        c = c < 0 ? -c : c; c = c >= w ? 2 * w - 2 - c : c; c = min(max(c, 0), w - 1);
        lrow[q] = r; lcol[q] = c; loff[q] = r * G::SS(0) + cc;
    }
    // Source rows are requested two steps ahead into two register sets.  A set is written to LDS right after the row phase
    // (whose readers of the previous rows are past the barrier), i.e. BEFORE this iteration's level stores are issued: vmcnt
    // counts in order, so waiting for a set drains every older request -- with the wait placed after the stores (one set, one
    // step ahead) every iteration sat out the write acknowledgement of the stores it had just issued.
    float sA[NLD], sB[NLD];
    auto fetch = [&](float (&stage)[NLD], int step) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            int y = -G::P + 8 * step + lrow[q];
            y = y < 0 ? -y : y; y = y >= h ? 2 * h - 2 - y : y; y = min(max(y, 0), h - 1);
            stage[q] = s[(size_t)y * w + lcol[q]];
        }
    };
    auto put = [&](const float (&stage)[NLD]) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) lds[G::OFF_SRC(0) + loff[q]] = stage[q];      // (the clamped duplicates of the last slot rewrite the same value)
    };
    const int nsteps = (h + 2 * G::P + 7) / 8;
    // iteration t: level j is at step t - j (its source rows were written one iteration earlier)
    auto iter = [&](int t, float (&cur)[NLD]) {
        if (wv < G::WBASE(1)) { if (t < nsteps) march_rowfilter<G, 0>(lds, kid0, tid); }
        else if (NL > 1 && wv < G::WBASE(NL > 1 ? 2 : 1)) { if (t >= 1 && t - 1 < nsteps) march_rowfilter<G, (NL > 1 ? 1 : 0)>(lds, kid0 + 1, tid - 64 * G::WBASE(1)); }
        else if (NL > 2) { if (t >= 2 && t - 2 < nsteps) march_rowfilter<G, (NL > 2 ? 2 : 0)>(lds, kid0 + 2, tid - 64 * G::WBASE(NL > 2 ? 2 : 1)); }
        __syncthreads();
        if (t + 1 < nsteps) put(cur);                            // source rows of step t + 1 (requested two iterations ago)
        if (wv < G::WBASE(1)) { if (t < nsteps) march_colfilter<G, 0, NL>(lds, kid0, tid, t, x0, w, h, d, dec_level == 0 ? decp : nullptr, w2, h2); }
        else if (NL > 1 && wv < G::WBASE(NL > 1 ? 2 : 1)) { if (t >= 1 && t - 1 < nsteps) march_colfilter<G, (NL > 1 ? 1 : 0), NL>(lds, kid0 + 1, tid - 64 * G::WBASE(1), t - 1, x0, w, h, d + pn, dec_level == 1 ? decp : nullptr, w2, h2); }
        else if (NL > 2) { if (t >= 2 && t - 2 < nsteps) march_colfilter<G, (NL > 2 ? 2 : 0), NL>(lds, kid0 + 2, tid - 64 * G::WBASE(NL > 2 ? 2 : 1), t - 2, x0, w, h, d + 2 * pn, dec_level == 2 ? decp : nullptr, w2, h2); }
        // requested AFTER this iteration's stores and unconditionally (rows past the end are clamped): vmcnt counts in order, so the
        // next wait (for the other set) can name exactly these NLD requests as the ones that may stay in flight
        fetch(cur, t + 3);
        __syncthreads();                                         // every window access of this iteration done; step t + 1's rows in place
    };
    fetch(sA, 0); fetch(sB, 1);
    put(sA);
    __syncthreads();
    fetch(sA, 2);
    const int T = nsteps + NL - 1;
    for (int t = 0; t < T; t += 2) {
        iter(t, sB);
        iter(t + 1, sA);                                         // (an iteration past the end finds nothing to do)
    }
}

// INTER_NEAREST octave decimation: dst(x, y) = src(2x, 2y).  A lane writes 4 neighbouring outputs with one 16-byte store
// from two 16-byte loads (one dword per lane each way left this copy issue-bound in its many small launches).
__global__ __launch_bounds__(256) void sift_halve_kernel(const float *__restrict__ src, float *__restrict__ dst, long long stride,
                                                          int sw, int w, int h)
{
    const int x = 4 * (blockIdx.x * 256 + threadIdx.x), y = blockIdx.y;
    if (x >= w) return;
    const long long b = (long long)blockIdx.z * stride;
    const float *sp = src + b + (size_t)(2 * y) * sw + 2 * x;
    float *dp = dst + b + (size_t)y * w + x;
    typedef float f4_t __attribute__((ext_vector_type(4), aligned(4)));
    if (x + 3 < w) {
        const f4_t a0 = *(const f4_t *)sp, a1 = *(const f4_t *)(sp + 4);
        f4_t o; o[0] = a0[0]; o[1] = a0[2]; o[2] = a1[0]; o[3] = a1[2];
        *(f4_t *)dp = o;
    } else {
        for (int j = 0; x + j < w; ++j) dp[j] = sp[2 * j];
    }
}

// d = a - b (DoG of one level pair; only the unfused fallback path uses it)
__global__ __launch_bounds__(256) void sift_sub_kernel(const float *__restrict__ a, long long astride, const float *__restrict__ b,
                                                        long long bstride, float *__restrict__ d, long long dstride, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    d[(long long)blockIdx.z * dstride + i] = a[(long long)blockIdx.z * astride + i] - b[(long long)blockIdx.z * bstride + i];
}

// ------------------------------------------------------------------ extrema
__device__ __forceinline__ int s_block_excl_scan(int v, int *s_wave, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(inc, o); if (lane >= o) inc += n; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { int s = s_wave[k]; if (k < wv) base += s; }
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + inc - v;
}

// Pass 1 (tiled, any order): 26-neighbour test of the three inner DoG layers of a 128x16 tile (rows of 520 bytes
// per level: 16.9 -> 16.2 ms against 64x32 tiles with 264-byte rows; 256x8 reads 18 % more halo and gains nothing).  The
// five layers roll through three LDS planes (each DoG value leaves HBM once per tile).  "No
// neighbour is greater" <=> val >= max of the 3x3x3 block (val itself included): per lane (= column)
// the max/min over 3 layers x 3 columns of one row (9 LDS reads per new row; the planes carry the
// halo columns), rolled over 3 rows.  A wave owns 8 consecutive rows; the hit flags of a row are a
// 64-bit ballot = one mask word, its popcount goes to the row's band counter (integer atomic:
// deterministic).
#define SX_TW 128
#define SX_TH 16
#define SX_P 131
__global__ __launch_bounds__(256) void sift_extrema_mask_kernel(const float *__restrict__ gauss, SiftDev dv, const SiftXTile *__restrict__ tiles,
                                                                 unsigned long long *__restrict__ mask, int *__restrict__ band_cnt, int ntiles)
{
    __shared__ float s_d[3][(SX_TH + 2) * SX_P];
    // XCD-aware order (workgroups are dealt round-robin over 8 XCDs): a contiguous raster run of tiles per XCD, so the
    // 128-B lines straddling two tiles (264-B tile rows) and the halo rows are fetched into one L2, not two
    const int ti = ((blockIdx.x & 7) * ((ntiles + 7) >> 3)) + (blockIdx.x >> 3);
    if (ti >= ntiles) return;
    const SiftXTile t = tiles[ti];
    const int img = blockIdx.y, o = t.o, w = dv.w[o], h = dv.h[o], x0 = t.x0, y0 = t.y0;
    const long long n = (long long)w * h;
    const float *d = gauss + (long long)img * dv.gstride + dv.goff[o];     // Gaussian levels; DoG l = G[l+1] - G[l]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index in a scalar register: row numbers, mask and counter addresses stay scalar
    constexpr int NEL = (SX_TH + 2) * (SX_TW + 2), NLD = (NEL + 255) / 256;
    float stage[NLD], prevg[NLD];
    // element q of this lane: clamped source offset inside a level (32-bit, the level base stays scalar) and LDS offset;
    // both are the same for all six levels
    unsigned goffs[NLD]; int loffs[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int i = min(tid + 256 * q, NEL - 1);
        const int r = i / (SX_TW + 2), c = i - r * (SX_TW + 2);
        const int y = min(max(y0 - 1 + r, 0), h - 1), x = min(max(x0 - 1 + c, 0), w - 1);
        goffs[q] = (unsigned)(y * w + x);
        loffs[q] = tid + 256 * q < NEL ? r * SX_P + c : -1;
    }
    auto fetch = [&](int layer) {               // all loads of a Gaussian level in flight
        const float *src = d + layer * n;
#pragma unroll
        for (int q = 0; q < NLD; ++q) stage[q] = src[goffs[q]];
    };
    auto keep = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) prevg[q] = stage[q];
    };
    auto commit = [&](int slot) {               // DoG = this level - the level below, into an LDS plane
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            if (q < NLD - 1 || loffs[q] >= 0) s_d[slot][loffs[q]] = stage[q] - prevg[q];
            prevg[q] = stage[q];
        }
    };
    fetch(0); keep();
    fetch(1); commit(0);
    fetch(2); commit(1);
    fetch(3);
    unsigned long long *mk = mask + (long long)img * dv.bmstride + dv.bmoff[o];
    int *bc = band_cnt + (long long)img * dv.nbands + dv.band0[o];
    for (int l = 1; l <= S_NOL; ++l) {
        commit((l + 1) % 3);
        __syncthreads();
        if (l < S_NOL) fetch(l + 3);                        // next level's loads fly during the tests
        const float *lo = s_d[(l - 1) % 3], *mid = s_d[l % 3], *hi = s_d[(l + 1) % 3];
        // per LDS row: max / min over the 3 layers x 3 columns around this lane's pixel (the plane carries the halo
        // columns, so the left / right neighbours are plain LDS reads: no cross-lane traffic, no halo special case);
        // rolled over 3 rows it is the 27-value max / min.  ctr = the centre value of the row.
        float pmx[3], pmn[3], ctr[3];
        constexpr int WX = SX_TW / 64;                       // waves side by side; each wave tests 64 columns x 8 rows
        static_assert(WX * (SX_TH / 8) == 4, "four waves cover the tile");
        const int rb = 8 * (wv / WX), cb = 64 * (wv % WX);   // first output row / column of this wave (tile-relative)
        auto rowmm = [&](int lr, float &mx, float &mn, float &cv) {       // lr = LDS row
            const int q = lr * SX_P + cb + lane;
            const float a0 = lo[q], a1 = lo[q + 1], a2 = lo[q + 2];
            const float b0 = mid[q], b1 = mid[q + 1], b2 = mid[q + 2];
            const float c0 = hi[q], c1 = hi[q + 1], c2 = hi[q + 2];
            mx = fmaxf(fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(b0, b1), b2)), fmaxf(fmaxf(c0, c1), c2));
            mn = fminf(fminf(fminf(fminf(a0, a1), a2), fminf(fminf(b0, b1), b2)), fminf(fminf(c0, c1), c2));
            cv = b1;
        };
        rowmm(rb, pmx[0], pmn[0], ctr[0]);
        rowmm(rb + 1, pmx[1], pmn[1], ctr[1]);
        const int x = x0 + cb + lane;
        const bool xin = x >= S_BORDER && x < w - S_BORDER;
        // mask word / band counter of this wave's first row; the rows below are one word row / one counter further (the scalar
        // unit is shared by the CU's four SIMDs: per-row 64-bit index arithmetic made this kernel issue as many scalar
        // instructions as vector ones)
        const int wpr = dv.wpr[o];
        unsigned long long *mrow = mk + ((long long)(l - 1) * h + (y0 + rb)) * wpr + ((x0 + cb) >> 6);
        int *brow = bc + (l - 1) * (h - 2 * S_BORDER) + (y0 + rb - S_BORDER);
        const bool colin = x0 + cb < w;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int y = y0 + rb + k;
            rowmm(rb + k + 2, pmx[2], pmn[2], ctr[2]);
            const float M = fmaxf(fmaxf(pmx[0], pmx[1]), pmx[2]), m = fminf(fminf(pmn[0], pmn[1]), pmn[2]);
            const float val = ctr[1];
            if (y >= S_BORDER && y < h - S_BORDER && colin) {            // wave-uniform
                // "no neighbour is greater / smaller": M and m include val itself, so val >= M <=> val == M, val <= m <=> val == m
                const bool hit = xin && fabsf(val) > 1.f && val == (val > 0.f ? M : m);
                const unsigned long long bal = __ballot(hit);
                if (lane == 0) {
                    mrow[(long long)k * wpr] = bal;
                    if (bal) atomicAdd(brow + k, __popcll(bal));
                }
            }
            pmx[0] = pmx[1]; pmx[1] = pmx[2]; pmn[0] = pmn[1]; pmn[1] = pmn[2]; ctr[0] = ctr[1]; ctr[1] = ctr[2];
        }
        __syncthreads();
    }
}

// exclusive scan of the band counters of one image (256 threads, a run of consecutive bands each)
__global__ __launch_bounds__(256) void sift_band_scan_kernel(const int *__restrict__ band_cnt, int *__restrict__ band_off, int *__restrict__ nseeds,
                                                              unsigned *__restrict__ overflow, SiftDev dv)
{
    __shared__ int s_wave[5];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int per = (dv.nbands + 255) / 256, b0 = tid * per, b1 = min(b0 + per, dv.nbands);
    const int *bc = band_cnt + (long long)img * dv.nbands;
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += bc[b];
    int total;
    int acc = s_block_excl_scan(sum, s_wave, total);
    int *bo = band_off + (long long)img * dv.nbands;
    for (int b = b0; b < b1; ++b) { bo[b] = acc; acc += bc[b]; }
    if (tid == 0) {
        nseeds[img] = total < dv.seed_cap ? total : dv.seed_cap;
        if (total > dv.seed_cap) atomicOr(&overflow[img], (unsigned)RPE_OVF_SIFT_SEEDS);
    }
}

// Pass 2: one wave per band (row) walks the row's mask words in order and writes the seeds at the
// band's offset: raster order inside the row, bands in (octave, layer, row) order.
__global__ __launch_bounds__(256) void sift_extrema_emit_kernel(const unsigned long long *__restrict__ mask, SiftDev dv,
                                                                 const int *__restrict__ band_cnt, const int *__restrict__ band_off,
                                                                 unsigned *__restrict__ seeds)
{
    const int lane = threadIdx.x & 63, band = blockIdx.x * 4 + (threadIdx.x >> 6), img = blockIdx.y;
    if (band >= dv.nbands) return;
    if (band_cnt[(long long)img * dv.nbands + band] == 0) return;
    int o = 0;
    for (int k = 1; k < dv.noct; ++k) if (band >= dv.band0[k]) o = k;
    const int hv = dv.h[o] - 2 * S_BORDER, rem = band - dv.band0[o];
    const int l = rem / hv + 1, r = rem - (l - 1) * hv + S_BORDER, wpr = dv.wpr[o];
    const unsigned long long *words = mask + (long long)img * dv.bmstride + dv.bmoff[o] + ((long long)(l - 1) * dv.h[o] + r) * wpr;
    int base = band_off[(long long)img * dv.nbands + band];
    unsigned *out = seeds + (long long)img * dv.seed_cap;
    for (int w0 = 0; w0 < wpr; w0 += 64) {
        const int wi = w0 + lane;
        unsigned long long m = wi < wpr ? words[wi] : 0ull;
        const int cnt = __popcll(m);
        int inc = cnt;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { int v = __shfl_up(inc, s); if (lane >= s) inc += v; }
        int idx = base + inc - cnt;
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (idx < dv.seed_cap) out[idx] = ((unsigned)o << 28) | ((unsigned)l << 26) | ((unsigned)r << 13) | (unsigned)(wi * 64 + b);
            ++idx;
        }
        base += __shfl(inc, 63);
    }
}

// ------------------------------------------------------- refine + orientation
// DoG layer l = G[l+1] - G[l], formed where it is consumed (the DoG pyramid is never stored: 20 B per octave pixel less
// HBM traffic in the blur, 221 MB per HD image less workspace); same f32 subtraction as a stored DoG
struct DogCtx { const float *d; long long n; int w, h; };      // d = Gaussian levels of the octave
__device__ __forceinline__ float DOGV(const DogCtx &c, int l, int r, int x)
{
    const size_t p = (size_t)r * c.w + x;
    return c.d[(l + 1) * c.n + p] - c.d[l * c.n + p];
}

__device__ static bool sift_adjust(const DogCtx &c, int &layer, int &r, int &x, float &xi_, float &xr_, float &xc_, float &contr_)
{
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0, l = layer, rr = r, cc = x;
    for (; i < 5; ++i) {
        float dD0 = (DOGV(c, l, rr, cc + 1) - DOGV(c, l, rr, cc - 1)) * deriv_scale;
        float dD1 = (DOGV(c, l, rr + 1, cc) - DOGV(c, l, rr - 1, cc)) * deriv_scale;
        float dD2 = (DOGV(c, l + 1, rr, cc) - DOGV(c, l - 1, rr, cc)) * deriv_scale;
        float v2 = DOGV(c, l, rr, cc) * 2;
        float dxx = (DOGV(c, l, rr, cc + 1) + DOGV(c, l, rr, cc - 1) - v2) * second;
        float dyy = (DOGV(c, l, rr + 1, cc) + DOGV(c, l, rr - 1, cc) - v2) * second;
        float dss = (DOGV(c, l + 1, rr, cc) + DOGV(c, l - 1, rr, cc) - v2) * second;
        float dxy = (DOGV(c, l, rr + 1, cc + 1) - DOGV(c, l, rr + 1, cc - 1) - DOGV(c, l, rr - 1, cc + 1) + DOGV(c, l, rr - 1, cc - 1)) * cross;
        float dxs = (DOGV(c, l + 1, rr, cc + 1) - DOGV(c, l + 1, rr, cc - 1) - DOGV(c, l - 1, rr, cc + 1) + DOGV(c, l - 1, rr, cc - 1)) * cross;
        float dys = (DOGV(c, l + 1, rr + 1, cc) - DOGV(c, l + 1, rr - 1, cc) - DOGV(c, l - 1, rr + 1, cc) + DOGV(c, l - 1, rr - 1, cc)) * cross;
        float A[3][4] = {{dxx, dxy, dxs, dD0}, {dxy, dyy, dys, dD1}, {dxs, dys, dss, dD2}};
        bool ok = true;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            int piv = p;
#pragma unroll
            for (int q = p + 1; q < 3; ++q) if (fabsf(A[q][p]) > fabsf(A[piv][p])) piv = q;
            if (fabsf(A[piv][p]) < FLT_EPSILON) { ok = false; break; }
            if (piv != p) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { float t = A[p][q]; A[p][q] = A[piv][q]; A[piv][q] = t; }
            }
            float d = -1.f / A[p][p];
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                float al = A[q][p] * d;
#pragma unroll
                for (int s = p + 1; s < 4; ++s) A[q][s] += al * A[p][s];
            }
        }
        float X0 = 0, X1 = 0, X2 = 0;
        if (ok) {
            X2 = A[2][3] / A[2][2];
            X1 = (A[1][3] - A[1][2] * X2) / A[1][1];
            X0 = (A[0][3] - A[0][1] * X1 - A[0][2] * X2) / A[0][0];
        }
        xi = -X2; xr = -X1; xc = -X0;
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return false;
        cc += __float2int_rn(xc); rr += __float2int_rn(xr); l += __float2int_rn(xi);
        if (l < 1 || l > S_NOL || cc < S_BORDER || cc >= c.w - S_BORDER || rr < S_BORDER || rr >= c.h - S_BORDER) return false;
    }
    if (i >= 5) return false;
    {
        float dD0 = (DOGV(c, l, rr, cc + 1) - DOGV(c, l, rr, cc - 1)) * deriv_scale;
        float dD1 = (DOGV(c, l, rr + 1, cc) - DOGV(c, l, rr - 1, cc)) * deriv_scale;
        float dD2 = (DOGV(c, l + 1, rr, cc) - DOGV(c, l - 1, rr, cc)) * deriv_scale;
        float t = (dD0 * xc + dD1 * xr) + dD2 * xi;
        float contr = DOGV(c, l, rr, cc) * img_scale + t * 0.5f;
        if (fabsf(contr) * S_NOL < 0.04f) return false;
        float v2 = DOGV(c, l, rr, cc) * 2.f;
        float dxx = (DOGV(c, l, rr, cc + 1) + DOGV(c, l, rr, cc - 1) - v2) * second;
        float dyy = (DOGV(c, l, rr + 1, cc) + DOGV(c, l, rr - 1, cc) - v2) * second;
        float dxy = (DOGV(c, l, rr + 1, cc + 1) - DOGV(c, l, rr + 1, cc - 1) - DOGV(c, l, rr - 1, cc + 1) + DOGV(c, l, rr - 1, cc - 1)) * cross;
        float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * 10.f >= (10.f + 1) * (10.f + 1) * det) return false;
        contr_ = contr;
    }
    layer = l; r = rr; x = cc; xi_ = xi; xr_ = xr; xc_ = xc;
    return true;
}

// Refinement is split in two so that neither half idles lanes: adjustLocalExtrema is a serial
// per-seed computation (one LANE per seed; most seeds die here on the contrast / edge tests), the
// orientation histogram is a per-survivor reduction (one WAVE per survivor).  Survivors and raw
// keypoints are appended through integer atomics; their order is irrelevant (sorted afterwards).
#define SURV_W 8        // floats per survivor record: packed(o,l,r,c), xi, xr, xc, contr
__global__ __launch_bounds__(256) void sift_adjust_kernel(const float *__restrict__ gauss, SiftDev dv,
                                                           const unsigned *__restrict__ seeds, const int *__restrict__ nseeds,
                                                           float *__restrict__ surv, int *__restrict__ nsurv)
{
    // image = fastest grid dimension (see sift_orient_kernel: the survivors' atomicAdd(&nsurv[img]) spread over the images)
    const int sidx = blockIdx.y * 256 + threadIdx.x, img = blockIdx.x;
    if (sidx >= nseeds[img]) return;
    const unsigned sd = seeds[(long long)img * dv.seed_cap + sidx];
    const int o = sd >> 28;
    int l = (sd >> 26) & 3, r = (sd >> 13) & 0x1FFF, c = sd & 0x1FFF;
    const int w = dv.w[o], h = dv.h[o];
    DogCtx dc = {gauss + (long long)img * dv.gstride + dv.goff[o], (long long)w * h, w, h};
    float xi = 0, xr = 0, xc = 0, contr = 0;
    if (!sift_adjust(dc, l, r, c, xi, xr, xc, contr)) return;
    const int slot = atomicAdd(&nsurv[img], 1);               // <= nseeds <= seed_cap
    float *q = surv + ((long long)img * dv.seed_cap + slot) * SURV_W;
    q[0] = __int_as_float((int)(((unsigned)o << 28) | ((unsigned)l << 26) | ((unsigned)r << 13) | (unsigned)c));
    q[1] = xi; q[2] = xr; q[3] = xc; q[4] = contr;
}

// retainBest(nfeatures) keeps the strongest responses, and a keypoint's response (|contrast|) is known before its orientation
// is: only the sel_k = nfeatures * 5/4 + 256 strongest survivors (ties included) get an orientation histogram -- a quarter
// of the ~10 k survivors of a textured HD frame at nfeatures = 2048.  Exact as long as those produce >= nfeatures unique
// keypoints (every survivor yields one or more, except a histogram whose maximum is a plateau): then the nfeatures-th best
// response, the threshold of retainBest, lies inside the selected set and nothing below it could have been kept.  Otherwise
// sift_finalize_kernel raises RPE_OVF_SIFT_PREFILTER.  sel_k = 0 (no cap): every survivor is selected.
__device__ __forceinline__ unsigned s_float_key(float f);
__global__ __launch_bounds__(256) void sift_select_kernel(const float *__restrict__ surv, const int *__restrict__ nsurv, SiftDev dv, int sel_k,
                                                           unsigned *__restrict__ sel, int *__restrict__ nsel, int *__restrict__ nraw, int redo)
{
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_kk, s_cnt;
    const int img = blockIdx.x, tid = threadIdx.x;
    // second round (redo): only the images whose selected survivors did not fill the cap, this time with every survivor
    if (redo) { if (!nsel[4 * img + 2]) return; sel_k = 0; }
    const int n = min(nsurv[img], dv.seed_cap);
    const float *sv = surv + (long long)img * dv.seed_cap * SURV_W;
    unsigned *out = sel + (long long)img * dv.seed_cap;
    unsigned thr_key = 0;
    if (sel_k > 0 && n > sel_k) {
        unsigned prefix = 0, mask = 0;
        if (tid == 0) s_kk = sel_k;
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = 8 * pass;
            s_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                const unsigned key = s_float_key(fabsf(sv[(long long)i * SURV_W + 4]));
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int kk = s_kk, acc = 0, bin = 0;
                for (int b = 255; b >= 0; --b) { int c = (int)s_hist[b]; if (acc + c >= kk) { bin = b; break; } acc += c; }
                s_kk = kk - acc;
                s_prefix = prefix | ((unsigned)bin << shift);
            }
            __syncthreads();
            prefix = s_prefix;
            mask |= 255u << shift;
        }
        thr_key = prefix;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        if (s_float_key(fabsf(sv[(long long)i * SURV_W + 4])) >= thr_key) out[atomicAdd(&s_cnt, 1)] = (unsigned)i;   // order is irrelevant: the keypoints are sorted later
    __syncthreads();
    if (tid == 0) {
        nsel[4 * img] = s_cnt; nsel[4 * img + 1] = s_cnt < n ? 1 : 0;
        if (!redo) nsel[4 * img + 2] = 0; else nraw[img] = 0;
    }
}

// One wave per workgroup: survivors differ in window size (6 to 14+ batches of 64 samples), and a 4-wave workgroup
// holds its wave slots and LDS until its slowest wave is done.
#define SIFT_ORI_WPW 1
#define S_ORI_RMAX1 20      // radius + 1 <= 18: radius = round(4.5 * 1.6 * 2^((l + xi) / 3)), l <= 3, |xi| < 0.5 -> <= 17
__global__ __launch_bounds__(64 * SIFT_ORI_WPW) void sift_orient_kernel(const float *__restrict__ gauss, SiftDev dv,
                                                           const float *__restrict__ surv, const unsigned *__restrict__ sel, const int *__restrict__ nsel,
                                                           float *__restrict__ raw, int *__restrict__ nraw, unsigned *__restrict__ overflow, int redo)
{
    __shared__ float s_part[SIFT_ORI_WPW][S_BINS][8];
    __shared__ float s_hist[SIFT_ORI_WPW][S_BINS + 4];
    __shared__ float s_wtab[SIFT_ORI_WPW][S_ORI_RMAX1 * S_ORI_RMAX1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // image = fastest grid dimension: workgroups in flight together then belong to different images and their
    // atomicAdd(&nraw[img]) go to different addresses (with one image at a time 14 k returning atomics per image queued
    // on a single L2 line: 12 of this kernel's 20 ms)
    const int img = blockIdx.x, ns = nsel[4 * img];
    if (redo && !nsel[4 * img + 2]) return;
    // waves stride over the list of selected survivors (a grid sized for seed_cap would be millions of empty workgroups)
    for (int si = blockIdx.y * SIFT_ORI_WPW + wv; si < ns; si += gridDim.y * SIFT_ORI_WPW) {
    const int sidx = (int)sel[(long long)img * dv.seed_cap + si];
    const float *sq = surv + ((long long)img * dv.seed_cap + sidx) * SURV_W;
    const unsigned sd = (unsigned)__float_as_int(sq[0]);
    const int o = sd >> 28, l = (sd >> 26) & 3, r = (sd >> 13) & 0x1FFF, c = sd & 0x1FFF;
    const float xi = sq[1], xr = sq[2], xc = sq[3], contr = sq[4];
    const int w = dv.w[o], h = dv.h[o];
    const long long n = (long long)w * h;
    const float kx = ((float)c + xc) * (float)(1 << o), ky = ((float)r + xr) * (float)(1 << o);
    const int koct = o + (l << 8) + (__double2int_rn(((double)xi + 0.5) * 255) << 16);
    const float ksize = 1.6f * det_exp2f(((float)l + xi) / S_NOL) * (float)(1 << o) * 2;
    const float kresp = fabsf(contr);
    // ---- calcOrientationHist on gaussian level l of this octave
    const float scl_octv = ksize * 0.5f / (float)(1 << o);
    const int radius = __float2int_rn(4.5f * scl_octv);
    const float sigma = 1.5f * scl_octv;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    const float *img_l = gauss + (long long)img * dv.gstride + dv.goff[o] + (long long)l * n;
    // LDS accessed as LDS (ds_* instructions, in order within the wave); S_WAVE_SYNC orders the lanes' accesses for the
    // compiler.  (Through a volatile generic pointer every access became a flat_* instruction with system-scope cache
    // bits and a full wait: 16 serialized round trips per 64 samples.)
    float (*part)[8] = s_part[wv];
    for (int i = lane; i < S_BINS * 8; i += 64) (&part[0][0])[i] = 0.f;
    S_WAVE_SYNC();
    // Gaussian weights: exp((i*i + j*j) * expf_scale) is the same number for the up to 8 samples (+-i, +-j), (+-j, +-i),
    // and det_expf (22 dependent f64 operations) was 60 % of this kernel: evaluate it once per pair a <= b into an LDS
    // table.  Rows a and radius - a of the triangle together have radius + 2 entries, so the triangle is walked as a
    // ceil((radius+1)/2) x (radius+2) rectangle: 2 rounds of 64 lanes at radius 14 instead of 14 batches of samples.
    float *wtab = s_wtab[wv];
    const int r1 = radius + 1;
    if (r1 <= S_ORI_RMAX1) {
        const int wdt = radius + 2, nent = ((r1 + 1) >> 1) * wdt;
        for (int t = lane; t < nent; t += 64) {
            const int q = t / wdt, e = t - q * wdt;
            const bool first = e < r1 - q;
            const int a = first ? q : radius - q, b = first ? q + e : radius - q + (e - (r1 - q));
            if (first || q < radius - q) wtab[a * r1 + b] = det_expf((float)(a * a + b * b) * expf_scale);
        }
    }
    S_WAVE_SYNC();
    const int side = 2 * radius + 1, nsamp = side * side;
    const float inv_side = 1.f / (float)side;
    constexpr int DU = 4;                                   // batches whose gradient loads fly together
    const int kl = ((lane & 7) << 3) | (lane >> 3);         // sample of a 64-batch held by this lane: slot k & 7 = lane >> 3
    for (int k0 = 0; k0 < nsamp; k0 += 64 * DU) {
        bool vld[DU];
        int di2[DU], wix[DU];
        float g0[DU], g1[DU], g2[DU], g3[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int k = k0 + 64 * u + kl;
            int qi = (int)((float)k * inv_side);
            int rem = k - qi * side;
            if (rem < 0) { --qi; rem += side; } else if (rem >= side) { ++qi; rem -= side; }
            const int i = qi - radius, j = rem - radius;
            const int y = r + i, x = c + j;
            di2[u] = i * i + j * j;
            { const int ai = i < 0 ? -i : i, aj = j < 0 ? -j : j; wix[u] = min(ai, aj) * r1 + max(ai, aj); }
            vld[u] = k < nsamp && !(y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1);
            g0[u] = g1[u] = g2[u] = g3[u] = 0.f;
            if (vld[u]) {
                const float *pc = img_l + (size_t)y * w + x;
                g0[u] = pc[1]; g1[u] = pc[-1]; g2[u] = pc[-w]; g3[u] = pc[w];
            }
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            if (k0 + 64 * u >= nsamp) break;                // wave-uniform
            const bool valid = vld[u];
            int bin = 0; float contrib = 0.f;
            if (valid) {
                const float dx = g0[u] - g1[u];
                const float dy = g2[u] - g3[u];
                const float wgt = r1 <= S_ORI_RMAX1 ? wtab[wix[u]] : det_expf((float)di2[u] * expf_scale);
                const float ori = fast_atan2_deg(dy, dx);
                const float mag = sqrtf(dx * dx + dy * dy);
                bin = __float2int_rn((S_BINS / 360.f) * ori);
                if (bin >= S_BINS) bin -= S_BINS;
                if (bin < 0) bin += S_BINS;
                contrib = wgt * mag;
            }
            // Slot p = lane >> 3 receives its 8 samples of the batch in ascending k = ascending lane.  Each lane builds the
            // value its accumulator has after its own sample: the accumulator as the previous batch left it, plus the
            // contributions of the lanes before it in its group of 8 that hit the same bin (taken in lane order from
            // the neighbours by DPP row shifts), plus its own.  The 8 masked stores then go out in lane order, so the
            // last sample of a bin leaves the final value.  One LDS round trip per batch instead of 8 dependent ones;
            // the additions and their order are those of the sequential loop.
            {
                const int binv = valid ? bin : -1;
                float acc = valid ? part[bin][lane >> 3] : 0.f;
#define S_ORI_STEP(s) { const int bs = __builtin_amdgcn_update_dpp(-2, binv, 0x110 + (s), 0xF, 0xF, false); \
                        const float cs = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(contrib), 0x110 + (s), 0xF, 0xF, false)); \
                        if (bs == binv && (lane & 7) >= (s)) acc = acc + cs; }
                S_ORI_STEP(7) S_ORI_STEP(6) S_ORI_STEP(5) S_ORI_STEP(4) S_ORI_STEP(3) S_ORI_STEP(2) S_ORI_STEP(1)
#undef S_ORI_STEP
                acc = acc + contrib;
#pragma unroll
                for (int rd = 0; rd < 8; ++rd) {
                    if ((lane & 7) == rd && valid) part[bin][lane >> 3] = acc;
                    S_WAVE_SYNC();
                }
            }
        }
    }
    float *th = s_hist[wv] + 2;
    if (lane < S_BINS) {
        float p0 = part[lane][0], p1 = part[lane][1], p2 = part[lane][2], p3 = part[lane][3];
        float p4 = part[lane][4], p5 = part[lane][5], p6 = part[lane][6], p7 = part[lane][7];
        p0 = p0 + p4; p1 = p1 + p5; p2 = p2 + p6; p3 = p3 + p7;
        p0 = p0 + p2; p1 = p1 + p3;
        th[lane] = p0 + p1;
    }
    S_WAVE_SYNC();
    if (lane == 0) { th[-1] = th[S_BINS - 1]; th[-2] = th[S_BINS - 2]; th[S_BINS] = th[0]; th[S_BINS + 1] = th[1]; }
    S_WAVE_SYNC();
    float hv = -1.f;
    if (lane < S_BINS)
        hv = (th[lane - 2] + th[lane + 2]) * (1.f / 16.f) + (th[lane - 1] + th[lane + 1]) * (4.f / 16.f) + th[lane] * (6.f / 16.f);
    float omax = hv;
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) omax = fmaxf(omax, __shfl_xor(omax, of));
    const float mag_thr = omax * 0.8f;
    const float hl = __shfl(hv, lane > 0 ? lane - 1 : S_BINS - 1), hr = __shfl(hv, lane < S_BINS - 1 ? lane + 1 : 0);
    if (lane < S_BINS && hv > hl && hv > hr && hv >= mag_thr) {
        float bin = (float)lane + 0.5f * (hl - hr) / (hl - 2 * hv + hr);
        bin = bin < 0 ? S_BINS + bin : bin >= S_BINS ? bin - S_BINS : bin;
        float angle = 360.f - (360.f / S_BINS) * bin;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        const int slot = atomicAdd(&nraw[img], 1);
        if (slot < dv.raw_cap) {
            float *q = raw + ((long long)img * dv.raw_cap + slot) * 6;
            q[0] = kx; q[1] = ky; q[2] = ksize; q[3] = angle; q[4] = kresp; q[5] = __int_as_float(koct);
        } else atomicOr(&overflow[img], (unsigned)RPE_OVF_SIFT_RAW);
    }
    S_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------ sort
// KeyPoint_LessThan as a 128-bit key: (x, y) ascending, size descending, angle ascending.
// Exact duplicates (same seed end point) get identical keys and become neighbours.
__device__ __forceinline__ unsigned s_float_key(float f)
{
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Response prefilter: retainBest(nfeatures) only ever keeps the strongest keypoints, so only the
// K = 2*nfeatures + 1024 strongest raw entries (ties included) go through the sort.  Equivalent
// to OpenCV's dedup -> retainBest as long as the top K hold >= nfeatures unique keypoints
// (exact duplicates are rare); otherwise the overflow flag is raised.
__global__ __launch_bounds__(256) void sift_prefilter_kernel(const float *__restrict__ raw, const int *__restrict__ nraw, SiftDev dv, int pad,
                                                              unsigned long long *__restrict__ k0, unsigned long long *__restrict__ k1,
                                                              unsigned *__restrict__ sidx, int *__restrict__ ncand, unsigned *__restrict__ overflow,
                                                              const int *__restrict__ nsel, int redo)
{
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_kk, s_cnt;
    const int img = blockIdx.x, tid = threadIdx.x;
    if (redo && !nsel[4 * img + 2]) return;
    const int n = min(nraw[img], dv.raw_cap);
    const float *rw = raw + (long long)img * dv.raw_cap * 6;
    const int K = 2 * dv.nfeatures + 1024;
    unsigned thr_key = 0;
    if (dv.nfeatures > 0 && n > K) {
        unsigned prefix = 0, mask = 0;
        if (tid == 0) s_kk = K;
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = 8 * pass;
            s_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                unsigned key = s_float_key(rw[(long long)i * 6 + 4]);
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int kk = s_kk, acc = 0, bin = 0;
                for (int b = 255; b >= 0; --b) { int c = (int)s_hist[b]; if (acc + c >= kk) { bin = b; break; } acc += c; }
                s_kk = kk - acc;
                s_prefix = prefix | ((unsigned)bin << shift);
            }
            __syncthreads();
            prefix = s_prefix;
            mask |= 255u << shift;
        }
        thr_key = prefix;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    unsigned long long *a0 = k0 + (long long)img * pad, *a1 = k1 + (long long)img * pad;
    unsigned *ix = sidx + (long long)img * pad;
    for (int i = tid; i < n; i += 256) {
        const float *q = rw + (long long)i * 6;
        if (s_float_key(q[4]) >= thr_key) {
            const int slot = atomicAdd(&s_cnt, 1);               // order is irrelevant: the list is sorted next
            if (slot < pad) {
                a0[slot] = ((unsigned long long)__float_as_uint(q[0]) << 32) | __float_as_uint(q[1]);   // x, y > 0: bit order = value order
                a1[slot] = ((unsigned long long)(~__float_as_uint(q[2])) << 32) | __float_as_uint(q[3]);
                ix[slot] = (unsigned)i;
            }
        }
    }
    __syncthreads();
    if (tid == 0) { ncand[img] = min(s_cnt, pad); if (s_cnt > pad) atomicOr(&overflow[img], (unsigned)RPE_OVF_SIFT_PREFILTER); }
}

// Bitonic sort of (k0, k1, index) triples, one workgroup per image.  Stages whose partner distance is below SORT_C run on a
// block of SORT_C entries held in LDS (40 KB: all of k = 2 .. SORT_C in one visit, then one visit per larger k); only the
// stages with partner distance >= SORT_C go through global memory (6 of the 105 stages at 16 384 entries: through global memory
// every stage was a dependent L2 round trip -- 0.74 ms per 256-image launch, 1.0 ms of the single uncapped pair's call).
#define SORT_C 2048
__global__ __launch_bounds__(1024) void sift_sort_kernel(const int *__restrict__ ncand, int pad,
                                                          unsigned long long *__restrict__ k0, unsigned long long *__restrict__ k1,
                                                          unsigned *__restrict__ sidx, const int *__restrict__ nsel, int redo)
{
    __shared__ unsigned long long s0[SORT_C], s1[SORT_C];
    __shared__ unsigned si[SORT_C];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (redo && !nsel[4 * img + 2]) return;
    const int n = ncand[img];
    int P = 64;
    while (P < n) P <<= 1;
    unsigned long long *a0 = k0 + (long long)img * pad, *a1 = k1 + (long long)img * pad;
    unsigned *ix = sidx + (long long)img * pad;
    for (int i = n + tid; i < P; i += 1024) { a0[i] = ~0ull; a1[i] = ~0ull; ix[i] = 0xFFFFFFFFu; }
    __syncthreads();
    const int C = P < SORT_C ? P : SORT_C;
    // one visit of every block: stages (k, j) for k in [k_lo, k_hi], j from min(k / 2, C / 2) down to 1
    auto lds_visit = [&](int k_lo, int k_hi) {
        for (int base = 0; base < P; base += C) {
            for (int i = tid; i < C; i += 1024) { s0[i] = a0[base + i]; s1[i] = a1[base + i]; si[i] = ix[base + i]; }
            __syncthreads();
            for (int k = k_lo; k <= k_hi; k <<= 1)
                for (int j = min(k >> 1, C >> 1); j > 0; j >>= 1) {
                    for (int t = tid; t < (C >> 1); t += 1024) {
                        const int i = 2 * j * (t / j) + (t % j), ixj = i + j;
                        const bool asc = ((base + i) & k) == 0;
                        const unsigned long long x0 = s0[i], x1 = s1[i], y0 = s0[ixj], y1 = s1[ixj];
                        const unsigned xi = si[i], yi = si[ixj];
                        const bool gt = x0 > y0 || (x0 == y0 && (x1 > y1 || (x1 == y1 && xi > yi)));
                        if (gt == asc) { s0[i] = y0; s1[i] = y1; si[i] = yi; s0[ixj] = x0; s1[ixj] = x1; si[ixj] = xi; }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < C; i += 1024) { a0[base + i] = s0[i]; a1[base + i] = s1[i]; ix[base + i] = si[i]; }
            __syncthreads();
        }
    };
    lds_visit(2, C);
    for (int k = 2 * C; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int i = 2 * j * (t / j) + (t % j), ixj = i + j;
                const bool asc = (i & k) == 0;
                const unsigned long long x0 = a0[i], x1 = a1[i], y0 = a0[ixj], y1 = a1[ixj];
                const unsigned xi = ix[i], yi = ix[ixj];
                const bool gt = x0 > y0 || (x0 == y0 && (x1 > y1 || (x1 == y1 && xi > yi)));
                if (gt == asc) { a0[i] = y0; a1[i] = y1; ix[i] = yi; a0[ixj] = x0; a1[ixj] = x1; ix[ixj] = xi; }
            }
            __syncthreads();
        }
        lds_visit(k, k);
    }
}

// ---------------------------------------------------------------- finalize
__global__ __launch_bounds__(256) void sift_finalize_kernel(const float *__restrict__ raw, const int *__restrict__ ncand, SiftDev dv, int pad,
                                                             const unsigned long long *__restrict__ k0, const unsigned long long *__restrict__ k1,
                                                             const unsigned *__restrict__ sidx, int *__restrict__ nsel, int redo, float *__restrict__ fin,
                                                             float2 *__restrict__ kp_pt, int *__restrict__ kp_count, unsigned *__restrict__ overflow)
{
    __shared__ unsigned s_hist[256];
    __shared__ int s_wave[5];
    __shared__ unsigned s_prefix;
    __shared__ int s_kk, s_nuniq;
    const int img = blockIdx.x, tid = threadIdx.x;
    if (redo && !nsel[4 * img + 2]) return;
    const int n = ncand[img];
    const unsigned long long *a0 = k0 + (long long)img * pad, *a1 = k1 + (long long)img * pad;
    const unsigned *ix = sidx + (long long)img * pad;
    const float *rw = raw + (long long)img * dv.raw_cap * 6;
    auto uniq = [&](int i) { return i == 0 || a0[i] != a0[i - 1] || a1[i] != a1[i - 1]; };
    // count unique keypoints (removeDuplicatedSorted)
    if (tid == 0) s_nuniq = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < n; i += 256) cnt += uniq(i) ? 1 : 0;
    atomicAdd(&s_nuniq, cnt);
    __syncthreads();
    const int nuniq = s_nuniq;
    const bool cut = nsel[4 * img + 1] != 0;             // sift_select_kernel left weaker survivors without an orientation
    if (cut && nuniq < dv.nfeatures) {
        // the selected survivors did not fill the cap: keypoints of the ones left out belong to the result.  Nothing is
        // written; the second round orients every survivor of this image and comes back here with cut = 0
        if (tid == 0) nsel[4 * img + 2] = 1;
        return;
    }
    unsigned thr_key = 0;
    if (dv.nfeatures > 0 && nuniq > dv.nfeatures) {       // retainBest: response >= the nfeatures-th best
        unsigned prefix = 0, mask = 0;
        if (tid == 0) s_kk = dv.nfeatures;
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = 8 * pass;
            s_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 256)
                if (uniq(i)) {
                    unsigned key = s_float_key(rw[(long long)ix[i] * 6 + 4]);
                    if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255], 1u);
                }
            __syncthreads();
            if (tid == 0) {
                int kk = s_kk, acc = 0, bin = 0;
                for (int b = 255; b >= 0; --b) { int c = (int)s_hist[b]; if (acc + c >= kk) { bin = b; break; } acc += c; }
                s_kk = kk - acc;
                s_prefix = prefix | ((unsigned)bin << shift);
            }
            __syncthreads();
            prefix = s_prefix;
            mask |= 255u << shift;
        }
        thr_key = prefix;
    }
    int offset = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        bool keep = false;
        if (i < n && uniq(i)) keep = s_float_key(rw[(long long)ix[i] * 6 + 4]) >= thr_key;
        int total;
        const int ex = s_block_excl_scan(keep ? 1 : 0, s_wave, total);
        if (keep) {
            const int o = offset + ex;
            if (o < dv.kcap) {
                const float *q = rw + (long long)ix[i] * 6;
                float *f = fin + ((long long)img * dv.kcap + o) * 6;
                f[0] = q[0]; f[1] = q[1]; f[2] = q[2]; f[3] = q[3]; f[4] = q[4]; f[5] = q[5];
                kp_pt[(long long)img * dv.kcap + o] = make_float2(q[0] * 0.5f, q[1] * 0.5f);   // firstOctave = -1
            }
        }
        offset += total;
    }
    if (tid == 0) {
        kp_count[img] = min(offset, dv.kcap);
        // the cap removed keypoints: the reference's SIFT_create() is uncapped (pose_estimator.py:93-94), so its
        // feature set is larger than this one
        if (dv.nfeatures > 0 && (nuniq > dv.nfeatures || cut)) atomicOr(&overflow[img], (unsigned)RPE_OVF_SIFT_CAP);
        if (offset > dv.kcap) atomicOr(&overflow[img], (unsigned)RPE_OVF_SIFT_KEYPOINTS);
    }
}

// ---------------------------------------------------------------- descriptor
// One keypoint per 64-lane workgroup: the 15 KB of accumulators per keypoint decide the occupancy (four keypoints per
// workgroup were 61 KB: 2 workgroups = 8 waves per CU; one per workgroup gives 10).
#define SIFT_DESC_KPW 1
#define S_DESC_ROWS 128       // rows of the sample square: 2 radius + 1 <= 77 (radius = round(3 scl sqrt2 2.5), scl <= 1.6 * 2^(3.5/3))
__global__ __launch_bounds__(64 * SIFT_DESC_KPW) void sift_describe_kernel(const float *__restrict__ gauss, SiftDev dv, const float *__restrict__ fin,
                                                             const int *__restrict__ kp_count, uint8_t *__restrict__ desc)
{
    // accumulators of the 4 x 4 inner cells only: the border cells of calcSIFTDescriptor's 6 x 6 x 10 histogram are never read
    // (each (bin, slot) accumulator is independent, so leaving them out changes nothing that is), and 5 KB instead of 11.5 KB
    // per keypoint decide how many workgroups a CU holds
    __shared__ float s_part[SIFT_DESC_KPW][160][8];
    __shared__ float s_hist[SIFT_DESC_KPW][160];
    __shared__ float s_stage[SIFT_DESC_KPW][64 * 9];
    __shared__ int s_ja[SIFT_DESC_KPW][S_DESC_ROWS], s_pre[SIFT_DESC_KPW][S_DESC_ROWS + 1];   // per raster row: first valid column, samples before the row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kidx = blockIdx.x * SIFT_DESC_KPW + wv, img = blockIdx.y;
    if (kidx >= kp_count[img]) return;
    const float *f = fin + ((long long)img * dv.kcap + kidx) * 6;
    const int koct = __float_as_int(f[5]);
    const int o = koct & 255, l = (koct >> 8) & 255;
    const float scale = 1.f / (float)(1 << o);
    const float size = f[2] * scale;
    float angle = 360.f - f[3];
    if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
    const float ptx = f[0] * scale, pty = f[1] * scale, ori = angle, scl = size * 0.5f;
    const int w = dv.w[o], h = dv.h[o];
    const float *img_l = gauss + (long long)img * dv.gstride + dv.goff[o] + (long long)l * w * h;
    const int d = 4, n = 8;
    const int px = __float2int_rn(ptx), py = __float2int_rn(pty);
    double sn, cs;
    det_sincos((double)(ori * (float)(3.141592653589793238462643383279502884 / 180.0)), sn, cs);
    float cos_t = (float)cs, sin_t = (float)sn;
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
    int radius = __float2int_rn(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt(((double)w) * w + ((double)h) * h);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    float (*part)[8] = s_part[wv];
    float *stg = s_stage[wv];
    for (int i = lane; i < 160 * 8; i += 64) (&part[0][0])[i] = 0.f;
    S_WAVE_SYNC();
    const int side = 2 * radius + 1;
    // calcSIFTDescriptor's first loop keeps the positions (i, j) of the bounding square that lie inside the rotated 4 x 4 grid
    // and off the image border, in raster order -- about half of the square.  Each of its conditions is a float expression that
    // is monotone in j (products and sums round monotonically), so the positions kept in a row are an interval [ja, jb]:
    // one lane per row finds it (real-valued bounds, widened by one, then moved inward / outward with the exact predicate),
    // a wave scan numbers the samples, and the expensive part below runs on dense batches of 64 samples whose (i, j) come from
    // the row table.  Sample number k counts the samples that pass, as cv2's arrays do.
    constexpr int DU = 4;                                   // dense batches whose gradient loads fly together
    int *rja = s_ja[wv], *rpre = s_pre[wv];
    auto okpos = [&](int i, int j) -> bool {
        const float crot = j * cos_t - i * sin_t, rrot = j * sin_t + i * cos_t;
        const float rbn = rrot + d / 2 - 0.5f, cbn = crot + d / 2 - 0.5f;
        const int r = py + i, c = px + j;
        return rbn > -1 && rbn < d && cbn > -1 && cbn < d && r > 0 && r < h - 1 && c > 0 && c < w - 1;
    };
    int ntotal = 0;                                         // wave-uniform
    if (side > S_DESC_ROWS) return;                         // cannot happen (see S_DESC_ROWS); no descriptor rather than a wrong one
    for (int r0 = 0; r0 < side; r0 += 64) {
        const int row = r0 + lane, i = row - radius;
        int ja = 0, cnt = 0;
        if (row < side) {
            const int jmin = max(-radius, 1 - px), jmax = min(radius, w - 2 - px);
            // |j S + i C| < 2.5 and |j C - i S| < 2.5 solved for j where the coefficient is not tiny (an estimate only)
            float lo = (float)jmin, hi = (float)jmax;
            if (fabsf(sin_t) >= 1e-3f) {
                const float a = (-2.5f - i * cos_t) / sin_t, b = (2.5f - i * cos_t) / sin_t;
                lo = fmaxf(lo, fminf(a, b)); hi = fminf(hi, fmaxf(a, b));
            }
            if (fabsf(cos_t) >= 1e-3f) {
                const float a = (-2.5f + i * sin_t) / cos_t, b = (2.5f + i * sin_t) / cos_t;
                lo = fmaxf(lo, fminf(a, b)); hi = fminf(hi, fmaxf(a, b));
            }
            int jl = max(jmin, (int)floorf(lo) - 1), jh = min(jmax, (int)ceilf(hi) + 1);
            while (jl <= jh && !okpos(i, jl)) ++jl;         // inward to the first / last position that passes
            while (jh >= jl && !okpos(i, jh)) --jh;
            if (jl <= jh) {                                  // outward: the estimate is a superset unless rounding says otherwise
                while (jl > jmin && okpos(i, jl - 1)) --jl;
                while (jh < jmax && okpos(i, jh + 1)) ++jh;
                ja = jl; cnt = jh - jl + 1;
            }
        }
        int incl = cnt;                                      // inclusive scan over the 64 rows of this pass
#pragma unroll
        for (int of = 1; of < 64; of <<= 1) { const int t = __shfl_up(incl, of); if (lane >= of) incl += t; }
        if (row < side) { rja[row] = ja; rpre[row] = ntotal + incl - cnt; }
        ntotal += __shfl(incl, 63);
    }
    if (lane == 0) rpre[side] = ntotal;
    S_WAVE_SYNC();
    const int nb = (ntotal + 63) >> 6;
    int rstart = 0;                                         // row of the last sample of the previous batch (wave-uniform)
    {
        for (int b0 = 0; b0 < nb; b0 += DU) {
        bool vld[DU];
        float crot[DU], rrot[DU], rbn[DU], cbn[DU], g0[DU], g1[DU], g2[DU], g3[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = (b0 + u) * 64 + lane;
            vld[u] = b0 + u < nb && e < ntotal;
            g0[u] = g1[u] = g2[u] = g3[u] = 0.f; crot[u] = rrot[u] = rbn[u] = cbn[u] = 0.f;
            int row = rstart;
            if (vld[u]) {
                while (rpre[row + 1] <= e) ++row;           // a batch spans a few rows (empty ones included)
                const int i = row - radius, j = rja[row] + (e - rpre[row]);
                crot[u] = j * cos_t - i * sin_t; rrot[u] = j * sin_t + i * cos_t;
                rbn[u] = rrot[u] + d / 2 - 0.5f; cbn[u] = crot[u] + d / 2 - 0.5f;
                const float *pc = img_l + (size_t)(py + i) * w + (px + j);
                g0[u] = pc[1]; g1[u] = pc[-1]; g2[u] = pc[-w]; g3[u] = pc[w];
            }
            if (b0 + u < nb) rstart = __shfl(row, min(63, ntotal - 1 - (b0 + u) * 64));     // last valid lane of the batch
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
        if (b0 + u >= nb) break;                                          // wave-uniform
        const bool valid = vld[u];
        int idx = 0;
        float v000 = 0, v001 = 0, v010 = 0, v011 = 0, v100 = 0, v101 = 0, v110 = 0, v111 = 0;
        if (valid) {
            float rbin = rbn[u], cbin = cbn[u];
            const float c_rot = crot[u], r_rot = rrot[u];
            {
                const float dx = g0[u] - g1[u];
                const float dy = g2[u] - g3[u];
                const float wgt = det_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
                const float oo = fast_atan2_deg(dy, dx);
                const float mag = sqrtf(dx * dx + dy * dy) * wgt;
                float obin = (oo - ori) * bins_per_rad;
                const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
                int o0 = (int)floorf(obin);
                rbin -= r0; cbin -= c0; obin -= o0;
                if (o0 < 0) o0 += n;
                if (o0 >= n) o0 -= n;
                const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
                const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
                v111 = v_rc11 * obin; v110 = v_rc11 - v111; v101 = v_rc10 * obin; v100 = v_rc10 - v101;
                v011 = v_rc01 * obin; v010 = v_rc01 - v011; v001 = v_rc00 * obin; v000 = v_rc00 - v001;
                // compact address of corner (0, 0, 0) (may lie outside: + 64 keeps it positive) and which of the four
                // (row, column) corners fall on inner cells
                const int rr0 = r0 + 1, cc0 = c0 + 1;              // 0 .. 4 in the 6 x 6 grid
                const int mrow = (rr0 >= 1 ? 1 : 0) | (rr0 <= 3 ? 2 : 0), mcol = (cc0 >= 1 ? 1 : 0) | (cc0 <= 3 ? 2 : 0);
                const int mask = ((mrow & 1) && (mcol & 1) ? 1 : 0) | ((mrow & 1) && (mcol & 2) ? 2 : 0) | ((mrow & 2) && (mcol & 1) ? 4 : 0) | ((mrow & 2) && (mcol & 2) ? 8 : 0);
                idx = (((rr0 - 1) * 4 + (cc0 - 1)) * (n + 2) + o0 + 64) | (mask << 16);
            }
        }
        // Accumulation: sample k adds its 8 trilinear corners to slot (k & 7) of 8 distinct bins, and every
        // (bin, slot) accumulator must see its samples in ascending k.  Lane (s, q) = (slot, corner) walks the
        // 8 samples of this batch that use slot s in ascending order: lanes differ in slot or in bin, so the 64
        // read-modify-writes of a step never collide and the per-accumulator order is the sequential one.
        stg[lane * 9] = valid ? __int_as_float(idx) : __int_as_float(-1);
        stg[lane * 9 + 1] = v000; stg[lane * 9 + 2] = v001; stg[lane * 9 + 3] = v010; stg[lane * 9 + 4] = v011;
        stg[lane * 9 + 5] = v100; stg[lane * 9 + 6] = v101; stg[lane * 9 + 7] = v110; stg[lane * 9 + 8] = v111;
        S_WAVE_SYNC();
        {
            const int sl = lane & 7, q = lane >> 3;
            const int qoff = (q & 1) + ((q >> 1) & 1) * (n + 2) + (q >> 2) * d * (n + 2) - 64;
            // the 16 staged values of this lane's 8 steps are read in one go (they do not depend on the accumulators);
            // the steps themselves stay sequential: corners of different samples share bins
            int ids[8]; float vs[8];
#pragma unroll
            for (int st = 0; st < 8; ++st) { ids[st] = __float_as_int(stg[(sl + 8 * st) * 9]); vs[st] = stg[(sl + 8 * st) * 9 + 1 + q]; }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (ids[st] >= 0 && ((ids[st] >> (16 + (q >> 1))) & 1)) {
                    const int ad = (ids[st] & 0xFFFF) + qoff;
                    part[ad][sl] = part[ad][sl] + vs[st];
                }
                S_WAVE_SYNC();
            }
        }
        }
        }
    }
    S_WAVE_SYNC();
    float *hist = s_hist[wv];
    for (int b = lane; b < 160; b += 64) {
        float p0 = part[b][0], p1 = part[b][1], p2 = part[b][2], p3 = part[b][3], p4 = part[b][4], p5 = part[b][5], p6 = part[b][6], p7 = part[b][7];
        p0 = p0 + p4; p1 = p1 + p5; p2 = p2 + p6; p3 = p3 + p7;
        p0 = p0 + p2; p1 = p1 + p3;
        hist[b] = p0 + p1;
    }
    S_WAVE_SYNC();
    // circular orientation bins, then element e = (i*d + j)*n + k ; lane holds e = lane and lane + 64
    float dv0, dv1;
    {
        auto elem = [&](int e) {
            const int ij = e >> 3, kb = e & 7;
            const int id = ij * (n + 2);
            float v = hist[id + kb];
            if (kb < 2) v = v + hist[id + n + kb];
            return v;
        };
        dv0 = elem(lane); dv1 = elem(lane + 64);
    }
    auto tree64 = [&](float v) {
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { float u = __shfl_down(v, of); if (lane < of) v = v + u; }
        return __shfl(v, 0);
    };
    float thr = sqrtf(tree64(dv0 * dv0 + dv1 * dv1)) * 0.2f;
    dv0 = dv0 < thr ? dv0 : thr; dv1 = dv1 < thr ? dv1 : thr;
    const float nrm = sqrtf(tree64(dv0 * dv0 + dv1 * dv1));
    const float fct = 512.f / (nrm > FLT_EPSILON ? nrm : FLT_EPSILON);
    int q0 = __float2int_rn(dv0 * fct), q1 = __float2int_rn(dv1 * fct);
    q0 = q0 < 0 ? 0 : q0 > 255 ? 255 : q0; q1 = q1 < 0 ? 0 : q1 > 255 ? 255 : q1;
    uint8_t *dst = desc + ((long long)img * dv.kcap + kidx) * 128;
    dst[lane] = (uint8_t)q0; dst[lane + 64] = (uint8_t)q1;
}

// ================================================================== host side
static int s_round_d(double v) { return (int)lrint(v); }

static int sift_gauss_kernel(double sigma, float *k)
{
    int ks = s_round_d(sigma * 8 + 1) | 1;
    double sum = 0, t[64];
    for (int i = 0; i < ks; ++i) { double x = i - (ks - 1) * 0.5; t[i] = exp(-0.5 * x * x / (sigma * sigma)); sum += t[i]; }
    for (int i = 0; i < ks; ++i) k[i] = (float)(t[i] / sum);
    return ks;
}

#define SCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); return RPE_ERR_HIP; } } while (0)

int rpe_sift_create(rpe_handle *h)
{
    RpeSiftState *S = new RpeSiftState();
    h->sift = S;
    SiftDev &dv = S->dv;
    const int W = h->cfg.width, H = h->cfg.height;
    const int bw = 2 * W, bh = 2 * H, mn = bw < bh ? bw : bh;
    dv.noct = s_round_d(log((double)mn) / log(2.) - 2) + 1;
    if (dv.noct > 12) dv.noct = 12;
    long long go = 0, dof = 0;
    for (int o = 0; o < dv.noct; ++o) {
        dv.w[o] = o ? dv.w[o - 1] / 2 : bw; dv.h[o] = o ? dv.h[o - 1] / 2 : bh;
        dv.goff[o] = go; go += (long long)S_NG * dv.w[o] * dv.h[o];
        dv.doff[o] = dof; dof += (long long)S_ND * dv.w[o] * dv.h[o];
    }
    dv.gstride = go; dv.dstride = dof; dv.tstride = (long long)bw * bh;
    dv.nfeatures = h->cfg.nfeatures;
    dv.kcap = h->lay.kcap;
    // seeds: 1/16 of the base-image pixels (>= 16384); the oracle uses the same bound
    dv.seed_cap = (int)(((long long)bw * bh) / 16); if (dv.seed_cap < 16384) dv.seed_cap = 16384;
    dv.raw_cap = dv.seed_cap;
    float kern[6][32]; int ks[6];
    memset(kern, 0, sizeof(kern));
    const double sigma = 1.6, kk = pow(2., 1. / S_NOL);
    float sd = sqrtf(fmaxf((float)(sigma * sigma) - 0.5f * 0.5f * 4, 0.01f));
    ks[0] = sift_gauss_kernel((double)sd, kern[0]);
    for (int i = 1; i < S_NG; ++i) {
        double sp = pow(kk, (double)(i - 1)) * sigma, st = sp * kk;
        ks[i] = sift_gauss_kernel(sqrt(st * st - sp * sp), kern[i]);
        if (ks[i] > 31) { h->err = "SIFT kernel too wide"; return RPE_ERR_INVALID; }
    }
    SCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_skern), kern, sizeof(kern)));
    SCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_sks), ks, sizeof(ks)));
    for (int i = 0; i < 6; ++i) S->ks[i] = ks[i];
    // extrema scan: bands (one per row of every (octave, inner layer)) in the oracle's enumeration order, the hit
    // mask layout and the tile list of the tiled first pass
    std::vector<SiftXTile> xt;
    {
        int nb = 0; long long bo = 0;
        for (int o = 0; o < 12; ++o) { dv.band0[o] = 0; dv.wpr[o] = 0; dv.bmoff[o] = 0; }
        for (int o = 0; o < dv.noct; ++o) {
            dv.band0[o] = nb; dv.wpr[o] = (dv.w[o] + 63) / 64; dv.bmoff[o] = bo;
            S->xtile_oct_end[o] = (int)xt.size();
            if (dv.w[o] <= 2 * S_BORDER || dv.h[o] <= 2 * S_BORDER) continue;
            nb += S_NOL * (dv.h[o] - 2 * S_BORDER);
            bo += (long long)S_NOL * dv.h[o] * dv.wpr[o];
            for (int y = 0; y < dv.h[o] - S_BORDER; y += SX_TH)
                for (int x = 0; x < dv.w[o] - S_BORDER; x += SX_TW)
                    if (y + SX_TH > S_BORDER && x + SX_TW > S_BORDER) xt.push_back({o, x, y});
            S->xtile_oct_end[o] = (int)xt.size();
        }
        dv.nbands = nb > 0 ? nb : 1; dv.bmstride = bo > 0 ? bo : 1;
    }
    const size_t NI = (size_t)h->n_img_cap;
    // sort capacity: the K = 2*nfeatures + 1024 candidates the response prefilter lets through (+ ties), or, without a
    // cap (nfeatures = 0: nothing is filtered), twice the keypoint capacity (an oriented keypoint list is ~1.2x its unique ones)
    S->raw_pad = 16384;
    { const int need = dv.nfeatures > 0 ? 2 * dv.nfeatures + 1024 + 2048 : 2 * dv.kcap; while (S->raw_pad < need) S->raw_pad <<= 1; }
    S->n_xtiles = (int)xt.size();
    SCHK(hipMalloc(&S->d_xtiles, sizeof(SiftXTile) * (xt.size() ? xt.size() : 1)));
    if (!xt.empty()) SCHK(hipMemcpy(S->d_xtiles, xt.data(), sizeof(SiftXTile) * xt.size(), hipMemcpyHostToDevice));
    SCHK(hipMalloc(&S->d_xmask, sizeof(unsigned long long) * NI * dv.bmstride));
    SCHK(hipMalloc(&S->d_gauss, sizeof(float) * NI * dv.gstride));
    {   // scratch of the unfused fallbacks (upsampled image + row-pass image, 66 MB per HD image): only when a tap count has
        // no fused instantiation, which the reference's SIFT parameters (sigma 1.6, 3 layers: radii 5 5 6 8 10 13) never produce
        bool need_tmp = (ks[0] >> 1) != 5;
        for (int i = 0; i < S_NG; ++i) { const int r = ks[i] >> 1; if (r != 5 && r != 6 && r != 8 && r != 10 && r != 13) need_tmp = true; }
        if (need_tmp) SCHK(hipMalloc(&S->d_tmp, sizeof(float) * NI * dv.tstride * 2));
        S->fused_all = !need_tmp;
        S->march = getenv("RPE_SIFT_MARCH") && (ks[1] >> 1) == 5 && (ks[2] >> 1) == 6 && (ks[3] >> 1) == 8 && (ks[4] >> 1) == 10 && (ks[5] >> 1) == 13;
        if (const char *e = getenv("RPE_SIFT_GROUP")) S->group = atoi(e);
        if (const char *e = getenv("RPE_SIFT_GROUP_OCTAVES")) S->group_octaves = atoi(e) > 0 ? atoi(e) : 1;
    }
    SCHK(hipMalloc(&S->d_band_cnt, sizeof(int) * NI * dv.nbands));
    SCHK(hipMalloc(&S->d_band_off, sizeof(int) * NI * dv.nbands));
    SCHK(hipMalloc(&S->d_seeds, sizeof(unsigned) * NI * dv.seed_cap));
    SCHK(hipMalloc(&S->d_nseeds, sizeof(int) * NI));
    SCHK(hipMalloc(&S->d_raw, sizeof(float) * NI * dv.raw_cap * 6));
    SCHK(hipMalloc(&S->d_nraw, sizeof(int) * NI));
    SCHK(hipMalloc(&S->d_surv, sizeof(float) * NI * dv.seed_cap * SURV_W));
    SCHK(hipMalloc(&S->d_nsurv, sizeof(int) * NI));
    SCHK(hipMalloc(&S->d_sel, sizeof(unsigned) * NI * dv.seed_cap));
    SCHK(hipMalloc(&S->d_nsel, sizeof(int) * NI * 4));
    if (const char *e = getenv("RPE_SIFT_SEL_K")) S->sel_k_override = atoi(e);      // tests: a small value forces the second round
    SCHK(hipMalloc(&S->d_overflow, sizeof(int) * NI));
    SCHK(hipMalloc(&S->d_ncand, sizeof(int) * NI));
    SCHK(hipMalloc(&S->d_k0, sizeof(unsigned long long) * NI * S->raw_pad));
    SCHK(hipMalloc(&S->d_k1, sizeof(unsigned long long) * NI * S->raw_pad));
    SCHK(hipMalloc(&S->d_sidx, sizeof(unsigned) * NI * S->raw_pad));
    SCHK(hipMalloc(&S->d_fin, sizeof(float) * NI * dv.kcap * 6));
    return RPE_OK;
}

void rpe_sift_destroy(rpe_handle *h)
{
    RpeSiftState *S = h->sift;
    if (!S) return;
    void *p[] = {S->d_gauss, S->d_dog, S->d_tmp, S->d_xtiles, S->d_xmask, S->d_band_cnt, S->d_band_off, S->d_seeds, S->d_nseeds, S->d_raw,
                 S->d_nraw, S->d_overflow, S->d_ncand, S->d_k0, S->d_k1, S->d_sidx, S->d_fin, S->d_surv, S->d_nsurv, S->d_sel, S->d_nsel};
    for (void *q : p) if (q) hipFree(q);
    delete S;
    h->sift = nullptr;
}

template <int R, bool UPS = false>
static void sift_blur_launch(rpe_handle *h, const float *src, long long sstride, float *dst, long long dstride, float *dog,
                             long long dogstride, int w, int hh, int kid, int n_img, const uint8_t *u8a = nullptr,
                             const uint8_t *u8b = nullptr, int na = 0, float *dec = nullptr, int w2 = 0, int h2 = 0)
{
    // tile height: 32 rows everywhere measured best while the kernel was issue-limited; RPE_SIFT_TH64=<min radius> (diagnostic)
    // gives the radii from that one on 64-row tiles (less halo per output row, half the workgroups per CU)
    static const int th64_from = getenv("RPE_SIFT_TH64") ? atoi(getenv("RPE_SIFT_TH64")) : 1000;
    const int tcols = (w + 63) / 64;
    if (R >= th64_from) {
        constexpr int TH = 64;
        const int ntiles = tcols * ((hh + TH - 1) / TH);
        hipLaunchKernelGGL((sift_blur_fused_kernel<R, TH, UPS>), dim3((ntiles + 7) / 8 * 8, n_img), dim3(256), 0, h->stream, src, sstride, dst, dstride,
                           w, hh, kid, tcols, ntiles, u8a, u8b, na, w / 2, hh / 2, dec, w2, h2);
    } else {
        constexpr int TH = 32;
        const int ntiles = tcols * ((hh + TH - 1) / TH);
        hipLaunchKernelGGL((sift_blur_fused_kernel<R, TH, UPS>), dim3((ntiles + 7) / 8 * 8, n_img), dim3(256), 0, h->stream, src, sstride, dst, dstride,
                           w, hh, kid, tcols, ntiles, u8a, u8b, na, w / 2, hh / 2, dec, w2, h2);
    }
    if (dog)      // DoG planes are not stored by the product path (layers are formed where they are consumed); kept for callers that ask
        hipLaunchKernelGGL(sift_sub_kernel, dim3((unsigned)(((long long)w * hh + 255) / 256), 1, n_img), dim3(256), 0, h->stream,
                           (const float *)dst, dstride, src, sstride, dog, dogstride, (long long)w * hh);
}

// G[dst] = gauss(kid) * G[src]; dog (optional) = G[dst] - G[src]; dec (optional, same per-image stride as dst) = dst at even
// x, even y.  Returns true when dec was written (fused instantiations only; the caller runs sift_halve_kernel otherwise).
static bool sift_blur(rpe_handle *h, const float *src, long long sstride, float *dst, long long dstride, float *dog, long long dogstride,
                      int w, int hh, int kid, int n_img, float *dec = nullptr, int w2 = 0, int h2 = 0)
{
    switch (h->sift->ks[kid] >> 1) {
    case 5:  sift_blur_launch<5>(h, src, sstride, dst, dstride, dog, dogstride, w, hh, kid, n_img, nullptr, nullptr, 0, dec, w2, h2); return dec != nullptr;
    case 6:  sift_blur_launch<6>(h, src, sstride, dst, dstride, dog, dogstride, w, hh, kid, n_img, nullptr, nullptr, 0, dec, w2, h2); return dec != nullptr;
    case 8:  sift_blur_launch<8>(h, src, sstride, dst, dstride, dog, dogstride, w, hh, kid, n_img, nullptr, nullptr, 0, dec, w2, h2); return dec != nullptr;
    case 10: sift_blur_launch<10>(h, src, sstride, dst, dstride, dog, dogstride, w, hh, kid, n_img, nullptr, nullptr, 0, dec, w2, h2); return dec != nullptr;
    case 13: sift_blur_launch<13>(h, src, sstride, dst, dstride, dog, dogstride, w, hh, kid, n_img, nullptr, nullptr, 0, dec, w2, h2); return dec != nullptr;
    default: {      // any other width: unfused two-pass path
        float *tmp = h->sift->d_tmp + (long long)h->n_img_cap * h->sift->dv.tstride;
        hipLaunchKernelGGL(sift_blur_row_kernel, dim3((w + 255) / 256, hh, n_img), dim3(256), 0, h->stream, src, sstride, tmp, h->sift->dv.tstride, w, hh, kid);
        hipLaunchKernelGGL(sift_blur_col_kernel, dim3((w + 255) / 256, hh, n_img), dim3(256), 0, h->stream, (const float *)tmp, h->sift->dv.tstride, dst, dstride, w, hh, kid);
        if (dog) hipLaunchKernelGGL(sift_sub_kernel, dim3((unsigned)(((long long)w * hh + 255) / 256), 1, n_img), dim3(256), 0, h->stream,
                                    (const float *)dst, dstride, src, sstride, dog, dogstride, (long long)w * hh);
    } }
    return false;
}

// d_imgs: n_img tightly packed u8 images already resident (d_a followed by d_b as in the ORB path)
int rpe_sift_run(rpe_handle *h, const uint8_t *d_a, const uint8_t *d_b, int na, int nb)
{
    RpeSiftState *S = h->sift;
    const SiftDev &dv = S->dv;
    const int W = h->cfg.width, H = h->cfg.height, bw = 2 * W, bh = 2 * H;
    const size_t img = (size_t)W * H;
    // stage events (rpe_get_stage_ms slots reused for SIFT): PYRAMID = upsample + Gaussian pyramid + DoG, FAST = extrema
    // scan, SELECT = adjustLocalExtrema, HARRIS = orientation histograms, KEYPOINTS = prefilter + sort + retainBest,
    // DESCRIBE = descriptors; NMS / ANGLE / BLUR are empty
    MARK(h, RPE_STAGE_PYRAMID);
    const int n = na + nb;
    // Pyramid of images [i0, i0 + g), octaves [o0, o1).  Octave 0 starts with the upsample + initial blur in one kernel (the
    // upsampled image is formed in the blur's window loader; the separate upsample kernel only feeds the unfused fallback
    // for an unexpected tap count); level 0 of octave o + 1 (= level S_NOL of octave o, halved) is written by the blur that
    // makes that level.
    auto pyramid = [&](int i0, int g, int o0, int o1) {
        float *G = S->d_gauss + (long long)i0 * dv.gstride;
        bool have_l0 = true;
        if (o0 == 0) {
            if ((S->ks[0] >> 1) == 5) {
                const int na_g = na - i0 < 0 ? 0 : na - i0 > g ? g : na - i0;
                sift_blur_launch<5, true>(h, nullptr, 0, G + dv.goff[0], dv.gstride, nullptr, 0, bw, bh, 0, g,
                                          d_a + (size_t)i0 * img, d_b + (size_t)(i0 > na ? i0 - na : 0) * img, na_g);
            } else {
                for (int part = 0; part < 2; ++part) {
                    const uint8_t *src = part ? d_b : d_a; const int cnt = part ? nb : na, first = part ? na : 0;
                    if (!cnt) continue;
                    hipLaunchKernelGGL(sift_upsample_kernel, dim3((bw / 4 + 256) / 256, H + 1, cnt), dim3(256), 0, h->stream, src, W, H, img,
                                       S->d_tmp + (long long)first * dv.tstride, dv.tstride);
                }
                sift_blur(h, S->d_tmp, dv.tstride, S->d_gauss + dv.goff[0], dv.gstride, nullptr, 0, bw, bh, 0, n);
            }
        }
        for (int o = o0; o < o1; ++o) {
            const int w = dv.w[o], hh = dv.h[o];
            const long long pn = (long long)w * hh;
            if (!have_l0)
                hipLaunchKernelGGL(sift_halve_kernel, dim3((w + 1023) / 1024, hh, g), dim3(256), 0, h->stream,
                                   (const float *)(G + dv.goff[o - 1] + (long long)S_NOL * dv.w[o - 1] * dv.h[o - 1]),
                                   G + dv.goff[o], dv.gstride, dv.w[o - 1], w, hh);
            have_l0 = false;
            if (S->march && w >= 256 && hh >= 64) {
                // levels 1-3 in one march over level 0 (+ level 0 of the next octave), levels 4-5 in one march over level 3
                const bool last = o + 1 >= dv.noct;
                float *dec = !last ? G + dv.goff[o + 1] : nullptr;
                const dim3 grid((w + MARCH_SW - 1) / MARCH_SW, g);
                hipLaunchKernelGGL((sift_march_kernel<3, 5, 6, 8>), grid, dim3(MarchGeo<3, 5, 6, 8>::NT), 0, h->stream, (const float *)(G + dv.goff[o]), dv.gstride,
                                   G + dv.goff[o] + pn, dv.gstride, pn, w, hh, 1, dec, 2, last ? 0 : dv.w[o + 1], last ? 0 : dv.h[o + 1]);
                hipLaunchKernelGGL((sift_march_kernel<2, 10, 13, 0>), grid, dim3(MarchGeo<2, 10, 13, 0>::NT), 0, h->stream, (const float *)(G + dv.goff[o] + 3 * pn), dv.gstride,
                                   G + dv.goff[o] + 4 * pn, dv.gstride, pn, w, hh, 4, (float *)nullptr, -1, 0, 0);
                have_l0 = !last;
                continue;
            }
            for (int i = 1; i < S_NG; ++i) {
                const bool last = o + 1 >= dv.noct;
                float *dec = i == S_NOL && !last ? G + dv.goff[o + 1] : nullptr;
                const bool wrote = sift_blur(h, G + dv.goff[o] + (i - 1) * pn, dv.gstride, G + dv.goff[o] + i * pn, dv.gstride,
                                             nullptr, 0, w, hh, i, g, dec, last ? 0 : dv.w[o + 1], last ? 0 : dv.h[o + 1]);
                if (i == S_NOL) have_l0 = wrote;
            }
        }
    };
    // extrema scan of tiles [t0, t0 + nt) of images [i0, i0 + g)
    auto extrema = [&](int i0, int g, int t0, int nt) {
        if (nt > 0)
            hipLaunchKernelGGL(sift_extrema_mask_kernel, dim3((nt + 7) / 8 * 8, g), dim3(256), 0, h->stream,
                               (const float *)(S->d_gauss + (long long)i0 * dv.gstride), dv, (const SiftXTile *)(S->d_xtiles + t0),
                               S->d_xmask + (long long)i0 * dv.bmstride, S->d_band_cnt + (long long)i0 * dv.nbands, nt);
    };
    hipMemsetAsync(S->d_band_cnt, 0, sizeof(int) * (size_t)n * dv.nbands, h->stream);
    if (S->group > 0 && S->fused_all && dv.noct > 1) {
        // image-major schedule for the large octaves: the six levels of octave 0 of a few images (199 MB each at 1920x1080) are
        // written, read by the next blur and scanned for extrema while they are still in the 256 MB memory-side cache
        const int om = S->group_octaves < dv.noct ? S->group_octaves : dv.noct;
        for (int i0 = 0; i0 < n; i0 += S->group) {
            const int g = n - i0 < S->group ? n - i0 : S->group;
            pyramid(i0, g, 0, om);
            extrema(i0, g, 0, S->xtile_oct_end[om - 1]);
        }
        if (om < dv.noct) pyramid(0, n, om, dv.noct);
        MARK(h, RPE_STAGE_FAST);
        extrema(0, n, S->xtile_oct_end[om - 1], S->n_xtiles - S->xtile_oct_end[om - 1]);
    } else {
        pyramid(0, n, 0, dv.noct);
        // 3. seeds (count, scan, emit)
        MARK(h, RPE_STAGE_FAST);
        extrema(0, n, 0, S->n_xtiles);
    }
    hipMemsetAsync(h->d_ovf, 0, sizeof(unsigned) * n, h->stream);
    hipLaunchKernelGGL(sift_band_scan_kernel, dim3(n), dim3(256), 0, h->stream, (const int *)S->d_band_cnt, S->d_band_off, S->d_nseeds, h->d_ovf, dv);
    hipLaunchKernelGGL(sift_extrema_emit_kernel, dim3((dv.nbands + 3) / 4, n), dim3(256), 0, h->stream, (const unsigned long long *)S->d_xmask, dv,
                       (const int *)S->d_band_cnt, (const int *)S->d_band_off, S->d_seeds);
    // 4. refine + orientation -> raw keypoints
    hipMemsetAsync(S->d_nraw, 0, sizeof(int) * n, h->stream);
    MARK(h, RPE_STAGE_NMS); MARK(h, RPE_STAGE_SELECT);
    hipMemsetAsync(S->d_nsurv, 0, sizeof(int) * n, h->stream);
    hipLaunchKernelGGL(sift_adjust_kernel, dim3(n, (dv.seed_cap + 255) / 256), dim3(256), 0, h->stream, (const float *)S->d_gauss, dv,
                       (const unsigned *)S->d_seeds, (const int *)S->d_nseeds, S->d_surv, S->d_nsurv);
    MARK(h, RPE_STAGE_HARRIS);
    // 5. orientation of the strongest survivors, sort, dedup, retainBest, compaction.  Round 1 only does something for images
    //    whose selected survivors did not fill the cap (sift_finalize_kernel marked them): every survivor is oriented then.
    const int sel_k = dv.nfeatures > 0 ? (S->sel_k_override > 0 ? S->sel_k_override : dv.nfeatures + dv.nfeatures / 4 + 256) : 0;
    for (int redo = 0; redo < (sel_k > 0 ? 2 : 1); ++redo) {
        hipLaunchKernelGGL(sift_select_kernel, dim3(n), dim3(256), 0, h->stream, (const float *)S->d_surv, (const int *)S->d_nsurv, dv,
                           sel_k, S->d_sel, S->d_nsel, S->d_nraw, redo);
        hipLaunchKernelGGL(sift_orient_kernel, dim3(n, (redo ? 64 : 8192) / SIFT_ORI_WPW), dim3(64 * SIFT_ORI_WPW), 0, h->stream, (const float *)S->d_gauss, dv,
                           (const float *)S->d_surv, (const unsigned *)S->d_sel, (const int *)S->d_nsel, S->d_raw, S->d_nraw, h->d_ovf, redo);
        if (!redo) MARK(h, RPE_STAGE_KEYPOINTS);
        hipLaunchKernelGGL(sift_prefilter_kernel, dim3(n), dim3(256), 0, h->stream, (const float *)S->d_raw, (const int *)S->d_nraw, dv, S->raw_pad,
                           S->d_k0, S->d_k1, S->d_sidx, S->d_ncand, h->d_ovf, (const int *)S->d_nsel, redo);
        hipLaunchKernelGGL(sift_sort_kernel, dim3(n), dim3(1024), 0, h->stream, (const int *)S->d_ncand, S->raw_pad, S->d_k0, S->d_k1, S->d_sidx,
                           (const int *)S->d_nsel, redo);
        hipLaunchKernelGGL(sift_finalize_kernel, dim3(n), dim3(256), 0, h->stream, (const float *)S->d_raw, (const int *)S->d_ncand, dv, S->raw_pad,
                           (const unsigned long long *)S->d_k0, (const unsigned long long *)S->d_k1, (const unsigned *)S->d_sidx, S->d_nsel, redo,
                           S->d_fin, h->d_kp_pt, h->d_kp_count, h->d_ovf);
    }
    // 6. descriptors
    MARK(h, RPE_STAGE_ANGLE); MARK(h, RPE_STAGE_BLUR); MARK(h, RPE_STAGE_DESCRIBE);
    hipLaunchKernelGGL(sift_describe_kernel, dim3((dv.kcap + SIFT_DESC_KPW - 1) / SIFT_DESC_KPW, n), dim3(64 * SIFT_DESC_KPW), 0, h->stream, (const float *)S->d_gauss, dv,
                       (const float *)S->d_fin, (const int *)h->d_kp_count, h->d_desc);
    SCHK(hipGetLastError());
    return RPE_OK;
}

// keypoint records of the last run (stage API / tests): fin rows x,y,size,angle,response,octave(bits) un-halved
int rpe_sift_fetch(rpe_handle *h, int n_images, float *fin_host, int *counts)
{
    RpeSiftState *S = h->sift;
    SCHK(hipMemcpyAsync(fin_host, S->d_fin, sizeof(float) * 6 * (size_t)n_images * S->dv.kcap, hipMemcpyDeviceToHost, h->stream));
    SCHK(hipMemcpyAsync(counts, h->d_kp_count, sizeof(int) * n_images, hipMemcpyDeviceToHost, h->stream));
    SCHK(hipStreamSynchronize(h->stream));
    return RPE_OK;
}

int rpe_sift_fetch_gauss(rpe_handle *h, int index, float *out)
{
    RpeSiftState *S = h->sift;
    SCHK(hipMemcpyAsync(out, S->d_gauss + (long long)index * S->dv.gstride, sizeof(float) * S->dv.gstride, hipMemcpyDeviceToHost, h->stream));
    SCHK(hipStreamSynchronize(h->stream));
    return RPE_OK;
}
long long rpe_sift_gauss_floats(rpe_handle *h) { return h->sift ? h->sift->dv.gstride : 0; }
