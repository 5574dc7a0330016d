// match_kernels.hip -- brute-force Hamming matcher with OpenCV crossCheck semantics.
//
// Replaces cv2.BFMatcher(NORM_HAMMING, crossCheck=True).match(desc1, desc2)
// + sorted(key=distance) + [:max_matches]   (reference src/core/pose_estimator.py:131,:144-151)
// and the point gather of :518-519 (fused epilogue).
//
// One workgroup (256 lanes) per image pair.  Each lane owns one train descriptor
// (8 dwords in VGPRs); query descriptors are staged through LDS in 1024-row tiles
// with coalesced 16-B loads and read back as wave-uniform (broadcast) ds_read_b128;
// distance = 8 x (v_xor_b32 + v_bcnt_u32_b32).  Selection follows
// core/batch_distance.cpp (crosscheck=true) of OpenCV >= 4.5.x, two passes: every train elects its nearest query
// (strict '<', ascending query => lowest index on ties) and every query keeps its best elector through a packed
// (dist<<18 | trainIdx) LDS atomicMin (lowest train on ties); then every query finds its OWN nearest train (same
// key, lowest train on ties) and the match survives only if that train elected the query -- the two keys are
// equal exactly then.  (Rounds 1-2 ran the first pass only, the rule of older OpenCV: a superset.  The reference's
// result rows single out the two-pass rule: tests/test_reference_rows_cpu.py.)
// The stable sort by distance is a bitonic sort of (dist<<16 | queryIdx) keys in LDS.
//
// RATIO = true is the opt-in extension named by the project brief and absent from the reference (which uses
// crossCheck, pose_estimator.py:131): cv2's knnMatch(k=2) + Lowe's ratio test.  Roles swap: a lane owns a QUERY
// descriptor, the TRAIN descriptors stream through LDS; the lane keeps its best (distance, lowest train index) and
// second-best distance and emits the match iff  best < ratio * second  (compared in f64, as Python compares
// m.distance < ratio * n.distance); queries with fewer than two candidates emit nothing.  Sort / truncation /
// point gather are shared.
#include "rpe_internal.h"
#include <stdlib.h>
#include <algorithm>

#define QTILE 1024

__device__ __forceinline__ int ham256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    int d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y); d += __popc(a0.z ^ b0.z); d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x); d += __popc(a1.y ^ b1.y); d += __popc(a1.z ^ b1.z); d += __popc(a1.w ^ b1.w);
    return d;
}

template <bool RATIO>
__global__ __launch_bounds__(256) void match_hamming_kernel(const uint8_t *__restrict__ desc, const int *__restrict__ kp_count,
                                                             const float2 *__restrict__ kp_pt, int img2_base, int kcap,
                                                             int max_matches, double ratio,
                                                             int *__restrict__ m_q, int *__restrict__ m_t, int *__restrict__ m_d,
                                                             int *__restrict__ m_n, float2 *__restrict__ pts1, float2 *__restrict__ pts2)
{
    extern __shared__ uint4 s_dyn[];
    uint4 *s_q = s_dyn;                                   // QTILE*2 uint4 = 32 KB (later: sort keys)
    unsigned *s_best = (unsigned *)(s_dyn + QTILE * 2);   // kcap entries
    unsigned *s_row = s_best + kcap;                      // kcap entries: the query's own nearest train (second crossCheck pass)
    __shared__ int s_valid;
    const int tid = threadIdx.x, pair = blockIdx.x;
    const int img1 = pair, img2 = img2_base + pair;
    const int n1 = min(kp_count[img1], kcap), n2 = min(kp_count[img2], kcap);
    for (int i = tid; i < n1; i += 256) { s_best[i] = 0xFFFFFFFFu; if (!RATIO) s_row[i] = 0xFFFFFFFEu; }      // ratio mode: no s_row (and no LDS for it)
    if (tid == 0) s_valid = 0;
    const uint4 *d1 = (const uint4 *)(desc + (long long)img1 * kcap * 32);
    const uint4 *d2 = (const uint4 *)(desc + (long long)img2 * kcap * 32);
    // owner descriptors (one per lane, in registers) x scanned descriptors (through LDS):
    // crossCheck pass 0: owner = train j, scanned = queries; pass 1 and ratio: owner = query i, scanned = trains
#pragma unroll 1
    for (int pass = 0; pass < (RATIO ? 1 : 2); ++pass) {
    const bool qown = RATIO || pass == 1;
    const uint4 *d_own = qown ? d1 : d2, *d_scan = qown ? d2 : d1;
    const int n_own = qown ? n1 : n2, n_scan = qown ? n2 : n1;
    for (int tc = 0; tc < n_own; tc += 256) {
        const int j = tc + tid;
        const bool valid = j < n_own;
        uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
        if (valid) { t0 = d_own[2 * j]; t1 = d_own[2 * j + 1]; }
        int bestd = 0x7FFFFFFF, besti = 0, second = 0x7FFFFFFF;
        for (int qt = 0; qt < n_scan; qt += QTILE) {
            const int nq = min(QTILE, n_scan - qt);
            __syncthreads();
            for (int idx = tid; idx < nq * 2; idx += 256) s_q[idx] = d_scan[2 * qt + idx];
            __syncthreads();
            int i = 0;
            if (!RATIO) {
                for (; i + 4 <= nq; i += 4) {
                    int da = ham256(s_q[2 * i], s_q[2 * i + 1], t0, t1);
                    int db = ham256(s_q[2 * i + 2], s_q[2 * i + 3], t0, t1);
                    int dc = ham256(s_q[2 * i + 4], s_q[2 * i + 5], t0, t1);
                    int dd = ham256(s_q[2 * i + 6], s_q[2 * i + 7], t0, t1);
                    if (da < bestd) { bestd = da; besti = qt + i; }
                    if (db < bestd) { bestd = db; besti = qt + i + 1; }
                    if (dc < bestd) { bestd = dc; besti = qt + i + 2; }
                    if (dd < bestd) { bestd = dd; besti = qt + i + 3; }
                }
                for (; i < nq; ++i) {
                    int da = ham256(s_q[2 * i], s_q[2 * i + 1], t0, t1);
                    if (da < bestd) { bestd = da; besti = qt + i; }
                }
            } else {
                for (; i < nq; ++i) {
                    const int da = ham256(s_q[2 * i], s_q[2 * i + 1], t0, t1);
                    if (da < bestd) { second = bestd; bestd = da; besti = qt + i; }
                    else if (da < second) second = da;
                }
            }
        }
        if (!RATIO) {
            if (valid && n_scan > 0) {
                if (pass == 0) atomicMin(&s_best[besti], ((unsigned)bestd << 18) | (unsigned)j);
                else s_row[j] = ((unsigned)bestd << 18) | (unsigned)besti;   // owner j is the query, besti its nearest train
            }
        } else if (valid && n_scan >= 2 && (double)bestd < ratio * (double)second) {
            s_best[j] = ((unsigned)bestd << 18) | (unsigned)besti;          // owner j is the query, besti the train
        }
    }
    }
    __syncthreads();
    // (dist, queryIdx) keys; unmatched queries sort to the end
    int sortP = 64;
    while (sortP < n1) sortP <<= 1;
    unsigned *s_key = (unsigned *)s_q;
    int myvalid = 0;
    for (int i = tid; i < sortP; i += 256) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n1) {
            unsigned b = s_best[i];
            if (b != 0xFFFFFFFFu && (RATIO || b == s_row[i])) { key = ((b >> 18) << 16) | (unsigned)i; ++myvalid; }
        }
        s_key[i] = key;
    }
    if (myvalid) atomicAdd(&s_valid, myvalid);
    __syncthreads();
    for (int k = 2; k <= sortP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (sortP >> 1); t += 256) {
                int i = 2 * j * (t / j) + (t % j);
                int ixj = i + j;
                bool asc = (i & k) == 0;
                unsigned a = s_key[i], b = s_key[ixj];
                if ((a > b) == asc) { s_key[i] = b; s_key[ixj] = a; }
            }
            __syncthreads();
        }
    }
    const int nm = min(s_valid, max_matches);
    for (int r = tid; r < nm; r += 256) {
        unsigned key = s_key[r];
        int i = key & 0xFFFF, d = key >> 16;
        int j = s_best[i] & 0x3FFFF;
        long long o = (long long)pair * max_matches + r;
        m_q[o] = i; m_t[o] = j; m_d[o] = d;
        pts1[o] = kp_pt[(long long)img1 * kcap + i];
        pts2[o] = kp_pt[(long long)img2 * kcap + j];
    }
    if (tid == 0) m_n[pair] = nm;
}

// ---------------------------------------------------------------- crossCheck on the matrix cores
// The measured bound of the VALU kernel above is vector-instruction issue: 19 instructions per 64 distances
// (8 v_xor + 8 v_bcnt + compare / select) at the 4-cycle integer issue cadence = 0.75 ms for 1024 pairs of
// 1000 x 1000 descriptors, 91 % of the measured issue roof (profiles/r02_counters.json, r02_calibration.json).
// Hamming distance is a dot product in disguise: d(q, t) = |q| + |t| - 2 q.t with the 256 bits as 0/1 bytes, so the
// O(N^2) part goes to v_mfma_i32_32x32x32_i8 (exact integers): 8 MFMAs = one 32 x 32 tile of q.t over K = 256.
// What is left for the vector ALU per tile is the O(N) bit -> byte expansion and a 2-instruction epilogue per
// accumulator: key = ((|q| + 512) << 16 | queryIdx) - (q.t << 17) (one v_mad_i32_i24), running v_min_u32 per train
// column -- the minimum of (distance, queryIdx) keys IS batchDistance's "strict <, ascending query" election.  The second
// crossCheck pass (the query's own nearest train) is the same loop with the two descriptor sets swapped.
// Workgroup = one pair, 8 waves; wave w owns train tile 8 p + w of pass p (its 32 descriptors expanded once into
// 32 VGPRs: the B operand of all 8 K-steps), the query tiles stream through LDS, expanded by all 512 threads
// (16 bits -> 16 bytes: nibble * 0x00204081 & 0x01010101), double buffered: one barrier per query tile.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
#define MM_NT 512
#define MM_PF 4                      // query tiles fetched ahead of the one being multiplied

__device__ __forceinline__ v4i_t expand16(unsigned b)
{
    v4i_t r;
    r.x = (int)(__umul24(b & 15u, 0x00204081u) & 0x01010101u);
    r.y = (int)(__umul24((b >> 4) & 15u, 0x00204081u) & 0x01010101u);
    r.z = (int)(__umul24((b >> 8) & 15u, 0x00204081u) & 0x01010101u);
    r.w = (int)(__umul24((b >> 12) & 15u, 0x00204081u) & 0x01010101u);
    return r;
}

// SPLIT = true: small batches (the drop-in's estimate() is a batch of ONE pair: a single workgroup walked 2 x 127 x 127 tiles
// alone, 2.3 ms of a 2.9 ms call).  The rounds of 8 owner tiles are dealt over gridDim.y workgroups per pair, the election
// words live in HBM (integer atomicMin: order independent), and match_hamming_select_kernel sorts and emits afterwards.
template <bool SPLIT>
__global__ __launch_bounds__(MM_NT) void match_hamming_mfma_kernel(const uint8_t *__restrict__ desc, const int *__restrict__ kp_count,
                                                                    const float2 *__restrict__ kp_pt, int img2_base, int kcap,
                                                                    int max_matches, int region0,
                                                                    unsigned *__restrict__ g_best, unsigned *__restrict__ g_row,
                                                                    int *__restrict__ m_q, int *__restrict__ m_t, int *__restrict__ m_d,
                                                                    int *__restrict__ m_n, float2 *__restrict__ pts1, float2 *__restrict__ pts2)
{
    extern __shared__ uint4 s_dyn[];
    // [0, region0 x 16 B): two expanded scanned tiles (2 x 8 K-steps x 64 lanes x 16 B = 16 KB) + their packed (|x| + 512, index)
    // words + the popcounts; reused as the sort-key array afterwards.  Then kcap election words and kcap own-nearest words.
    v4i_t *s_a = (v4i_t *)s_dyn;                               // [2][8][64]
    unsigned *s_qpk = (unsigned *)(s_dyn + 2 * 8 * 64);        // [2][32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pair = blockIdx.x;
    unsigned *s_best = SPLIT ? g_best + (long long)pair * kcap : (unsigned *)(s_dyn + region0);          // kcap entries
    unsigned *s_row = SPLIT ? g_row + (long long)pair * kcap : s_best + kcap;     // kcap entries: the query's own nearest train (second crossCheck pass)
    __shared__ int s_valid;
    const int img1 = pair, img2 = img2_base + pair;
    const int n1 = min(kp_count[img1], kcap), n2 = min(kp_count[img2], kcap);
    if (!SPLIT) for (int i = tid; i < n1; i += MM_NT) { s_best[i] = 0xFFFFFFFFu; s_row[i] = 0xFFFFFFFEu; }      // SPLIT: the host memsets them
    if (tid == 0) s_valid = 0;
    // |x| + 512 of every scanned descriptor, once per pass (the passes over the scanned tiles all need them); kept in the free
    // part of the first 32 KB (kcap <= 8064 entries)
    unsigned short *s_qpop = (unsigned short *)(s_dyn + 2 * 8 * 64 + 16);
    const int h = lane >> 5, col = lane & 31;
    // this thread's share of a scanned tile's expansion: item = tid: K-step s = tid >> 6, lane slot l = tid & 63
    // (row = l & 31, half = l >> 5): the 16 bits [32 s + 16 half, +16) of scanned descriptor (tile * 32 + row)
    const int xs = tid >> 6, xl = tid & 63, xrow = xl & 31, xh = xl >> 5;
    // pass 0: the TRAINS own the MFMA columns and elect their nearest query (scanned through LDS);
    // pass 1: the QUERIES own the columns and find their own nearest train -- the same loop with the roles swapped
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
    const int n_own = pass ? n1 : n2, n_scan = pass ? n2 : n1;
    const unsigned *q32 = (const unsigned *)(desc + (long long)(pass ? img2 : img1) * kcap * 32);      // scanned: 8 dwords per descriptor
    const uint4 *d2 = (const uint4 *)(desc + (long long)(pass ? img1 : img2) * kcap * 32);             // owners
    __syncthreads();                                           // the previous pass is done with s_qpop
    for (int i = tid; i < n_scan; i += MM_NT) {
        const uint4 a = ((const uint4 *)q32)[2 * i], b = ((const uint4 *)q32)[2 * i + 1];
        s_qpop[i] = (unsigned short)(512u + __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w));
    }
    const int ntq = (n_scan + 31) >> 5, ntt = (n_own + 31) >> 5;
    for (int tt0 = SPLIT ? 8 * (int)blockIdx.y : 0; tt0 < ntt && n_scan > 0; tt0 += SPLIT ? 8 * (int)gridDim.y : 8) {
        const int tt = tt0 + wv;                               // wave-uniform
        const int j = tt * 32 + col;
        const bool valid_t = tt < ntt && j < n_own;
        // B operand of the 8 K-steps: this lane's own descriptor, bits [32 s + 16 h, +16) of step s
        v4i_t bop[8];
        int tpop = 0;
        {
            uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
            if (valid_t) { t0 = d2[2 * j]; t1 = d2[2 * j + 1]; }
            const unsigned tw[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int sK = 0; sK < 8; ++sK) { bop[sK] = expand16((tw[sK] >> (16 * h)) & 0xFFFFu); tpop += __popc(tw[sK]); }
        }
        unsigned best_key = 0xFFFFFFFFu;
        // The raw scanned words are fetched MM_PF tiles ahead into a register ring: with a one-tile lookahead every step of
        // the loop waited for an L2 round trip (diagnostic build: the loop without its MFMAs took 0.17 of the kernel's
        // 0.35 ms -- 128 steps of 1.3 us).  The loop is unrolled by MM_PF so that the ring is indexed statically.
        unsigned ring[MM_PF];
        auto fetch = [&](int qt) -> unsigned {
            const int q = qt * 32 + xrow;
            return (qt < ntq && q < n_scan) ? q32[(long long)q * 8 + xs] : 0u;
        };
        auto stage = [&](int qt, int buf, unsigned raw) {
            s_a[(buf * 8 + xs) * 64 + xl] = expand16((raw >> (16 * xh)) & 0xFFFFu);
            if (tid < 32) {
                const int qq = qt * 32 + tid;
                const unsigned pk = qq < n_scan ? (unsigned)s_qpop[qq] : 0x7000u;      // rows past the end: a key no real distance can beat
                s_qpk[buf * 32 + tid] = (pk << 16) | (unsigned)qq;
            }
        };
        __syncthreads();                                       // the previous round has finished reading both buffers (and s_qpop is complete)
#pragma unroll
        for (int u = 0; u < MM_PF; ++u) ring[u] = fetch(u);    // tiles 0 .. MM_PF - 1 in flight
        stage(0, 0, ring[0]);
        ring[0] = fetch(MM_PF);
        __syncthreads();
        for (int qt0 = 0; qt0 < ntq; qt0 += MM_PF) {
#pragma unroll
            for (int u = 0; u < MM_PF; ++u) {
                const int qt = qt0 + u;
                if (qt < ntq) {                                // workgroup-uniform
                    const int buf = qt & 1;
                    if (tt < ntt) {
                        v16i_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int sK = 0; sK < 8; ++sK)
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_a[(buf * 8 + sK) * 64 + lane], bop[sK], acc, 0, 0, 0);
                        // C layout (dtype independent): column = lane & 31, row of register r = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned qpk = s_qpk[buf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                            best_key = min(best_key, (unsigned)__mul24(acc[r], -131072) + qpk);     // (|x| + 512 - 2 x.own) << 16 | scannedIdx
                        }
                    }
                    if (qt + 1 < ntq) {
                        // tile qt + 1 sits in ring slot (u + 1) % MM_PF (tile 0 of this round was staged in the prologue)
                        stage(qt + 1, buf ^ 1, ring[(u + 1) % MM_PF]);
                        ring[(u + 1) % MM_PF] = fetch(qt + 1 + MM_PF);
                    }
                    __syncthreads();
                }
            }
        }
        best_key = min(best_key, (unsigned)__shfl_xor((int)best_key, 32));
        if (valid_t && h == 0) {
            const unsigned d = (best_key >> 16) - 512u + (unsigned)tpop, i = best_key & 0xFFFFu;
            if (pass == 0) atomicMin(&s_best[i], (d << 18) | (unsigned)j);      // train j elects query i
            else s_row[j] = (d << 18) | i;                                      // query j's own nearest train i
        }
    }
    }
    __syncthreads();
    if (SPLIT) return;
    // (dist, queryIdx) keys; unmatched queries sort to the end -- same epilogue as the VALU kernel
    int sortP = 64;
    while (sortP < n1) sortP <<= 1;
    unsigned *s_key = (unsigned *)s_dyn;
    int myvalid = 0;
    for (int i = tid; i < sortP; i += MM_NT) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n1) {
            unsigned b = s_best[i];
            if (b != 0xFFFFFFFFu && b == s_row[i]) { key = ((b >> 18) << 16) | (unsigned)i; ++myvalid; }
        }
        s_key[i] = key;
    }
    if (myvalid) atomicAdd(&s_valid, myvalid);
    __syncthreads();
    for (int k = 2; k <= sortP; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int t = tid; t < (sortP >> 1); t += MM_NT) {
                int i = 2 * jj * (t / jj) + (t % jj);
                int ixj = i + jj;
                bool asc = (i & k) == 0;
                unsigned a = s_key[i], b = s_key[ixj];
                if ((a > b) == asc) { s_key[i] = b; s_key[ixj] = a; }
            }
            __syncthreads();
        }
    }
    const int nm = min(s_valid, max_matches);
    for (int r = tid; r < nm; r += MM_NT) {
        unsigned key = s_key[r];
        int i = key & 0xFFFF, d = key >> 16;
        int j = s_best[i] & 0x3FFFF;
        long long o = (long long)pair * max_matches + r;
        m_q[o] = i; m_t[o] = j; m_d[o] = d;
        pts1[o] = kp_pt[(long long)img1 * kcap + i];
        pts2[o] = kp_pt[(long long)img2 * kcap + j];
    }
    if (tid == 0) m_n[pair] = nm;
}


// sort + top-max_matches + point gather of a SPLIT run (one workgroup per pair; the same steps as the fused epilogue)
__global__ __launch_bounds__(MM_NT) void match_hamming_select_kernel(const unsigned *__restrict__ g_best, const unsigned *__restrict__ g_row,
                                                                      const int *__restrict__ kp_count, const float2 *__restrict__ kp_pt,
                                                                      int img2_base, int kcap, int max_matches,
                                                                      int *__restrict__ m_q, int *__restrict__ m_t, int *__restrict__ m_d,
                                                                      int *__restrict__ m_n, float2 *__restrict__ pts1, float2 *__restrict__ pts2)
{
    extern __shared__ uint4 s_dyn[];
    __shared__ int s_valid;
    const int tid = threadIdx.x, pair = blockIdx.x;
    const int img1 = pair, img2 = img2_base + pair;
    const int n1 = min(kp_count[img1], kcap);
    const unsigned *s_best = g_best + (long long)pair * kcap, *s_row = g_row + (long long)pair * kcap;
    if (tid == 0) s_valid = 0;
    __syncthreads();
    // (dist, queryIdx) keys; unmatched queries sort to the end -- same epilogue as the VALU kernel
    int sortP = 64;
    while (sortP < n1) sortP <<= 1;
    unsigned *s_key = (unsigned *)s_dyn;
    int myvalid = 0;
    for (int i = tid; i < sortP; i += MM_NT) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n1) {
            unsigned b = s_best[i];
            if (b != 0xFFFFFFFFu && b == s_row[i]) { key = ((b >> 18) << 16) | (unsigned)i; ++myvalid; }
        }
        s_key[i] = key;
    }
    if (myvalid) atomicAdd(&s_valid, myvalid);
    __syncthreads();
    for (int k = 2; k <= sortP; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int t = tid; t < (sortP >> 1); t += MM_NT) {
                int i = 2 * jj * (t / jj) + (t % jj);
                int ixj = i + jj;
                bool asc = (i & k) == 0;
                unsigned a = s_key[i], b = s_key[ixj];
                if ((a > b) == asc) { s_key[i] = b; s_key[ixj] = a; }
            }
            __syncthreads();
        }
    }
    const int nm = min(s_valid, max_matches);
    for (int r = tid; r < nm; r += MM_NT) {
        unsigned key = s_key[r];
        int i = key & 0xFFFF, d = key >> 16;
        int j = s_best[i] & 0x3FFFF;
        long long o = (long long)pair * max_matches + r;
        m_q[o] = i; m_t[o] = j; m_d[o] = d;
        pts1[o] = kp_pt[(long long)img1 * kcap + i];
        pts2[o] = kp_pt[(long long)img2 * kcap + j];
    }
    if (tid == 0) m_n[pair] = nm;
}

void rpe_launch_match(rpe_handle *h, int B)
{
    const int kcap = h->lay.kcap;
    size_t lds = (size_t)QTILE * 32 + (size_t)kcap * 8;       // staging / sort keys + election words + own-nearest words
    if (h->cfg.match_mode == RPE_MATCH_RATIO)                 // the ratio mode has no own-nearest words: 4 bytes per keypoint (<= 64 KB at 8064)
        hipLaunchKernelGGL((match_hamming_kernel<true>), dim3(B), dim3(256), (size_t)QTILE * 32 + (size_t)kcap * 4, h->stream,
                           h->d_desc, h->d_kp_count, h->d_kp_pt, h->img2_base ? h->img2_base : B, kcap, h->cfg.max_matches, h->cfg.match_ratio,
                           h->d_m_q, h->d_m_t, h->d_m_d, h->d_m_n, h->d_pts1, h->d_pts2);
    else if (getenv("RPE_MATCH_VALU") && lds <= 65536)    // diagnostic: the vector-ALU crossCheck kernel (A/B runs, parity tests)
        hipLaunchKernelGGL((match_hamming_kernel<false>), dim3(B), dim3(256), lds, h->stream,
                           h->d_desc, h->d_kp_count, h->d_kp_pt, h->img2_base ? h->img2_base : B, kcap, h->cfg.max_matches, 0.0,
                           h->d_m_q, h->d_m_t, h->d_m_d, h->d_m_n, h->d_pts1, h->d_pts2);
    else {
        // first LDS region: max(staging 16 KB + 256 B of packed words + 2 B per scanned descriptor, 4 B x sort size)
        int sortP = 64;
        while (sortP < kcap) sortP <<= 1;
        const size_t r0 = (std::max((size_t)(2 * 8 * 64 + 16) * 16 + (size_t)kcap * 2, (size_t)sortP * 4) + 15) / 16;
        const int rounds = ((kcap + 31) / 32 + 7) / 8;
        // the fused kernel keeps 8 bytes of election words per keypoint in LDS: beyond 64 KB per workgroup (nfeatures > ~4900) the
        // HBM-resident form serves every batch size (rpe_create sizes d_hm_* for the whole batch then)
        const bool lds_fits = r0 * 16 + (size_t)kcap * 8 <= 65536;
        const int split = B <= RPE_MATCH_SPLIT_PAIRS ? std::min(rounds, std::max(1, 256 / B)) : 1;
        if (split > 1 || !lds_fits) {
            hipMemsetAsync(h->d_hm_best, 0xFF, sizeof(unsigned) * (size_t)B * kcap, h->stream);
            hipMemsetAsync(h->d_hm_row, 0xFE, sizeof(unsigned) * (size_t)B * kcap, h->stream);
            hipLaunchKernelGGL((match_hamming_mfma_kernel<true>), dim3(B, split), dim3(MM_NT), r0 * 16, h->stream,
                               h->d_desc, h->d_kp_count, h->d_kp_pt, h->img2_base ? h->img2_base : B, kcap, h->cfg.max_matches, (int)r0,
                               h->d_hm_best, h->d_hm_row, h->d_m_q, h->d_m_t, h->d_m_d, h->d_m_n, h->d_pts1, h->d_pts2);
            hipLaunchKernelGGL(match_hamming_select_kernel, dim3(B), dim3(MM_NT), (size_t)sortP * 4, h->stream,
                               (const unsigned *)h->d_hm_best, (const unsigned *)h->d_hm_row, h->d_kp_count, h->d_kp_pt,
                               h->img2_base ? h->img2_base : B, kcap, h->cfg.max_matches,
                               h->d_m_q, h->d_m_t, h->d_m_d, h->d_m_n, h->d_pts1, h->d_pts2);
        } else
            hipLaunchKernelGGL((match_hamming_mfma_kernel<false>), dim3(B), dim3(MM_NT), r0 * 16 + (size_t)kcap * 8, h->stream,
                               h->d_desc, h->d_kp_count, h->d_kp_pt, h->img2_base ? h->img2_base : B, kcap, h->cfg.max_matches, (int)r0,
                               (unsigned *)nullptr, (unsigned *)nullptr, h->d_m_q, h->d_m_t, h->d_m_d, h->d_m_n, h->d_pts1, h->d_pts2);
    }
}

// ===================================================================== L2 (SIFT)
// cv2.BFMatcher(NORM_L2, crossCheck=True) on SIFT descriptors (pose_estimator.py:94,:127-131).
// SIFT descriptors are integer-valued 0..255 (saturate_cast<uchar> in calcSIFTDescriptor), so
// they are kept as u8[128] in HBM and the squared distance is EXACT in integers:
// |a-b|^2 = |a|^2 + |b|^2 - 2 a.b.  The reported / compared distance is the f32 sqrt of that integer, exactly what
// cv2 computes in f32 (sum < 2^24).  Keys are 64-bit because the sort key is the f32 distance (distinct integers
// can collide after sqrt -> tie broken by index).
//
// crossCheck, as batchDistance runs it (two passes): NNt(i) = the query's nearest train (lowest train on ties),
// NNq(j) = the train's nearest query (lowest query on ties); (i, NNt(i)) is a match iff NNq(NNt(i)) == i.  Both passes
// are the same "every OWNED descriptor finds its nearest SCANNED one" loop with the two images swapped; each writes
// one packed (f32 distance bits << 18 | scanned index) word per owned descriptor, the select kernel joins them.
//
// The O(N^2) part runs on the matrix cores (match_l2_mfma_kernel): with a' = a - 128 (a XOR 0x80 per byte: u8 -> i8),
// |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b' and a'.b' is v_mfma_i32_32x32x32_i8 over K = 128 (4 MFMAs per 32 x 32 tile; exact
// integers).  The vector-ALU kernel below it (32 x v_dot4_u32_u8 per distance) remains for the Lowe-ratio extension,
// which also needs the second-best distance, and as the A/B diagnostic (RPE_MATCH_VALU).
#define L2_QTILE 256

// NQ = descriptor bytes / 16: 8 for SIFT (128 B), 2 for ORB descriptors matched with NORM_L2 (32 B; cv2 builds this
// combination too, pose_estimator.py:115-131 -- batchDistance on CV_8U rows gives sqrt((float)sum of squared byte
// differences), the same exact-integer form)
template <int NQ>
__device__ __forceinline__ unsigned dot_u8(const uint4 *q, const uint4 (&t)[NQ])
{
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const uint4 a = q[k];
        acc = __builtin_amdgcn_udot4(a.x, t[k].x, acc, false);
        acc = __builtin_amdgcn_udot4(a.y, t[k].y, acc, false);
        acc = __builtin_amdgcn_udot4(a.z, t[k].z, acc, false);
        acc = __builtin_amdgcn_udot4(a.w, t[k].w, acc, false);
    }
    return acc;
}

// Vector-ALU form.  Workgroup = (256 owned descriptors, pair); the scanned descriptors stream through LDS.
// MODE 0: owned = trains (NNq), MODE 2: owned = queries (NNt), MODE 1: Lowe ratio (extension; owned = queries).
template <int NQ, int MODE>
__global__ __launch_bounds__(256) void match_l2_nearest_kernel(const uint8_t *__restrict__ desc, const int *__restrict__ kp_count,
                                                                int img2_base, int kcap, double ratio, unsigned long long *__restrict__ best)
{
    constexpr int DIM = NQ * 16;
    __shared__ uint4 s_q[L2_QTILE * NQ];                                        // 32 KB (SIFT) / 8 KB (ORB)
    __shared__ unsigned s_qn[L2_QTILE];                                         // |q|^2 of the tile
    const int tid = threadIdx.x, pair = blockIdx.y, tc = blockIdx.x * 256;
    const int img1 = pair, img2 = img2_base + pair;
    const int n1 = min(kp_count[img1], kcap), n2 = min(kp_count[img2], kcap);
    constexpr bool RATIO = MODE == 1, QOWN = MODE != 0;
    const int n_own = QOWN ? n1 : n2, n_scan = QOWN ? n2 : n1;
    if (tc >= n_own || n_scan <= 0) return;
    const uint4 *d1 = (const uint4 *)(desc + (long long)img1 * kcap * DIM);
    const uint4 *d2 = (const uint4 *)(desc + (long long)img2 * kcap * DIM);
    const uint4 *d_own = QOWN ? d1 : d2, *d_scan = QOWN ? d2 : d1;
    const int j = tc + tid;
    const bool valid = j < n_own;
    uint4 t[NQ];
    unsigned tn = 0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        t[k] = valid ? d_own[NQ * j + k] : make_uint4(0, 0, 0, 0);
        tn = __builtin_amdgcn_udot4(t[k].x, t[k].x, tn, false); tn = __builtin_amdgcn_udot4(t[k].y, t[k].y, tn, false);
        tn = __builtin_amdgcn_udot4(t[k].z, t[k].z, tn, false); tn = __builtin_amdgcn_udot4(t[k].w, t[k].w, tn, false);
    }
    float bestd = __builtin_inff(), second = __builtin_inff();
    int besti = -1;
    for (int qt = 0; qt < n_scan; qt += L2_QTILE) {
        const int nq = min(L2_QTILE, n_scan - qt);
        __syncthreads();
        {
            uint4 stage[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const int idx = tid + 256 * q; stage[q] = idx < nq * NQ ? d_scan[NQ * qt + idx] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int q = 0; q < NQ; ++q) s_q[tid + 256 * q] = stage[q];
        }
        __syncthreads();
        if (tid < nq) {
            unsigned qn = 0;
            for (int k = 0; k < NQ; ++k) {
                const uint4 a = s_q[NQ * tid + k];
                qn = __builtin_amdgcn_udot4(a.x, a.x, qn, false); qn = __builtin_amdgcn_udot4(a.y, a.y, qn, false);
                qn = __builtin_amdgcn_udot4(a.z, a.z, qn, false); qn = __builtin_amdgcn_udot4(a.w, a.w, qn, false);
            }
            s_qn[tid] = qn;
        }
        __syncthreads();
        for (int i = 0; i < nq; ++i) {
            const unsigned ab = dot_u8<NQ>(s_q + NQ * i, t);
            const float d = sqrtf((float)(s_qn[i] + tn - 2u * ab));
            if (!RATIO) { if (d < bestd) { bestd = d; besti = qt + i; } }
            else {
                if (d < bestd) { second = bestd; bestd = d; besti = qt + i; }
                else if (d < second) second = d;
            }
        }
    }
    if (!RATIO) {
        if (valid && besti >= 0)
            best[(long long)pair * kcap + j] = ((unsigned long long)__float_as_uint(bestd) << 18) | (unsigned long long)besti;
    } else if (valid && n_scan >= 2 && (double)bestd < ratio * (double)second) {
        best[(long long)pair * kcap + j] = ((unsigned long long)__float_as_uint(bestd) << 18) | (unsigned long long)besti;
    }
}

// Per descriptor a (u = a - 128 per byte): { |u|^2, |u|^2 + 2 sum(u) } -- the owner's and the scanned row's terms of the MFMA kernel
template <int NQ>
__global__ __launch_bounds__(256) void match_l2_norms_kernel(const uint8_t *__restrict__ desc, const int *__restrict__ kp_count, int kcap,
                                                              int2 *__restrict__ norms)
{
    const int img = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(kp_count[img], kcap)) return;
    const uint4 *d = (const uint4 *)(desc + ((long long)img * kcap + k) * (NQ * 16));
    int acc = 0, sum = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint4 a = d[q];
        const int x = (int)(a.x ^ 0x80808080u), y = (int)(a.y ^ 0x80808080u), z = (int)(a.z ^ 0x80808080u), w = (int)(a.w ^ 0x80808080u);
        acc = __builtin_amdgcn_sdot4(x, x, acc, false); acc = __builtin_amdgcn_sdot4(y, y, acc, false);
        acc = __builtin_amdgcn_sdot4(z, z, acc, false); acc = __builtin_amdgcn_sdot4(w, w, acc, false);
        sum = __builtin_amdgcn_sdot4(x, 0x01010101, sum, false); sum = __builtin_amdgcn_sdot4(y, 0x01010101, sum, false);
        sum = __builtin_amdgcn_sdot4(z, 0x01010101, sum, false); sum = __builtin_amdgcn_sdot4(w, 0x01010101, sum, false);
    }
    norms[(long long)img * kcap + k] = make_int2(acc, acc + 2 * sum);
}

// Matrix-core form of one crossCheck pass.  Workgroup = 4 waves; wave w owns 32 descriptors (tile 4 blockIdx.x + w): the
// B operand (MFMA columns) of all KS K-steps, in registers.  The scanned descriptors stream through LDS as A-operand
// tiles of 32 rows (lane l of K-step s holds bytes [32 s + 16 (l >> 5), +16) of row l & 31: a plain 16-byte piece of
// the descriptor), double buffered, fetched PF tiles ahead.  blockIdx.y deals the scanned tiles over gridDim.y
// workgroups (small batches: one pair of 12 k x 12 k descriptors fills the chip), blockIdx.z = 2 pair + pass.
// The owner operand is the one's complement w = -v - 1 = b XOR 0x7F of v = b - 128 (its negative would not fit int8), so
// u.v = -u.w - sum(u) and |a-b|^2 = |v|^2 + e with e = (|u|^2 + 2 sum(u)) + 2 u.w: one v_lshl_add_u32 per accumulator,
// the row term comes from the norms kernel, the owner's |v|^2 is a per-lane constant and stays out of the comparison.  cv2 compares f32 distances: sqrtf is monotone, so a row can only win if its e is no larger than
// the running best's -- or larger by at most 2, when two integers share one f32 square root (only beyond 2^22) and the
// row has the lower index.  That test runs on the tile minimum; the exact (f32 distance, index) comparison of the
// 16 rows runs only in a wave where some lane passes it.
#define L2M_PF 3
template <int KS>
__global__ __launch_bounds__(256) void match_l2_mfma_kernel(const uint8_t *__restrict__ desc, const int2 *__restrict__ norms,
                                                             const int *__restrict__ kp_count, int img2_base, int kcap,
                                                             unsigned long long *__restrict__ nn_t, unsigned long long *__restrict__ nn_q)
{
    constexpr int DIM = KS * 32;
    __shared__ v4i_t s_a[2][KS][64];
    __shared__ __attribute__((aligned(16))) int s_qn[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5, col = lane & 31;
    const int pair = blockIdx.z >> 1, pass = blockIdx.z & 1;
    const int img1 = pair, img2 = img2_base + pair;
    const int n1 = min(kp_count[img1], kcap), n2 = min(kp_count[img2], kcap);
    // pass 0: the queries own the columns and find their nearest train (NNt); pass 1: the trains find their nearest query
    const int img_own = pass ? img2 : img1, img_scan = pass ? img1 : img2;
    const int n_own = pass ? n2 : n1, n_scan = pass ? n1 : n2;
    if (blockIdx.x * 128 >= n_own || n_scan <= 0) return;
    const int ntq = (n_scan + 31) >> 5;
    const int per = (ntq + (int)gridDim.y - 1) / (int)gridDim.y;
    const int qt_lo = (int)blockIdx.y * per, qt_hi = min(ntq, qt_lo + per);
    if (qt_lo >= qt_hi) return;
    const uint4 *d_own = (const uint4 *)(desc + (long long)img_own * kcap * DIM);
    const uint4 *d_scan = (const uint4 *)(desc + (long long)img_scan * kcap * DIM);
    const int2 *nrm_scan = norms + (long long)img_scan * kcap;
    const int j = (blockIdx.x * 4 + wv) * 32 + col;
    const bool valid_own = j < n_own;
    v4i_t bop[KS];
#pragma unroll
    for (int sK = 0; sK < KS; ++sK) {
        uint4 t = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
        if (valid_own) t = d_own[(long long)j * (DIM / 16) + 2 * sK + h];
        bop[sK].x = (int)(t.x ^ 0x7F7F7F7Fu); bop[sK].y = (int)(t.y ^ 0x7F7F7F7Fu); bop[sK].z = (int)(t.z ^ 0x7F7F7F7Fu); bop[sK].w = (int)(t.w ^ 0x7F7F7F7Fu);
    }
    const int tn = valid_own ? norms[(long long)img_own * kcap + j].x : 0;
    // loader: thread (s = tid >> 6, l = tid & 63) brings piece (row l & 31, bytes [32 s + 16 (l >> 5), +16)) of a tile
    const int xs = tid >> 6, xrow = lane & 31, xh = lane >> 5;
    const bool loader = xs < KS;
    auto fetch = [&](int qt) -> uint4 {
        const int q = qt * 32 + xrow;
        return (loader && qt < qt_hi && q < n_scan) ? d_scan[(long long)q * (DIM / 16) + 2 * xs + xh] : make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    };
    auto fetch_n = [&](int qt) -> int {
        const int q = qt * 32 + tid;
        return (tid < 32 && qt < qt_hi && q < n_scan) ? nrm_scan[q].y : 0x3FFFFFFF;     // rows past the end never win
    };
    auto stage = [&](int buf, const uint4 &raw, int qn) {
        if (loader) {
            v4i_t v; v.x = (int)(raw.x ^ 0x80808080u); v.y = (int)(raw.y ^ 0x80808080u); v.z = (int)(raw.z ^ 0x80808080u); v.w = (int)(raw.w ^ 0x80808080u);
            s_a[buf][xs][lane] = v;
        }
        if (tid < 32) s_qn[buf][tid] = qn;
    };
    uint4 ring[L2M_PF]; int ring_n[L2M_PF];
#pragma unroll
    for (int u = 0; u < L2M_PF; ++u) { ring[u] = fetch(qt_lo + u); ring_n[u] = fetch_n(qt_lo + u); }
    stage(0, ring[0], ring_n[0]);
    ring[0] = fetch(qt_lo + L2M_PF); ring_n[0] = fetch_n(qt_lo + L2M_PF);
    __syncthreads();
    int best_e = 0x3FFFFFF0, best_idx = 0x7FFFFFFF;
    float best_g = __builtin_inff();
    for (int qt0 = qt_lo; qt0 < qt_hi; qt0 += L2M_PF) {
#pragma unroll
        for (int u = 0; u < L2M_PF; ++u) {
            const int qt = qt0 + u;
            if (qt < qt_hi) {                                  // workgroup-uniform
                const int buf = (qt - qt_lo) & 1;
                v16i_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int sK = 0; sK < KS; ++sK)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_a[buf][sK][lane], bop[sK], acc, 0, 0, 0);
                // C layout: column = lane & 31, row of register r = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
                int e[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const v4i_t qn4 = *(const v4i_t *)&s_qn[buf][8 * g4 + 4 * h];
                    e[4 * g4 + 0] = (acc[4 * g4 + 0] << 1) + qn4.x; e[4 * g4 + 1] = (acc[4 * g4 + 1] << 1) + qn4.y;
                    e[4 * g4 + 2] = (acc[4 * g4 + 2] << 1) + qn4.z; e[4 * g4 + 3] = (acc[4 * g4 + 3] << 1) + qn4.w;
                }
                int tmin = e[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmin = min(tmin, e[r]);
                if (__any(tmin <= best_e + 2)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (e[r] <= best_e + 2) {
                            const float g = sqrtf((float)(e[r] + tn));
                            const int idx = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (g < best_g || (g == best_g && idx < best_idx)) { best_g = g; best_idx = idx; best_e = e[r]; }
                        }
                    }
                }
                if (qt + 1 < qt_hi) {
                    stage(buf ^ 1, ring[(u + 1) % L2M_PF], ring_n[(u + 1) % L2M_PF]);
                    ring[(u + 1) % L2M_PF] = fetch(qt + 1 + L2M_PF); ring_n[(u + 1) % L2M_PF] = fetch_n(qt + 1 + L2M_PF);
                }
                __syncthreads();
            }
        }
    }
    // the two half-waves hold the same columns (different rows)
    {
        const float og = __shfl_xor(best_g, 32); const int oi = __shfl_xor(best_idx, 32);
        if (og < best_g || (og == best_g && oi < best_idx)) { best_g = og; best_idx = oi; }
    }
    if (valid_own && h == 0 && best_idx != 0x7FFFFFFF) {
        unsigned long long *out = (pass ? nn_q : nn_t) + (long long)pair * kcap + j;
        atomicMin(out, ((unsigned long long)__float_as_uint(best_g) << 18) | (unsigned long long)best_idx);
    }
}

// Kernel 2: workgroup = pair: stable sort of the (distance, queryIdx) keys of the surviving matches, first max_matches,
// point gather.  nn_t[i] = query i's nearest train; nn_q (crossCheck only, nullptr for the ratio mode) = train j's
// nearest query: the match survives iff nn_q[nn_t[i]] names i
__global__ __launch_bounds__(256) void match_l2_select_kernel(const unsigned long long *__restrict__ nn_t, const unsigned long long *__restrict__ nn_q,
                                                               const int *__restrict__ kp_count,
                                                               const float2 *__restrict__ kp_pt, int img2_base, int kcap, int max_matches,
                                                               int *__restrict__ m_q, int *__restrict__ m_t, float *__restrict__ m_d,
                                                               int *__restrict__ m_n, float2 *__restrict__ pts1, float2 *__restrict__ pts2)
{
    extern __shared__ unsigned long long s_key[];          // sortP <= 16384 keys (128 KB)
    __shared__ int s_valid;
    const int tid = threadIdx.x, pair = blockIdx.x;
    const int img1 = pair, img2 = img2_base + pair;
    const int n1 = min(kp_count[img1], kcap);
    const unsigned long long *bp = nn_t + (long long)pair * kcap;
    if (tid == 0) s_valid = 0;
    __syncthreads();
    int sortP = 64;
    while (sortP < n1) sortP <<= 1;
    int myvalid = 0;
    for (int i = tid; i < sortP; i += 256) {
        unsigned long long key = ~0ull;
        if (i < n1) {
            unsigned long long b = bp[i];
            if (b != ~0ull && (!nn_q || (int)(nn_q[(long long)pair * kcap + (int)(b & 0x3FFFF)] & 0x3FFFF) == i)) { key = ((b >> 18) << 16) | (unsigned long long)i; ++myvalid; }
        }
        s_key[i] = key;
    }
    if (myvalid) atomicAdd(&s_valid, myvalid);
    __syncthreads();
    for (int k = 2; k <= sortP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t2 = tid; t2 < (sortP >> 1); t2 += 256) {
                int i = 2 * j * (t2 / j) + (t2 % j);
                int ixj = i + j;
                bool asc = (i & k) == 0;
                unsigned long long a = s_key[i], b = s_key[ixj];
                if ((a > b) == asc) { s_key[i] = b; s_key[ixj] = a; }
            }
            __syncthreads();
        }
    }
    const int nm = min(s_valid, max_matches);
    for (int r = tid; r < nm; r += 256) {
        unsigned long long key = s_key[r];
        int i = (int)(key & 0xFFFF);
        int j = (int)(bp[i] & 0x3FFFF);
        long long o = (long long)pair * max_matches + r;
        m_q[o] = i; m_t[o] = j; m_d[o] = __uint_as_float((unsigned)(key >> 16));
        pts1[o] = kp_pt[(long long)img1 * kcap + i];
        pts2[o] = kp_pt[(long long)img2 * kcap + j];
    }
    if (tid == 0) m_n[pair] = nm;
}

void rpe_launch_match_l2(rpe_handle *h, int B)
{
    const int kcap = h->lay.kcap;
    const int img2_base = h->img2_base ? h->img2_base : B;
    // d_m_best2 = NNt (per query; the ratio mode's only list), d_m_best = NNq (per train)
    hipMemsetAsync(h->d_m_best2, 0xFF, sizeof(unsigned long long) * (size_t)B * kcap, h->stream);
    const bool rt = h->cfg.match_mode == RPE_MATCH_RATIO;
    if (!rt) hipMemsetAsync(h->d_m_best, 0xFF, sizeof(unsigned long long) * (size_t)B * kcap, h->stream);
    const bool sift = h->desc_bytes == 128;
    if (rt || getenv("RPE_MATCH_VALU")) {
        const dim3 grid((kcap + 255) / 256, B);
#define L2_LAUNCH(NQ, MODE, DST) hipLaunchKernelGGL((match_l2_nearest_kernel<NQ, MODE>), grid, dim3(256), 0, h->stream, \
                                                    h->d_desc, h->d_kp_count, img2_base, kcap, h->cfg.match_ratio, DST)
        if (sift) { if (rt) L2_LAUNCH(8, 1, h->d_m_best2); else { L2_LAUNCH(8, 0, h->d_m_best); L2_LAUNCH(8, 2, h->d_m_best2); } }
        else      { if (rt) L2_LAUNCH(2, 1, h->d_m_best2); else { L2_LAUNCH(2, 0, h->d_m_best); L2_LAUNCH(2, 2, h->d_m_best2); } }
#undef L2_LAUNCH
    } else {
        const int n_img = img2_base + B;                       // images [0, B) and [img2_base, img2_base + B) (a stream: B + 1 frames)
        const dim3 gn((kcap + 255) / 256, n_img);
        if (sift) hipLaunchKernelGGL(match_l2_norms_kernel<8>, gn, dim3(256), 0, h->stream, h->d_desc, h->d_kp_count, kcap, (int2 *)h->d_m_norm);
        else      hipLaunchKernelGGL(match_l2_norms_kernel<2>, gn, dim3(256), 0, h->stream, h->d_desc, h->d_kp_count, kcap, (int2 *)h->d_m_norm);
        // scanned tiles dealt over `split` workgroups until ~1024 workgroups are in flight (a single pair included)
        const int chunks = (kcap + 127) / 128;
        int split = (1024 + chunks * 2 * B - 1) / (chunks * 2 * B);
        split = split < 1 ? 1 : split > 8 ? 8 : split;
        const dim3 grid(chunks, split, 2 * B);
        if (sift) hipLaunchKernelGGL(match_l2_mfma_kernel<4>, grid, dim3(256), 0, h->stream, h->d_desc, (const int2 *)h->d_m_norm, h->d_kp_count, img2_base, kcap, h->d_m_best2, h->d_m_best);
        else      hipLaunchKernelGGL(match_l2_mfma_kernel<1>, grid, dim3(256), 0, h->stream, h->d_desc, (const int2 *)h->d_m_norm, h->d_kp_count, img2_base, kcap, h->d_m_best2, h->d_m_best);
    }
    int sortP = 64;
    while (sortP < kcap) sortP <<= 1;
    if (sizeof(unsigned long long) * (size_t)sortP > 65536)   // more than 8128 keypoints per image: up to 128 KB of the CU's 160 KB
        hipFuncSetAttribute((const void *)match_l2_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned long long) * (size_t)sortP));
    hipLaunchKernelGGL(match_l2_select_kernel, dim3(B), dim3(256), sizeof(unsigned long long) * (size_t)sortP, h->stream,
                       (const unsigned long long *)h->d_m_best2, rt ? (const unsigned long long *)nullptr : (const unsigned long long *)h->d_m_best,
                       h->d_kp_count, h->d_kp_pt, img2_base, kcap, h->cfg.max_matches,
                       h->d_m_q, h->d_m_t, (float *)h->d_m_d, h->d_m_n, h->d_pts1, h->d_pts2);
}
