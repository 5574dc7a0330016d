// geom_kernels.hip -- essential-matrix RANSAC and pose recovery on gfx950.
//
// Replaces cv2.findEssentialMat(pts1, pts2, K, RANSAC, 0.999, 1.0)
// (reference src/core/pose_estimator.py:522-527) and cv2.recoverPose(E, pts1, pts2, K)
// (:533).  Sequential OpenCV semantics (calib3d/ptsetreg.cpp RANSACPointSetRegistrator::run:
// fixed RNG stream, "strictly more inliers wins", adaptive niters) are reproduced on a
// parallel machine by evaluating a chunk of iterations at a time (32, 96, 384, 512) and replaying
// the update rule over the per-model inlier counts:
//   ransac_prepare : normalise points with K (f64), reset per-pair state
//   ransac_poly    : one lane = one minimal sample: null space, 10x20 elimination in lane-interleaved
//                    LDS, determinant polynomial of degree 10 (f64)
//   ransac_roots   : 16 lanes per sample: lane = bracketing interval of the derivative chain, safeguarded
//                    Newton, ballot/shuffle compaction; back-substitution lane = root -> up to 10 models
//   ransac_score   : workgroup per (pair, 64 iterations), wave per model: Sampson error (f64 -> f32
//                    compare) over the matches in LDS, shuffle-reduced inlier counts
//   ransac_update  : wave per pair: the sequential "strictly more inliers wins / niters shrinks / stop at
//                    niters" rule as prefix-max and prefix-min scans (bit-identical termination)
//   ransac_mask    : inlier mask of the winning model (stage API only)
//   recover_pose   : 3x3 one-sided Jacobi SVD, 4 candidate poses, per-point 4x4 Jacobi
//                    DLT triangulation + cheirality vote, wave-reduced counts
// All f64 arithmetic uses the oracle's operation order (compiled with -ffp-contract=off).
#include "rpe_internal.h"
#include <float.h>
#include <stdlib.h>
#include <algorithm>

// ------------------------------------------------ polynomial bookkeeping
// lin: x y z 1 ; quad: x2 y2 z2 xy xz yz x y z 1 ;
// cubic (Nister's elimination order): x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1
__device__ static const signed char LL2Q[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
__device__ static const signed char QL2C[10][4] = {{0, 2, 4, 5},   {3, 1, 6, 7},   {10, 13, 16, 17}, {2, 3, 8, 9},    {4, 8, 10, 11},
                                                    {8, 6, 13, 14}, {5, 9, 11, 12}, {9, 7, 14, 15},   {11, 14, 17, 18}, {12, 15, 18, 19}};

__device__ static void ll_acc(double *c, const double *a, const double *b)
{
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[LL2Q[i][j]] += a[i] * b[j];
}
__device__ static void ql_acc(double *c, const double *a, const double *b, double s)
{
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) c[QL2C[i][j]] += s * (a[i] * b[j]);
}

// Fixed Estrin scheme for degree <= 10 (dependency depth 7 instead of Horner's 20; the
// root finder is latency bound).  Identical arithmetic to oracle/geom_oracle.c.
__device__ static double horner(const double *c, int n, double x)
{
    double cc[11];
    for (int i = 0; i < 11; ++i) cc[i] = i <= n ? c[i] : 0.;
    const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    const double a0 = cc[0] + cc[1] * x, a1 = cc[2] + cc[3] * x, a2 = cc[4] + cc[5] * x;
    const double a3 = cc[6] + cc[7] * x, a4 = cc[8] + cc[9] * x, a5 = cc[10];
    const double b0 = a0 + a1 * x2, b1 = a2 + a3 * x2, b2 = a4 + a5 * x2;
    return (b0 + b1 * x4) + b2 * x8;
}

// Safeguarded Newton on a bracket with a sign change (same code path as the oracle).
__device__ static double refine_root(const double *p, const double *dp, int k, double a, double b, int sa)
{
    double xl = sa ? b : a, xh = sa ? a : b;
    double rts = 0.5 * (a + b);
    double dxold = fabs(b - a), dx = dxold;
    double f = horner(p, k, rts), df = horner(dp, k - 1, rts);
    for (int it = 0; it < 100; ++it) {
        int bis = ((((rts - xh) * df - f) * ((rts - xl) * df - f)) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df));
        double nr;
        dxold = dx;
        if (bis) { dx = 0.5 * (xh - xl); nr = xl + dx; }
        else { dx = f / df; nr = rts - dx; }
        if (nr == rts || fabs(nr - rts) <= 2.3e-13 * fabs(nr)) { rts = nr; break; }
        rts = nr;
        f = horner(p, k, rts); df = horner(dp, k - 1, rts);
        if (f > 0.0) xh = rts; else xl = rts;
    }
    return rts;
}

// Root bound from binary exponents only (same integers on CPU and GPU): Fujiwara's |z| <= 2 max_i |a_{k-i}/a_k|^(1/i)
// with |a| < 2^(ilogb(a)+1): R = 2^(1 + max_i ceil((e_{k-i} - e_k + 1)/i)).  Cauchy's bound put the outer brackets
// orders of magnitude beyond the roots and the safeguarded Newton bisected its way back (oracle: root_bound()).
__device__ static double root_bound(const double *p, int k)
{
    const int ek = ilogb(p[k]);
    int emax = -100000;
    for (int i = 0; i < k; ++i) {
        if (p[i] == 0.) continue;
        const int d = ilogb(p[i]) - ek + 1, m = k - i;
        const int q = d >= 0 ? (d + m - 1) / m : -((-d) / m);
        if (q > emax) emax = q;
    }
    double R = emax == -100000 ? 1. : ldexp(1., emax + 1);
    if (!(R < 1e12)) R = 1e12;
    return R;
}
template <int K>
__device__ __forceinline__ double root_bound_s(const double (&p)[11])
{
    const int ek = ilogb(p[K]);
    int emax = -100000;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (p[i] != 0.) {
            const int d = ilogb(p[i]) - ek + 1;
            const int m = K - i;
            const int q = d >= 0 ? (d + m - 1) / m : -((-d) / m);
            if (q > emax) emax = q;
        }
    }
    double R = emax == -100000 ? 1. : ldexp(1., emax + 1);
    if (!(R < 1e12)) R = 1e12;
    return R;
}

// generic path (leading coefficient trimmed to degree < 10: rare): one lane walks the whole chain serially with dynamic
// indexing.  Its arrays live in workspace the CALLER
// provides (LDS in ransac_roots_kernel): as private arrays they are a scratch segment, and a kernel with one is
// dispatched with fewer resident waves.  Level k's polynomial is c differentiated n - k times, rebuilt from c on every
// level by the same multiplications in the same order as the oracle's derivative table (identical values), so the
// workspace is two rows instead of the 11 x 11 table.
#define GEN_WS_DOUBLES (4 * 11)          // p[11], dp[11], rts[2][11]
__device__ static int poly_real_roots_generic(const double *c, int n, double *roots, double *ws)
{
    double *p = ws, *dp = ws + 11, *rts = ws + 22;          // rts[which * 11 + i]
    int nr_prev = 0, cur = 0;
    for (int k = 1; k <= n; ++k) {
        for (int i = 0; i <= n; ++i) p[i] = c[i];
        for (int kk = n; kk > k; --kk)
            for (int i = 0; i < kk; ++i) p[i] = p[i + 1] * (double)(i + 1);
        if (k == 1) { rts[0] = -p[0] / p[1]; nr_prev = 1; cur = 0; continue; }
        for (int i = 0; i < k; ++i) dp[i] = p[i + 1] * (double)(i + 1);
        const double *crit = rts + cur * 11;
        double *out = rts + (cur ^ 1) * 11;
        int nout = 0;
        double R = root_bound(p, k);
        for (int iv = 0; iv <= nr_prev; ++iv) {
            double a = (iv == 0) ? -R : crit[iv - 1];
            double b = (iv == nr_prev) ? R : crit[iv];
            if (a < -R) a = -R;
            if (b > R) b = R;
            if (!(a < b)) continue;
            int sa = horner(p, k, a) > 0., sb = horner(p, k, b) > 0.;
            if (sa == sb) continue;
            out[nout++] = refine_root(p, dp, k, a, b, sa);
        }
        nr_prev = nout; cur ^= 1;
    }
    for (int i = 0; i < nr_prev; ++i) roots[i] = rts[cur * 11 + i];
    return nr_prev;
}

// ---- degree-10 fast path: the polynomial of each chain level lives in registers
// (static indexing); arithmetic identical to the generic path.
// static-degree Estrin: terms whose coefficients are structurally zero (index > K) are
// omitted; they would only add exact zeros, so the value equals horner(c, K, x) bit for bit.
template <int K>
__device__ __forceinline__ double horner_s(const double (&c)[11], double x)
{
    const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    auto A = [&](int i) -> double {           // a_i = c[2i] + c[2i+1] x
        if (2 * i + 1 <= K) return c[2 * i] + c[2 * i + 1] * x;
        return (2 * i <= K) ? c[2 * i] : 0.;
    };
    double b0, b1 = 0., b2 = 0.;
    b0 = (2 <= K) ? A(0) + A(1) * x2 : A(0);
    if (4 <= K) b1 = (6 <= K) ? A(2) + A(3) * x2 : A(2);
    if (8 <= K) b2 = (10 <= K) ? A(4) + c[10] * x2 : A(4);
    double r = (4 <= K) ? b0 + b1 * x4 : b0;
    if (8 <= K) r = r + b2 * x8;
    return r;
}

template <int K>
__device__ __forceinline__ double refine_root_s(const double (&p)[11], const double (&dp)[11], double a, double b, int sa)
{
    double xl = sa ? b : a, xh = sa ? a : b;
    double rts = 0.5 * (a + b);
    double dxold = fabs(b - a), dx = dxold;
    double f = horner_s<K>(p, rts), df = horner_s<K - 1>(dp, rts);
    for (int it = 0; it < 100; ++it) {
        int bis = ((((rts - xh) * df - f) * ((rts - xl) * df - f)) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df));
        double nr;
        dxold = dx;
        if (bis) { dx = 0.5 * (xh - xl); nr = xl + dx; }
        else { dx = f / df; nr = rts - dx; }
        if (nr == rts || fabs(nr - rts) <= 2.3e-13 * fabs(nr)) { rts = nr; break; }
        rts = nr;
        f = horner_s<K>(p, rts); df = horner_s<K - 1>(dp, rts);
        if (f > 0.0) xh = rts; else xl = rts;
    }
    return rts;
}

// Nister five-point solver (five-point.cpp EMEstimatorCallback::runKernel restated;
// same operation order as oracle/geom_oracle.c)
// mx: this lane's 10x20 elimination matrix in LDS, element (i,j) at mx[(i*20+j)*POLY_LANES]
// (lane-interleaved: conflict-free ds_read/write_b64, no scratch round trips)
#define SCORE_GROUP 8       // iterations scored by one workgroup of ransac_score_kernel
#define POLY_LANES 32        // minimal samples per wave of ransac_poly_kernel (lanes 32..63 idle): see the kernel
#define MX(i, j) mx[((i) * 20 + (j)) * POLY_LANES]
// Part A of the solver (LDS-heavy): null space, constraint matrix, Gauss-Jordan, det B(z).
// Writes the hypothesis record rec[k*64] (k = 0..86): c10[11], Bx[12], By[12], B1[15], Eb[36], degree n;
// returns 0 when the elimination is singular.
#define HYP_DOUBLES 88
__device__ static int five_point_poly(const double *x1, const double *x2, double *mx, double *rec)
{
    double A[9][5];
    for (int k = 0; k < 5; ++k) {
        double a = x1[2 * k], b = x1[2 * k + 1], c = x2[2 * k], d = x2[2 * k + 1];
        A[0][k] = c * a; A[1][k] = c * b; A[2][k] = c;
        A[3][k] = d * a; A[4][k] = d * b; A[5][k] = d;
        A[6][k] = a;     A[7][k] = b;     A[8][k] = 1.;
    }
    double hv[5][9], beta[5];
    for (int k = 0; k < 5; ++k) {
        double nrm = 0.;
        for (int i = k; i < 9; ++i) nrm += A[i][k] * A[i][k];
        nrm = sqrt(nrm);
        double alpha = A[k][k] > 0. ? -nrm : nrm;
        for (int i = 0; i < 9; ++i) hv[k][i] = 0.;
        hv[k][k] = A[k][k] - alpha;
        for (int i = k + 1; i < 9; ++i) hv[k][i] = A[i][k];
        double vn = 0.;
        for (int i = k; i < 9; ++i) vn += hv[k][i] * hv[k][i];
        beta[k] = vn > 0. ? 2. / vn : 0.;
        for (int j = k; j < 5; ++j) {
            double s = 0.;
            for (int i = k; i < 9; ++i) s += hv[k][i] * A[i][j];
            s *= beta[k];
            for (int i = k; i < 9; ++i) A[i][j] -= s * hv[k][i];
        }
    }
    double Eb[4][9];
    for (int m = 0; m < 4; ++m) {
        double v[9];
        for (int i = 0; i < 9; ++i) v[i] = 0.;
        v[5 + m] = 1.;
        for (int k = 4; k >= 0; --k) {
            double s = 0.;
            for (int i = k; i < 9; ++i) s += hv[k][i] * v[i];
            s *= beta[k];
            for (int i = k; i < 9; ++i) v[i] -= s * hv[k][i];
        }
        for (int i = 0; i < 9; ++i) Eb[m][i] = v[i];
    }
    double El[9][4];
    for (int e = 0; e < 9; ++e) for (int m = 0; m < 4; ++m) El[e][m] = Eb[m][e];
    double EEt[3][3][10];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int q = 0; q < 10; ++q) EEt[i][j][q] = 0.;
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) {
        for (int k = 0; k < 3; ++k) ll_acc(EEt[i][j], El[i * 3 + k], El[j * 3 + k]);
        if (j != i) for (int q = 0; q < 10; ++q) EEt[j][i][q] = EEt[i][j][q];
    }
    double htr[10];
    for (int q = 0; q < 10; ++q) htr[q] = 0.5 * ((EEt[0][0][q] + EEt[1][1][q]) + EEt[2][2][q]);
    for (int i = 0; i < 3; ++i) for (int q = 0; q < 10; ++q) EEt[i][i][q] -= htr[q];

    // constraint rows are accumulated in registers/scratch one at a time, then stored to LDS
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double row[20];
        for (int q = 0; q < 20; ++q) row[q] = 0.;
        for (int k = 0; k < 3; ++k) ql_acc(row, EEt[i][k], El[k * 3 + j], 1.);
        for (int q = 0; q < 20; ++q) MX(i * 3 + j, q) = row[q];
    }
    {
        double m0[10], m1[10], m2[10], neg[4];
        for (int q = 0; q < 10; ++q) { m0[q] = 0.; m1[q] = 0.; m2[q] = 0.; }
        ll_acc(m0, El[4], El[8]); for (int q = 0; q < 4; ++q) neg[q] = -El[5][q]; ll_acc(m0, neg, El[7]);
        ll_acc(m1, El[3], El[8]); ll_acc(m1, neg, El[6]);
        ll_acc(m2, El[3], El[7]); for (int q = 0; q < 4; ++q) neg[q] = -El[4][q]; ll_acc(m2, neg, El[6]);
        double row[20];
        for (int q = 0; q < 20; ++q) row[q] = 0.;
        ql_acc(row, m0, El[0], 1.);
        ql_acc(row, m1, El[1], -1.);
        ql_acc(row, m2, El[2], 1.);
        for (int q = 0; q < 20; ++q) MX(9, q) = row[q];
    }
    for (int c = 0; c < 10; ++c) {
        double col[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) col[r] = MX(r, c);          // 10 independent LDS reads in flight
        int piv = c; double best = fabs(col[c]);
        for (int r = c + 1; r < 10; ++r) { double a = fabs(col[r]); if (a > best) { best = a; piv = r; } }
        if (best == 0.) return 0;
        double prow[20], crow[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) { prow[j] = MX(piv, j); crow[j] = MX(c, j); }
        if (piv != c) {
#pragma unroll
            for (int j = 0; j < 20; ++j) MX(piv, j) = crow[j];
            double t = col[c]; col[c] = col[piv]; col[piv] = t;
        }
        const double inv = 1. / prow[c];
#pragma unroll
        for (int j = 0; j < 20; ++j) { if (j >= c) prow[j] = prow[j] * inv; }
#pragma unroll
        for (int j = 0; j < 20; ++j) MX(c, j) = prow[j];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r == c) continue;
            const double f = col[r];
            if (f == 0.) continue;
            double row[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) row[j] = MX(r, j);
#pragma unroll
            for (int j = 0; j < 20; ++j) { if (j >= c) row[j] -= f * prow[j]; }
#pragma unroll
            for (int j = 0; j < 20; ++j) MX(r, j) = row[j];
        }
    }
    double Bx[3][4], By[3][4], B1[3][5];
    for (int i = 0; i < 3; ++i) {
        double e[10], f[10];
        for (int q = 0; q < 10; ++q) { e[q] = MX(4 + 2 * i, 10 + q); f[q] = MX(5 + 2 * i, 10 + q); }
        Bx[i][3] = -f[0]; Bx[i][2] = e[0] - f[1]; Bx[i][1] = e[1] - f[2]; Bx[i][0] = e[2];
        By[i][3] = -f[3]; By[i][2] = e[3] - f[4]; By[i][1] = e[4] - f[5]; By[i][0] = e[5];
        B1[i][4] = -f[6]; B1[i][3] = e[6] - f[7]; B1[i][2] = e[7] - f[8]; B1[i][1] = e[8] - f[9]; B1[i][0] = e[9];
    }
    double c10[11];
    for (int i = 0; i < 11; ++i) c10[i] = 0.;
    for (int i = 0; i < 3; ++i) {
        int r0 = (i + 1) % 3, r1 = (i + 2) % 3;
        double minor[7];
        for (int k = 0; k < 7; ++k) minor[k] = 0.;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b)
            minor[a + b] += Bx[r0][a] * By[r1][b] - Bx[r1][a] * By[r0][b];
        for (int a = 0; a < 7; ++a) for (int b = 0; b < 5; ++b) c10[a + b] += minor[a] * B1[i][b];
    }
    int n = 10;
    for (; n > 1; --n) if (fabs(c10[n]) > DBL_EPSILON) break;
    for (int i = 0; i < 11; ++i) rec[i * 64] = c10[i];
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) { rec[(11 + i * 4 + k) * 64] = Bx[i][k]; rec[(23 + i * 4 + k) * 64] = By[i][k]; }
        for (int k = 0; k < 5; ++k) rec[(35 + i * 5 + k) * 64] = B1[i][k];
    }
    for (int m = 0; m < 4; ++m) for (int e = 0; e < 9; ++e) rec[(50 + m * 9 + e) * 64] = Eb[m][e];
    rec[86 * 64] = (double)n;
    return 1;
}

// ---------------------------------------------------------------- prepare
__global__ __launch_bounds__(256) void ransac_prepare_kernel(const float2 *__restrict__ pts1, const float2 *__restrict__ pts2,
                                                              const int *__restrict__ m_n, const double *__restrict__ K,
                                                              double2 *__restrict__ n1, double2 *__restrict__ n2,
                                                              RpeRansacState *__restrict__ st, int *__restrict__ found,
                                                              int max_matches, int max_iters)
{
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int M = min(m_n[pair], max_matches);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    if (i < M) {
        long long o = (long long)pair * max_matches + i;
        float2 a = pts1[o], b = pts2[o];
        n1[o] = make_double2(((double)a.x - cx) / fx, ((double)a.y - cy) / fy);
        n2[o] = make_double2(((double)b.x - cx) / fx, ((double)b.y - cy) / fy);
    }
    if (i == 0) {
        RpeRansacState s;
        s.best_count = 0; s.best_iter = -1; s.best_model = -1;
        s.niters = max_iters; s.next_iter = 0; s.done = (M < 5); s.found = 0; s.M = M; s.iters_run = 0; s.pad_ = 0;
        for (int e = 0; e < 9; ++e) s.E[e] = 0.;
        st[pair] = s;
        found[pair] = 0;
    }
}

// ------------------------------------------------------------------ solve
// A: one wave per (pair, 32 iterations), 32 active lanes.  The 10x20 elimination lives in LDS (1600 B per sample); the
// kernel retires 0.015 G vector instructions in 0.44 ms -- 3 % of the issue roof: it is pure single-wave latency, and a
// full 64-sample wave (100 KB) fits a CU only once.  Half-filled waves (50 KB) fit three times: three latency chains per
// CU instead of one (the idle lanes cost nothing a latency-bound kernel would have used).
__global__ __launch_bounds__(POLY_LANES) void ransac_poly_kernel(const double2 *__restrict__ n1, const double2 *__restrict__ n2,
                                                                  const RpeRansacState *__restrict__ st,
                                                                  const unsigned short *__restrict__ subsets,
                                                                  double *__restrict__ hyp, int *__restrict__ nmodels,
                                                                  int max_matches, int max_iters)
{
    __shared__ double s_mx[200 * POLY_LANES];
    // blockIdx.y = wave-chunk of 64 iterations * (64 / POLY_LANES) + part: lanes of this block = samples part*POLY_LANES .. of the chunk
    const int pair = blockIdx.x, wv = blockIdx.y / (64 / POLY_LANES), lane = (blockIdx.y % (64 / POLY_LANES)) * POLY_LANES + threadIdx.x;
    const RpeRansacState s = st[pair];
    if (s.done) return;
    const int it = s.next_iter + wv * 64 + lane;
    const long long slot = (long long)pair * RPE_RANSAC_MAXCHUNK + wv * 64 + lane;
    int ok = 0;
    const bool all5 = (s.M == 5);
    if (it < s.niters && (!all5 || it == 0)) {
        double x1[10], x2[10];
        const unsigned short *sub = subsets + ((long long)s.M * max_iters + it) * 5;
        for (int k = 0; k < 5; ++k) {
            int v = all5 ? k : (int)sub[k];
            double2 a = n1[(long long)pair * max_matches + v], b = n2[(long long)pair * max_matches + v];
            x1[2 * k] = a.x; x1[2 * k + 1] = a.y; x2[2 * k] = b.x; x2[2 * k + 1] = b.y;
        }
        ok = five_point_poly(x1, x2, s_mx + threadIdx.x, hyp + ((long long)pair * (RPE_RANSAC_MAXCHUNK / 64) + wv) * HYP_DOUBLES * 64 + lane);
    }
    nmodels[slot] = ok ? -1 : 0;          // -1: record valid, roots pending
}

// B: roots + back-substitution, RG = 8 lanes per minimal sample.  The real roots of the degree-k
// derivative split the line into <= k+1 intervals with at most one root of the degree-(k+1)
// derivative each, and those intervals are independent: lane j of a group brackets and refines
// interval j, the roots are compacted in interval order (ballot + n-th set bit + shuffle), so every
// lane performs exactly the arithmetic the sequential oracle performs for that interval and the
// root list comes out in the same order.  The kernel is bound by vector-instruction issue (r02 PMC: 0.2 G
// instructions in 0.43 ms for the first 64-iteration chunk, 76 % of the measured issue roof), and a wave costs what
// its slowest interval costs whatever the number of busy lanes: with 16 lanes per sample 70 % of the lanes idled (the
// derivative chain of these polynomials rarely has more than 4 real roots on a level); 8 lanes per sample halve the
// waves.  A level with more than 8 intervals (>= 8 real roots below it: levels 9 and 10 only) takes two rounds.
#define RG 8
#define RPE_RANSAC_FIRST_CHUNK 32   // iterations of the first launch group (schedule: rpe_launch_ransac)
__device__ __forceinline__ int nth_set_bit(unsigned m, int k)
{
    for (int t = 0; t < k; ++t) m &= m - 1;
    return m ? __ffs((int)m) - 1 : 0;
}

// crit: in = root #j of the level below (j < nr_prev), out = root #j of this level (j < return value)
template <int K>
__device__ __forceinline__ int roots_level_grp(const double (&c)[11], double &crit, int nr_prev, int j, int gbase)
{
    double p[11], dp[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) p[i] = c[i];
#pragma unroll
    for (int kk = 10; kk > K; --kk)
#pragma unroll
        for (int i = 0; i < kk; ++i) p[i] = p[i + 1] * (double)(i + 1);
#pragma unroll
    for (int i = 0; i < K; ++i) dp[i] = p[i + 1] * (double)(i + 1);
    const double R = root_bound_s<K>(p);
    const double below = __shfl_up(crit, 1);
    double a = (j == 0) ? -R : below;
    double b = (j == nr_prev) ? R : crit;
    if (a < -R) a = -R;
    if (b > R) b = R;
    bool has = false;
    double root = 0.;
    if (j <= nr_prev && a < b) {
        const int sa = horner_s<K>(p, a) > 0., sb = horner_s<K>(p, b) > 0.;
        if (sa != sb) { root = refine_root_s<K>(p, dp, a, b, sa); has = true; }
    }
    const unsigned m = (unsigned)(__ballot(has) >> gbase) & ((1u << RG) - 1u);
    crit = __shfl(root, gbase + nth_set_bit(m, j));
    return __popc(m);
}

// Levels 9 and 10 can have 9 / 10 intervals (8 / 9 real roots below): lane j then also takes interval 8 + j in a second
// round (lanes 0 and 1 only) and keeps root #(8 + j) in crit1.  Same arithmetic per interval, same root order.
template <int K>
__device__ __forceinline__ int roots_level_grp2(const double (&c)[11], double &crit0, double &crit1, int nr_prev, int j, int gbase)
{
    double p[11], dp[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) p[i] = c[i];
#pragma unroll
    for (int kk = 10; kk > K; --kk)
#pragma unroll
        for (int i = 0; i < kk; ++i) p[i] = p[i + 1] * (double)(i + 1);
#pragma unroll
    for (int i = 0; i < K; ++i) dp[i] = p[i + 1] * (double)(i + 1);
    const double R = root_bound_s<K>(p);
    // crit[iv - 1] and crit[iv] of this lane's interval in each round (nr_prev >= RG here)
    const double below0 = __shfl_up(crit0, 1);                    // crit[j - 1]
    const double c7 = __shfl(crit0, gbase + RG - 1), c8 = __shfl(crit1, gbase);
    double root[2] = {0., 0.};
    bool has[2] = {false, false};
#pragma unroll 1
    for (int rnd = 0; rnd < 2; ++rnd) {
        const int iv = rnd * RG + j;
        double a = rnd == 0 ? (j == 0 ? -R : below0) : (j == 0 ? c7 : c8);
        double b = (iv == nr_prev) ? R : (rnd == 0 ? crit0 : crit1);
        if (a < -R) a = -R;
        if (b > R) b = R;
        bool h = false;
        double r = 0.;
        if (iv <= nr_prev && a < b) {
            const int sa = horner_s<K>(p, a) > 0., sb = horner_s<K>(p, b) > 0.;
            if (sa != sb) { r = refine_root_s<K>(p, dp, a, b, sa); h = true; }
        }
        if (rnd == 0) { root[0] = r; has[0] = h; } else { root[1] = r; has[1] = h; }
    }
    const unsigned mA = (unsigned)(__ballot(has[0]) >> gbase) & ((1u << RG) - 1u);
    const unsigned mB = (unsigned)(__ballot(has[1]) >> gbase) & ((1u << RG) - 1u);
    const int nA = __popc(mA), nB = __popc(mB);
    // root #r of this level: the r-th found in round 0, then those of round 1
    const double a0 = __shfl(root[0], gbase + nth_set_bit(mA, j)), b0 = __shfl(root[1], gbase + nth_set_bit(mB, max(j - nA, 0)));
    const double a1 = __shfl(root[0], gbase + nth_set_bit(mA, min(j + RG, 31))), b1 = __shfl(root[1], gbase + nth_set_bit(mB, max(j + RG - nA, 0)));
    crit0 = j < nA ? a0 : b0;
    crit1 = j + RG < nA ? a1 : b1;
    return nA + nB;
}

#define ROOTS_WS (GEN_WS_DOUBLES + 11 + 10)      // generic-path workspace + coefficients + root list, per group
__global__ __launch_bounds__(256, 4) void ransac_roots_kernel(const RpeRansacState *__restrict__ st, const double *__restrict__ hyp,
                                                               double *__restrict__ models, int *__restrict__ nmodels, int n_pairs,
                                                               int per_pair /* samples of a pair in this launch: the chunk */)
{
    __shared__ double s_gen[256 / RG][ROOTS_WS];
    const int tid = threadIdx.x, j = tid & (RG - 1), gbase = tid & 63 & ~(RG - 1);
    const long long sidx = (long long)blockIdx.x * (256 / RG) + (tid / RG);     // sample = (pair, wv, lane)
    const int pair = (int)(sidx / per_pair), rem = (int)(sidx - (long long)pair * per_pair);
    const int wv = rem >> 6, lane = rem & 63;
    if (pair >= n_pairs) return;
    const RpeRansacState s = st[pair];
    if (s.done || s.next_iter + wv * 64 >= s.niters) return;          // uniform per workgroup
    const long long slot = (long long)pair * RPE_RANSAC_MAXCHUNK + wv * 64 + lane;
    if (nmodels[slot] != -1) return;                                   // uniform per group
    const double *rec = hyp + ((long long)pair * (RPE_RANSAC_MAXCHUNK / 64) + wv) * HYP_DOUBLES * 64 + lane;
    double c10[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) c10[i] = rec[i * 64];
    const int n = (int)rec[86 * 64];
    double z = 0., z1 = 0.;            // roots #j and #(RG + j) of the current level
    int nroots = 0;
    const bool generic = (n != 10);
    if (!generic) {
        {
            double p[11];
#pragma unroll
            for (int i = 0; i <= 10; ++i) p[i] = c10[i];
#pragma unroll
            for (int kk = 10; kk > 1; --kk)
#pragma unroll
                for (int i = 0; i < kk; ++i) p[i] = p[i + 1] * (double)(i + 1);
            z = -p[0] / p[1];
        }
        nroots = 1;
        // a level has nroots + 1 intervals; levels 2..8 cannot exceed the 8 lanes (nroots <= K - 1 <= 7 below level K <= 8),
        // levels 9 and 10 take a second round when they do
        nroots = roots_level_grp<2>(c10, z, nroots, j, gbase);
        nroots = roots_level_grp<3>(c10, z, nroots, j, gbase);
        nroots = roots_level_grp<4>(c10, z, nroots, j, gbase);
        nroots = roots_level_grp<5>(c10, z, nroots, j, gbase);
        nroots = roots_level_grp<6>(c10, z, nroots, j, gbase);
        nroots = roots_level_grp<7>(c10, z, nroots, j, gbase);
        nroots = roots_level_grp<8>(c10, z, nroots, j, gbase);
        if (nroots + 1 > RG) nroots = roots_level_grp2<9>(c10, z, z1, nroots, j, gbase);       // group-uniform
        else nroots = roots_level_grp<9>(c10, z, nroots, j, gbase);
        if (nroots + 1 > RG) nroots = roots_level_grp2<10>(c10, z, z1, nroots, j, gbase);
        else nroots = roots_level_grp<10>(c10, z, nroots, j, gbase);
    }
    double *ws = s_gen[tid / RG];
    if (generic) {
        // the group leader runs the whole chain serially in its LDS workspace (groups are wave-local: the leader's LDS
        // writes are visible to its neighbours after the wave barrier)
        int nr = 0;
        if (j == 0) {
#pragma unroll
            for (int t = 0; t < 11; ++t) ws[GEN_WS_DOUBLES + t] = c10[t];
            nr = poly_real_roots_generic(ws + GEN_WS_DOUBLES, n, ws + GEN_WS_DOUBLES + 11, ws);
        }
        __builtin_amdgcn_wave_barrier();
        nroots = __shfl(nr, gbase);
    }
    // back-substitution: lane j <- root #j (five_point_roots' loop body), models compacted in root order; more than RG
    // roots (generic path only) take a second round
    int nmod = 0;
    for (int r0 = 0; r0 < nroots; r0 += RG) {                          // group-uniform trip count (1, rarely 2)
        const int ri = r0 + j;
        __asm__ volatile("" ::: "memory");     // keep the ~75 record loads inside the round: hoisted out of the loop they spill
        if (generic) { if (ri < nroots) z = ws[GEN_WS_DOUBLES + 11 + ri]; }
        else if (r0) z = z1;
        bool okm = false;
        double Ev[9];
        if (ri < nroots) {
            double bz[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double x3 = rec[(11 + i * 4 + 3) * 64], x2 = rec[(11 + i * 4 + 2) * 64], x1 = rec[(11 + i * 4 + 1) * 64], x0 = rec[(11 + i * 4) * 64];
                const double y3 = rec[(23 + i * 4 + 3) * 64], y2 = rec[(23 + i * 4 + 2) * 64], y1 = rec[(23 + i * 4 + 1) * 64], y0 = rec[(23 + i * 4) * 64];
                const double w4 = rec[(35 + i * 5 + 4) * 64], w3 = rec[(35 + i * 5 + 3) * 64], w2 = rec[(35 + i * 5 + 2) * 64],
                             w1 = rec[(35 + i * 5 + 1) * 64], w0 = rec[(35 + i * 5) * 64];
                bz[i][0] = ((x3 * z + x2) * z + x1) * z + x0;
                bz[i][1] = ((y3 * z + y2) * z + y1) * z + y0;
                bz[i][2] = (((w4 * z + w3) * z + w2) * z + w1) * z + w0;
            }
            double bestn = -1., xv0 = 0., xv1 = 0., xv2 = 0.;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q0 = i, q1 = (i + 1) % 3;
                const double cx = bz[q0][1] * bz[q1][2] - bz[q0][2] * bz[q1][1];
                const double cy = bz[q0][2] * bz[q1][0] - bz[q0][0] * bz[q1][2];
                const double cz = bz[q0][0] * bz[q1][1] - bz[q0][1] * bz[q1][0];
                const double nn = cx * cx + cy * cy + cz * cz;
                if (nn > bestn) { bestn = nn; xv0 = cx; xv1 = cy; xv2 = cz; }
            }
            if (bestn > 0.) {
                const double inv = 1. / sqrt(bestn);
                const double w = xv2 * inv;
                if (!(fabs(w) < 1e-10)) {
                    const double x = xv0 / xv2, y = xv1 / xv2;
                    double nrm = 0.;
#pragma unroll
                    for (int e = 0; e < 9; ++e) {
                        Ev[e] = ((rec[(50 + e) * 64] * x + rec[(59 + e) * 64] * y) + rec[(68 + e) * 64] * z) + rec[(77 + e) * 64];
                        nrm += Ev[e] * Ev[e];
                    }
                    nrm = sqrt(nrm);
                    if (nrm > 0.) {
                        okm = true;
#pragma unroll
                        for (int e = 0; e < 9; ++e) Ev[e] = Ev[e] / nrm;
                    }
                }
            }
        }
        const unsigned m = (unsigned)(__ballot(okm) >> gbase) & ((1u << RG) - 1u);
        if (okm) {
            double *dst = models + (slot * RPE_MAX_MODELS + nmod + __popc(m & ((1u << j) - 1u))) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) dst[e] = Ev[e];
        }
        nmod += __popc(m);
    }
    if (j == 0) nmodels[slot] = nmod;
}

// ------------------------------------------------------- Sampson inlier test
// EMEstimatorCallback::computeError + findInliers: (float)err <= (float)(thr*thr)
// The same predicate without the f64 division (a ~30-instruction IEEE sequence in the innermost RANSAC loop).
// (float)(num / den) <= thr2  <=>  fl64(num / den) <= B, B = the largest double that still rounds (to nearest even) to a
// float <= thr2.  num <= 0.999.. * fl(B * den) proves the left side, num >= 1.000.. * fl(B * den) disproves it (margins
// 2^-40, far above the 2^-53 rounding of the product and of the quotient); only the sliver in between (and den <= 0 /
// non-finite values) takes the exact division.  Bit-identical to sampson_inlier by construction.
struct SampsonBound { double B; };
__device__ __forceinline__ SampsonBound sampson_bound(float thr2)
{
    const unsigned u = __float_as_uint(thr2);
    const float nf = __uint_as_float(u + 1u);                             // next float above thr2 (thr2 > 0, finite)
    const double m = 0.5 * ((double)thr2 + (double)nf);                   // exact midpoint: ties go to the even float
    SampsonBound b;
    b.B = (u & 1u) ? __longlong_as_double(__double_as_longlong(m) - 1) : m;
    return b;
}
__device__ __forceinline__ int sampson_inlier_fast(const double *E, double x1, double y1, double x2, double y2, float thr2, SampsonBound sb)
{
    double Ex0 = (E[0] * x1 + E[1] * y1) + E[2];
    double Ex1 = (E[3] * x1 + E[4] * y1) + E[5];
    double Ex2 = (E[6] * x1 + E[7] * y1) + E[8];
    double Et0 = (E[0] * x2 + E[3] * y2) + E[6];
    double Et1 = (E[1] * x2 + E[4] * y2) + E[7];
    double x2tEx1 = (x2 * Ex0 + y2 * Ex1) + Ex2;
    double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    const double num = x2tEx1 * x2tEx1, den = ((a + b) + c) + d;
    const double p = sb.B * den;
    if (den > 0. && p < 1e300) {
        if (num <= p * (1. - 0x1p-40)) return 1;
        if (num >= p * (1. + 0x1p-40)) return 0;
    }
    return (float)(num / den) <= thr2;
}

__device__ __forceinline__ int sampson_inlier(const double *E, double x1, double y1, double x2, double y2, float thr2)
{
    double Ex0 = (E[0] * x1 + E[1] * y1) + E[2];
    double Ex1 = (E[3] * x1 + E[4] * y1) + E[5];
    double Ex2 = (E[6] * x1 + E[7] * y1) + E[8];
    double Et0 = (E[0] * x2 + E[3] * y2) + E[6];
    double Et1 = (E[1] * x2 + E[4] * y2) + E[7];
    double x2tEx1 = (x2 * Ex0 + y2 * Ex1) + Ex2;
    double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    float err = (float)(x2tEx1 * x2tEx1 / (((a + b) + c) + d));
    return err <= thr2;
}

// RANSACUpdateNumIters with log() terms tabulated on the host per (M, goodCount)
__device__ __forceinline__ int update_niters(const double *nit_denom, const int *nit_round, double num, int M, int good, int niters)
{
    long long idx = (long long)M * (M + 1) / 2 + good;
    int r = nit_round[idx];
    if (r == -1) return 0;
    double denom = nit_denom[idx];
    return (denom >= 0 || -num >= niters * (-denom)) ? niters : r;
}

// ------------------------------------------------------------------ score
// One workgroup per (pair, group of SCORE_GROUP iterations): a wave scores its models one after the other, so the group
// size sets the kernel's latency (the later RANSAC rounds run few pairs and are pure latency): 16 iterations per
// workgroup = 4x the workgroups of the 64-iteration grouping, each a quarter as long.  The K-normalised matches sit in LDS; each of
// the 4 waves scores a different model (lanes stride over the matches, Sampson error f64 -> f32
// compare, wave-shuffle popcount), so there is no cross-wave reduction.  Counts go to HBM.
__global__ __launch_bounds__(256) void ransac_score_kernel(const double2 *__restrict__ n1, const double2 *__restrict__ n2,
                                                            const RpeRansacState *__restrict__ st, const double *__restrict__ models,
                                                            const int *__restrict__ nmodels, const double *__restrict__ K,
                                                            double threshold, int *__restrict__ counts, int max_matches, int use_lds)
{
    extern __shared__ double2 s_pts[];              // [2][max_matches] when the points fit LDS (use_lds)
    __shared__ int s_nm[SCORE_GROUP], s_first[SCORE_GROUP + 1];
    const int pair = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const RpeRansacState s = st[pair];
    if (s.done || s.M <= 5 || s.next_iter + grp * SCORE_GROUP >= s.niters) return;
    const int M = s.M;
    // max_matches > 2048 ("no truncation" configurations): 32 B per match no longer fit the 64 KB of dynamic LDS;
    // the points are then read from HBM / L2 directly (every wave walks the same 64 KB-scale array)
    const double2 *sp1 = n1 + (long long)pair * max_matches, *sp2 = n2 + (long long)pair * max_matches;
    if (use_lds) {
        double2 *l1 = s_pts, *l2 = s_pts + max_matches;
        for (int i = tid; i < M; i += 256) { l1[i] = sp1[i]; l2[i] = sp2[i]; }
        sp1 = l1; sp2 = l2;
    }
    const long long slot0 = (long long)pair * RPE_RANSAC_MAXCHUNK + grp * SCORE_GROUP;
    if (tid < SCORE_GROUP) s_nm[tid] = max(nmodels[slot0 + tid], 0);
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int k = 0; k < SCORE_GROUP; ++k) { s_first[k] = acc; acc += s_nm[k]; } s_first[SCORE_GROUP] = acc; }
    __syncthreads();
    const double fx = K[0], fy = K[4];
    const double thr = threshold / ((fx + fy) / 2);
    const float thr2 = (float)(thr * thr);
    const SampsonBound sbound = sampson_bound(thr2);
    const int total = s_first[SCORE_GROUP];
    // a wave scores models j = wv, wv + 4, ... one after the other; the 72 bytes of the NEXT model are fetched (lanes 0..8,
    // one double each) before the current one is scored, so the HBM / L2 round trip of the model hides behind ~400
    // instructions of Sampson arithmetic instead of heading every iteration (the kernel ran at 28 % of its issue roof)
    int k = 0, kn = 0;
    double e_next = 0.;
    int j = wv;
    if (j < total) {
        while (s_first[kn + 1] <= j) ++kn;
        if (lane < 9) e_next = models[((slot0 + kn) * RPE_MAX_MODELS + (j - s_first[kn])) * 9 + lane];
    }
    for (; j < total; j += 4) {                      // j-th model of this group, flattened (iteration-major)
        k = kn;
        const int m = j - s_first[k];
        const double e_cur = e_next;
        const int jn = j + 4;
        if (jn < total) {
            while (s_first[kn + 1] <= jn) ++kn;
            if (lane < 9) e_next = models[((slot0 + kn) * RPE_MAX_MODELS + (jn - s_first[kn])) * 9 + lane];
        }
        double E[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) E[e] = __shfl(e_cur, e);
        int cnt = 0;
        for (int i = lane; i < M; i += 64) {
            double2 a = sp1[i], b = sp2[i];
            cnt += sampson_inlier_fast(E, a.x, a.y, b.x, b.y, thr2, sbound);
        }
        cnt = wave_sum(cnt);                             // DPP row shifts + v_readlane: 8 instructions instead of 6 ds_bpermute rounds
        if (lane == 0) counts[(slot0 + k) * RPE_MAX_MODELS + m] = cnt;
    }
}

// Replay of OpenCV's sequential update rule (ptsetreg.cpp) over the counts of one chunk: "strictly
// more inliers wins", niters shrinks through RANSACUpdateNumIters, the loop stops at niters.
// RANSACUpdateNumIters(.., niters) == min(niters, R(good)) with R tabulated on the host (0 when the
// log underflows, unbounded when denom >= 0), so the serial recurrence is a pair of scans: one wave per
// pair, lane = iteration, 64 iterations per step: exclusive prefix max of the per-iteration best count
// (who is a record breaker), per-lane replay of its <= 10 models against that prefix, exclusive prefix
// min of the resulting niters, first lane whose iteration index reaches its niters = the break.
// Bit-identical to the serial loop (it was 0.25-1.1 ms of single-lane latency per step).
__device__ __forceinline__ int niters_cap(const double *nit_denom, const int *nit_round, double num, int M, int good, int niters)
{
    return update_niters(nit_denom, nit_round, num, M, good, niters);
}

__global__ __launch_bounds__(256) void ransac_update_kernel(RpeRansacState *__restrict__ st, const double *__restrict__ models,
                                                            const int *__restrict__ nmodels, const int *__restrict__ counts,
                                                            const double *__restrict__ nit_denom, const int *__restrict__ nit_round,
                                                            double nit_num, double *__restrict__ E_out, int *__restrict__ found,
                                                            int chunk, int n_pairs)
{
    const int lane = threadIdx.x & 63, pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    RpeRansacState s = st[pair];
    if (s.done) return;                                   // wave-uniform
    const int M = s.M;
    const long long slot0 = (long long)pair * RPE_RANSAC_MAXCHUNK;
    int best = s.best_count, niters = s.niters, bk = -1, bm = -1;
    int nstack = 0;
    if (M == 5) {
        // ptsetreg.cpp: count == modelPoints -> runKernel on all points; cv2 returns EVERY model stacked (3n x 3).
        // One model = a usable E; more than one is reported (found = n) and becomes RPE_PAIR_AMBIGUOUS_ESSENTIAL
        // (the reference's recoverPose call then fails its 3x3 assertion, pose_estimator.py:533); E keeps the first.
        nstack = nmodels[slot0];
        if (nstack > 0) { best = 5; bk = 0; bm = 0; s.best_iter = 0; s.best_model = 0; }
        niters = 1;
        s.next_iter = 1;
        s.iters_run = 1;
    } else {
        const int kmax = min(chunk, s.niters - s.next_iter);
        bool stopped = false;
        for (int k0 = 0; k0 < kmax && !stopped; k0 += 64) {
            const int k = k0 + lane, it = s.next_iter + k;
            const bool act = k < kmax;
            const int nm = act ? nmodels[slot0 + k] : 0;
            int c[RPE_MAX_MODELS];
            int cmax = 0;
#pragma unroll
            for (int m = 0; m < RPE_MAX_MODELS; ++m) {
                c[m] = (m < nm) ? counts[(slot0 + k) * RPE_MAX_MODELS + m] : 0;
                if (c[m] > 4) cmax = max(cmax, c[m]);          // a model needs > max(best, 4) inliers to count
            }
            // exclusive prefix max over the lanes (iterations), seeded with the incoming best
            const int inc = wave_inclusive_max(cmax);
            int pre = __shfl_up(inc, 1);
            if (lane == 0) pre = 0;
            pre = max(pre, best);
            // this lane's models against its prefix: local record breakers shrink niters
            int run = pre, lmin = 0x7FFFFFFF, wm = -1;
#pragma unroll
            for (int m = 0; m < RPE_MAX_MODELS; ++m) {
                if (m < nm && c[m] > max(run, 4)) {
                    run = c[m]; wm = m;
                    lmin = min(lmin, niters_cap(nit_denom, nit_round, nit_num, M, c[m], 0x7FFFFFFF));
                }
            }
            // exclusive prefix min of niters
            const int pmin = wave_inclusive_min(lmin);
            int nit_before = __shfl_up(pmin, 1);
            if (lane == 0) nit_before = 0x7FFFFFFF;
            nit_before = min(nit_before, niters);
            const unsigned long long brk = __ballot(!act || it >= nit_before);
            const int stop = brk ? (__ffsll((long long)brk) - 1) : 64;       // lanes [0, stop) are processed
            if (stop > 0) {
                // state after the last processed lane
                const int last = stop - 1;
                const int nbest = __shfl(run, last);
                const int nnit = min(niters, __shfl(pmin, last));
                if (nbest > best) {
                    // the winner is the first processed lane whose run reached nbest (first occurrence of the max)
                    const unsigned long long wmask = __ballot(lane < stop && run == nbest && wm >= 0);
                    const int wl = __ffsll((long long)wmask) - 1;
                    bk = k0 + wl; bm = __shfl(wm, wl);
                    s.best_iter = s.next_iter + bk; s.best_model = bm;
                    best = nbest;
                }
                niters = nnit;
                s.iters_run = s.next_iter + k0 + stop;
            }
            if (stop < 64) stopped = true;
        }
        s.next_iter += chunk;
    }
    if (bk >= 0 && lane < 9) {
        const double *Eg = models + ((slot0 + bk) * RPE_MAX_MODELS + bm) * 9;
        const double e = Eg[lane];
        E_out[pair * 9 + lane] = e;
        st[pair].E[lane] = e;
    }
    if (lane == 0) {
        RpeRansacState *d = st + pair;
        d->best_count = best; d->niters = niters; d->best_iter = s.best_iter; d->best_model = s.best_model;
        d->next_iter = s.next_iter; d->iters_run = s.iters_run;
        const int fnd = (M == 5) ? nstack : (best > 0 ? 1 : 0);
        d->found = fnd;
        d->done = s.next_iter >= niters;
        found[pair] = fnd;
    }
}

// ------------------------------------------------------------------- mask
__global__ __launch_bounds__(256) void ransac_mask_kernel(const double2 *__restrict__ n1, const double2 *__restrict__ n2,
                                                           const RpeRansacState *__restrict__ st, const double *__restrict__ K,
                                                           double threshold, uint8_t *__restrict__ mask, int max_matches)
{
    const int pair = blockIdx.x, tid = threadIdx.x;
    const RpeRansacState s = st[pair];
    const double fx = K[0], fy = K[4];
    const double thr = threshold / ((fx + fy) / 2);
    const float thr2 = (float)(thr * thr);
    for (int i = tid; i < max_matches; i += 256) {
        uint8_t v = 0;
        if (i < s.M && s.found) {
            if (s.M == 5) v = 1;
            else {
                double2 a = n1[(long long)pair * max_matches + i], b = n2[(long long)pair * max_matches + i];
                v = (uint8_t)sampson_inlier(s.E, a.x, a.y, b.x, b.y, thr2);
            }
        }
        mask[(long long)pair * max_matches + i] = v;
    }
}

void rpe_launch_ransac(rpe_handle *h, int B, bool want_mask)
{
    const int mm = h->cfg.max_matches, it = h->cfg.ransac_max_iters;
    double2 *n1 = h->d_n1, *n2 = h->d_n2;
    hipLaunchKernelGGL(ransac_prepare_kernel, dim3((mm + 255) / 256, B), dim3(256), 0, h->stream,
                       h->d_pts1, h->d_pts2, h->d_m_n, h->d_K, n1, n2, h->d_rstate, h->d_found, mm, it);
    // chunk schedule 32, 96, 384, 512, 512, ... (cumulative 32, 128, 512, 1024).  A launch group costs
    //   max(latency floor, work): the floor is one wave's dependent chain through poly -> roots -> score -> update
    //   (~170-200 us whatever the number of pairs still running), the work is ~12.6 ns per (pair, iteration) that is
    //   still below its pair's current niters (r02 trace: 414 us for 1024 pairs x 32 iterations).
    // RANSACUpdateNumIters ends most pairs early (inlier ratio 0.72 -> 32 iterations, 0.64 -> 64), so the first group is
    // small (its whole chunk is evaluated for every pair); after it the launches are floor-bound and every extra group
    // costs a floor, so the chunks grow fast.  Measured on the 1024-pair bench batch (same box, interleaved):
    //   64,64,128,256,512: 1.58 ms   32,32,64,128,256,512: 1.43   32,32,64,384,512: 1.29   32,96,384,512: 1.20
    //   32,480,512: 1.23   64,448,512: 1.26   96,416,512: 1.40   (results identical: the replay in ransac_update_kernel
    //   is exact for any chunking).
    const int use_lds = mm <= 2048;
    const size_t lds = use_lds ? sizeof(double2) * 2 * (size_t)mm : 0;
    static_assert(RPE_RANSAC_FIRST_CHUNK % POLY_LANES == 0 && RPE_RANSAC_FIRST_CHUNK % SCORE_GROUP == 0 && (RPE_RANSAC_FIRST_CHUNK * RG) % 256 == 0, "chunk granularity");
    // experiment hook: RPE_RANSAC_SCHEDULE="32,96,384,512" replaces the chunk schedule (multiples of 32, <= 512; the last
    // entry repeats)
    static std::vector<int> env_sched = [] {
        std::vector<int> v;
        if (const char *e = getenv("RPE_RANSAC_SCHEDULE")) {
            for (const char *q = e; *q;) { const int c = atoi(q); if (c >= 32 && c <= RPE_RANSAC_MAXCHUNK && c % 32 == 0) v.push_back(c); while (*q && *q != ',') ++q; if (*q) ++q; }
        }
        return v;
    }();
    static const int kSchedule[] = {RPE_RANSAC_FIRST_CHUNK, 96, 384, RPE_RANSAC_MAXCHUNK};
    int done_iters = 0, chunk = env_sched.empty() ? kSchedule[0] : env_sched[0], nlaunch = 0;
    while (done_iters < it) {
        hipLaunchKernelGGL(ransac_poly_kernel, dim3(B, chunk / POLY_LANES), dim3(POLY_LANES), 0, h->stream,
                           n1, n2, h->d_rstate, h->d_subsets, h->d_hyp, h->d_nmodels, mm, it);
        hipLaunchKernelGGL(ransac_roots_kernel, dim3((unsigned)((long long)B * chunk * RG / 256)), dim3(256), 0, h->stream,
                           (const RpeRansacState *)h->d_rstate, (const double *)h->d_hyp, h->d_models, h->d_nmodels, B, chunk);
        hipLaunchKernelGGL(ransac_score_kernel, dim3(B, chunk / SCORE_GROUP), dim3(256), lds, h->stream,
                           n1, n2, (const RpeRansacState *)h->d_rstate, (const double *)h->d_models, (const int *)h->d_nmodels,
                           (const double *)h->d_K, h->cfg.ransac_threshold, h->d_counts, mm, use_lds);
        hipLaunchKernelGGL(ransac_update_kernel, dim3((B + 3) / 4), dim3(256), 0, h->stream,
                           h->d_rstate, (const double *)h->d_models, (const int *)h->d_nmodels, (const int *)h->d_counts,
                           (const double *)h->d_nit_denom, (const int *)h->d_nit_round, h->nit_num, h->d_E, h->d_found, chunk, B);
        done_iters += chunk;
        ++nlaunch;
        if (!env_sched.empty()) chunk = env_sched[std::min((size_t)nlaunch, env_sched.size() - 1)];
        else chunk = kSchedule[std::min(nlaunch, 3)];
    }
    if (want_mask)
        hipLaunchKernelGGL(ransac_mask_kernel, dim3(B), dim3(256), 0, h->stream,
                           n1, n2, h->d_rstate, h->d_K, h->cfg.ransac_threshold, h->d_mask, mm);
}

// ------------------------------------------------------------ recoverPose
// one-sided Jacobi (Hestenes) on the columns of A (M x N row-major); V accumulates rotations.
template <int MM, int NN>
__device__ __forceinline__ void jacobi_cols(double *A, double *V)
{
    const double eps = DBL_EPSILON * 10;
#pragma unroll
    for (int i = 0; i < NN; ++i)
#pragma unroll
        for (int j = 0; j < NN; ++j) V[i * NN + j] = (i == j) ? 1. : 0.;
    for (int sweep = 0; sweep < 30; ++sweep) {
        int changed = 0;
#pragma unroll
        for (int p = 0; p < NN - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < NN; ++q) {
                double al = 0., be = 0., ga = 0.;
#pragma unroll
                for (int k = 0; k < MM; ++k) {
                    double ap = A[k * NN + p], aq = A[k * NN + q];
                    al += ap * ap; be += aq * aq; ga += ap * aq;
                }
                if (!(fabs(ga) <= eps * sqrt(al * be))) {
                    changed = 1;
                    double zeta = (be - al) / (2. * ga);
                    double t = (zeta >= 0. ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
                    double c = 1. / sqrt(1. + t * t), s = c * t;
#pragma unroll
                    for (int k = 0; k < MM; ++k) {
                        double ap = A[k * NN + p], aq = A[k * NN + q];
                        A[k * NN + p] = c * ap - s * aq; A[k * NN + q] = s * ap + c * aq;
                    }
#pragma unroll
                    for (int k = 0; k < NN; ++k) {
                        double vp = V[k * NN + p], vq = V[k * NN + q];
                        V[k * NN + p] = c * vp - s * vq; V[k * NN + q] = s * vp + c * vq;
                    }
                }
            }
        if (!changed) break;
    }
}

__device__ __forceinline__ double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// decomposeEssentialMat (five-point.cpp)
__device__ static void decompose_essential(const double *E, double *R1, double *R2, double *t)
{
    double A[9], V[9];
    for (int i = 0; i < 9; ++i) A[i] = E[i];
    jacobi_cols<3, 3>(A, V);
    double sv[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) sv[j] = sqrt((A[j] * A[j] + A[3 + j] * A[3 + j]) + A[6 + j] * A[6 + j]);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j)
        if (sv[ord[j]] < sv[ord[j + 1]]) { int tt = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = tt; }
    double U[9], Vt[9];
    for (int c = 0; c < 2; ++c) {
        int j = ord[c];
        double s = sv[j] > 0. ? 1. / sv[j] : 0.;
        for (int r = 0; r < 3; ++r) U[r * 3 + c] = A[r * 3 + j] * s;
    }
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Vt[c * 3 + r] = V[r * 3 + ord[c]];
    if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (det3(Vt) < 0) for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    double UW[9], UWt[9];
    for (int r = 0; r < 3; ++r) {
        UW[r * 3 + 0] = -U[r * 3 + 1]; UW[r * 3 + 1] = U[r * 3 + 0]; UW[r * 3 + 2] = U[r * 3 + 2];
        UWt[r * 3 + 0] = U[r * 3 + 1]; UWt[r * 3 + 1] = -U[r * 3 + 0]; UWt[r * 3 + 2] = U[r * 3 + 2];
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        R1[r * 3 + c] = (UW[r * 3] * Vt[c] + UW[r * 3 + 1] * Vt[3 + c]) + UW[r * 3 + 2] * Vt[6 + c];
        R2[r * 3 + c] = (UWt[r * 3] * Vt[c] + UWt[r * 3 + 1] * Vt[3 + c]) + UWt[r * 3 + 2] * Vt[6 + c];
    }
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

// triangulate.cpp DLT with P0 = [I|0], P = [R|t]; cheirality test of recoverPose (dist 50)
__device__ static int cheirality_one(const double *R, const double *t, double x1, double y1, double x2, double y2)
{
    double A[16], V[16];
    A[0] = -1.; A[1] = 0.;  A[2] = x1; A[3] = 0.;
    A[4] = 0.;  A[5] = -1.; A[6] = y1; A[7] = 0.;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[8 + k]  = x2 * R[6 + k] - R[k];
        A[12 + k] = y2 * R[6 + k] - R[3 + k];
    }
    A[11] = x2 * t[2] - t[0];
    A[15] = y2 * t[2] - t[1];
    jacobi_cols<4, 4>(A, V);
    double X = 0., Y = 0., Z = 0., W = 0., best = 0.;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double nn = ((A[j] * A[j] + A[4 + j] * A[4 + j]) + A[8 + j] * A[8 + j]) + A[12 + j] * A[12 + j];
        if (j == 0 || nn < best) { best = nn; X = V[j]; Y = V[4 + j]; Z = V[8 + j]; W = V[12 + j]; }
    }
    int good = (Z * W) > 0.;
    X /= W; Y /= W; Z /= W;
    good = good && (Z < 50.);
    double z2 = ((R[6] * X + R[7] * Y) + R[8] * Z) + t[2];
    good = good && (z2 > 0.) && (z2 < 50.);
    return good;
}

__global__ __launch_bounds__(256) void recover_pose_kernel(const double *__restrict__ Eall, const float2 *__restrict__ pts1,
                                                            const float2 *__restrict__ pts2, const int *__restrict__ m_n,
                                                            const int *__restrict__ found, const int *__restrict__ kp_count,
                                                            int img2_base, const double *__restrict__ K,
                                                            double *__restrict__ Rout, double *__restrict__ tout,
                                                            int *__restrict__ inliers, int *__restrict__ status, int max_matches)
{
    __shared__ int s_g[4];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int M = min(m_n[pair], max_matches);
    int stt = RPE_PAIR_OK;
    if (kp_count && (kp_count[pair] == 0 || kp_count[img2_base + pair] == 0)) stt = RPE_PAIR_NO_DESCRIPTORS;
    else if (M < 5) stt = RPE_PAIR_INSUFFICIENT_MATCHES;
    else if (found && !found[pair]) stt = RPE_PAIR_NO_ESSENTIAL;
    else if (found && found[pair] > 1) stt = RPE_PAIR_AMBIGUOUS_ESSENTIAL;
    if (stt != RPE_PAIR_OK) {
        if (tid < 9) Rout[pair * 9 + tid] = (tid % 4 == 0) ? 1. : 0.;
        if (tid < 3) tout[pair * 3 + tid] = 0.;
        if (tid == 0) { inliers[pair] = 0; if (status) status[pair] = stt; }
        return;
    }
    if (tid < 4) s_g[tid] = 0;
    __syncthreads();
    double E[9], R1[9], R2[9], tt[3], tn[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) E[e] = Eall[pair * 9 + e];
    decompose_essential(E, R1, R2, tt);
    tn[0] = -tt[0]; tn[1] = -tt[1]; tn[2] = -tt[2];
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    int g1 = 0, g2 = 0, g3 = 0, g4 = 0;
    for (int i = tid; i < M; i += 256) {
        float2 a = pts1[(long long)pair * max_matches + i], b = pts2[(long long)pair * max_matches + i];
        double x1 = ((double)a.x - cx) / fx, y1 = ((double)a.y - cy) / fy;
        double x2 = ((double)b.x - cx) / fx, y2 = ((double)b.y - cy) / fy;
        g1 += cheirality_one(R1, tt, x1, y1, x2, y2);
        g2 += cheirality_one(R2, tt, x1, y1, x2, y2);
        g3 += cheirality_one(R1, tn, x1, y1, x2, y2);
        g4 += cheirality_one(R2, tn, x1, y1, x2, y2);
    }
    g1 = wave_sum(g1); g2 = wave_sum(g2); g3 = wave_sum(g3); g4 = wave_sum(g4);
    if ((tid & 63) == 0) { atomicAdd(&s_g[0], g1); atomicAdd(&s_g[1], g2); atomicAdd(&s_g[2], g3); atomicAdd(&s_g[3], g4); }
    __syncthreads();
    if (tid == 0) {
        g1 = s_g[0]; g2 = s_g[1]; g3 = s_g[2]; g4 = s_g[3];
        const double *Rs, *ts; int g;
        if (g1 >= g2 && g1 >= g3 && g1 >= g4)      { Rs = R1; ts = tt; g = g1; }
        else if (g2 >= g1 && g2 >= g3 && g2 >= g4) { Rs = R2; ts = tt; g = g2; }
        else if (g3 >= g1 && g3 >= g2 && g3 >= g4) { Rs = R1; ts = tn; g = g3; }
        else                                        { Rs = R2; ts = tn; g = g4; }
        for (int e = 0; e < 9; ++e) Rout[pair * 9 + e] = Rs[e];
        for (int e = 0; e < 3; ++e) tout[pair * 3 + e] = ts[e];
        inliers[pair] = g;
        if (status) status[pair] = RPE_PAIR_OK;
    }
}

void rpe_launch_pose(rpe_handle *h, int B, bool fused)
{
    hipLaunchKernelGGL(recover_pose_kernel, dim3(B), dim3(256), 0, h->stream,
                       h->d_E, h->d_pts1, h->d_pts2, h->d_m_n, fused ? h->d_found : (const int *)nullptr,
                       fused ? h->d_kp_count : (const int *)nullptr, h->img2_base ? h->img2_base : B, h->d_K,
                       h->d_R, h->d_t, h->d_inliers, h->d_status, h->cfg.max_matches);
}
