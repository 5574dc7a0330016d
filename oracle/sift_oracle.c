/*
 * sift_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of
 *   cv2.SIFT_create().detectAndCompute(image, None)
 *   (reference src/core/pose_estimator.py:93-94, :108; BASELINE config 3 adds a
 *   keypoint cap = SIFT_create(nfeatures), an extension over the reference).
 *
 * Follows OpenCV 4.x features2d/sift.dispatch.cpp + sift.simd.hpp: createInitialImage
 * (2x INTER_LINEAR upsample + blur to sigma 1.6 assuming 0.5), buildGaussianPyramid
 * (nOctaveLayers 3, incremental sigmas, INTER_NEAREST octave halving), buildDoGPyramid,
 * findScaleSpaceExtrema (threshold floor(0.5*0.04/3*255)=1, border 5, 26 neighbours,
 * adjustLocalExtrema <= 5 steps, contrast 0.04, edge 10), calcOrientationHist (radius
 * round(4.5 s), sigma 1.5 s, 36 bins, smoothing, peaks >= 0.8 max, parabolic
 * interpolation), removeDuplicatedSorted, retainBest, calcSIFTDescriptor (4x4x8,
 * 3 s bins, trilinear, clamp 0.2, x512, saturate to u8).
 *
 * Own conventions (f32 results of cv2 depend on its SIMD summation orders and libm and
 * cannot be reproduced offline; "parity unpinned" vs cv2):
 *  - exp / cos / sin come from deterministic f64 kernels cast to f32; atan2 is cv's
 *    fastAtan2 polynomial; separable Gaussian sums taps in ascending order;
 *  - histogram sums (orientation, descriptor) are accumulated in 8 interleaved partial
 *    sums (window sample k -> partial k % 8, ascending k) reduced by a fixed binary tree --
 *    the order the HIP kernel (one 64-lane wave per keypoint, 8 lanes per LDS round) uses,
 *    so GPU == oracle bit for bit;
 *  - duplicates are removed by final integer location (lowest seed wins) before
 *    retainBest, equivalent to removeDuplicatedSorted for exact duplicates;
 *  - seeds are enumerated in raster order per (octave, layer); workspace caps truncate
 *    in that order.
 */
#include "oracle.h"
#include <math.h>
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define NOL 3                 /* nOctaveLayers */
#define NG (NOL + 3)          /* gaussians per octave */
#define ND (NOL + 2)          /* DoGs per octave */
#define SIFT_BORDER 5
#define ORI_BINS 36
#define MAX_OCT 16

static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_round_f(float v) { return (int)lrintf(v); }

/* ---- deterministic elementary functions (identical on the HIP side) ---- */
static double det_exp_core(double x)
{
    /* exp(x), |x| < 700: n = rint(x / ln2), Taylor degree 11 on the remainder, all f64 */
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00;
    double nf = rint(x * INV_LN2);
    double r = (x - nf * LN2_HI) - nf * LN2_LO;
    double p = 1.0 + r * (1.0 + r * (0.5 + r * (1.0 / 6 + r * (1.0 / 24 + r * (1.0 / 120 + r * (1.0 / 720 + r * (1.0 / 5040 +
               r * (1.0 / 40320 + r * (1.0 / 362880 + r * (1.0 / 3628800 + r * (1.0 / 39916800)))))))))));
    int n = (int)nf;
    union { double d; uint64_t u; } sc;
    sc.u = (uint64_t)(1023 + n) << 52;
    return p * sc.d;
}
static float det_expf(float xf)            /* Gaussian weights, x <= 0 */
{
    if (xf < -87.0f) return 0.f;
    return (float)det_exp_core((double)xf);
}
static float det_exp2f(float t)            /* powf(2.f, t) for the keypoint size, |t| small */
{
    return (float)det_exp_core((double)t * 0.69314718055994530942);
}

static void det_sincos(double x, double *sn, double *cs)   /* x in [0, 2*pi] */
{
    const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    int k = (int)(x * 0.63661977236758134308 + 0.5);
    double r = (x - (double)k * PIO2_HI) - (double)k * PIO2_LO;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double s = r + (r * z) * ps;
    double c = (1.0 - 0.5 * z) + (z * z) * pc;
    switch (k & 3) {
    case 0: *sn = s;  *cs = c;  break;
    case 1: *sn = c;  *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

/* fixed binary-tree reductions of interleaved partial sums (slot l adds slot l+o, o = n/2 .. 1) */
static float tree64(float *p)
{
    for (int o = 32; o > 0; o >>= 1) for (int l = 0; l < o; ++l) p[l] = p[l] + p[l + o];
    return p[0];
}
static float tree8(float *p)
{
    for (int o = 4; o > 0; o >>= 1) for (int l = 0; l < o; ++l) p[l] = p[l] + p[l + o];
    return p[0];
}

/* ---- pyramid geometry ---- */
typedef struct {
    int noct;
    int w[MAX_OCT], h[MAX_OCT];
    size_t goff[MAX_OCT];     /* float offset of gaussian image 0 of the octave */
    size_t gtotal;
    int ksize[NG]; float kern[NG][64];
    float sigma0_k[64]; int ksize0;    /* initial blur (sig_diff) */
} sift_geo;

static int gauss_kernel(double sigma, float *k)
{
    /* getGaussianKernel(ksize = cvRound(sigma*8+1)|1, sigma), f32 coefficients, f64 normalisation */
    int ks = cv_round_d(sigma * 8 + 1) | 1;
    double sum = 0, t[64];
    for (int i = 0; i < ks; ++i) { double x = i - (ks - 1) * 0.5; t[i] = exp(-0.5 * x * x / (sigma * sigma)); sum += t[i]; }
    for (int i = 0; i < ks; ++i) k[i] = (float)(t[i] / sum);
    return ks;
}

static void sift_geometry(int W, int H, sift_geo *g)
{
    int bw = 2 * W, bh = 2 * H;
    int mn = bw < bh ? bw : bh;
    g->noct = cv_round_d(log((double)mn) / log(2.) - 2) + 1;        /* firstOctave = -1 */
    if (g->noct > MAX_OCT) g->noct = MAX_OCT;
    size_t off = 0;
    for (int o = 0; o < g->noct; ++o) {
        g->w[o] = o ? g->w[o - 1] / 2 : bw; g->h[o] = o ? g->h[o - 1] / 2 : bh;
        g->goff[o] = off; off += (size_t)NG * g->w[o] * g->h[o];
    }
    g->gtotal = off;
    double sigma = 1.6, k = pow(2., 1. / NOL);
    g->ksize[0] = 0;
    for (int i = 1; i < NG; ++i) {
        double sp = pow(k, (double)(i - 1)) * sigma, st = sp * k;
        g->ksize[i] = gauss_kernel(sqrt(st * st - sp * sp), g->kern[i]);
    }
    float sd = sqrtf(fmaxf((float)(sigma * sigma) - 0.5f * 0.5f * 4, 0.01f));
    g->ksize0 = gauss_kernel((double)sd, g->sigma0_k);
}

static int refl(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; if (p >= n) p = 2 * n - 2 - p; } return p; }

/* GaussianBlur on an f32 image = sepFilter2D with the f32 Gaussian kernel (imgproc filter.simd.hpp), in the operation order
 * of the AVX2-dispatched build every x86 wheel runs (the reference's own ORB rows single out the fused form of this very
 * code for the descriptor blur, orb_oracle.c):
 *   row pass     RowVec_32f:        s = k[0] v[x-r];  s = fma(k[i], v[x-r+i], s), i = 1 .. ks-1
 *   column pass  SymmColumnVec_32f: s = k[r] c;       s = fma(k[r+j], v[y+j] + v[y-j], s), j = 1 .. r
 * (rounds 1-2 summed all taps of both passes unfused and in ascending order.)  SIFT itself stays parity unpinned against
 * cv2 -- the reference holds no SIFT answers. */
static void blur_f32(const float *src, float *dst, float *tmp, int w, int h, const float *k, int ks)
{
    int r = ks / 2;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        float s = k[0] * src[(size_t)y * w + refl(x - r, w)];
        for (int i = 1; i < ks; ++i) s = fmaf(k[i], src[(size_t)y * w + refl(x + i - r, w)], s);
        tmp[(size_t)y * w + x] = s;
    }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        float s = k[r] * tmp[(size_t)y * w + x];
        for (int j = 1; j <= r; ++j) s = fmaf(k[r + j], tmp[(size_t)refl(y + j, h) * w + x] + tmp[(size_t)refl(y - j, h) * w + x], s);
        dst[(size_t)y * w + x] = s;
    }
}

/* gaussian pyramid: gp[goff[o] + i*w*h] */
static void build_gauss(const uint8_t *img, int W, int H, const sift_geo *g, float *gp)
{
    int bw = 2 * W, bh = 2 * H;
    float *up = (float *)malloc(sizeof(float) * (size_t)bw * bh * 2), *tmp = up + (size_t)bw * bh;
    /* resize x2 INTER_LINEAR: src = (d + 0.5)*0.5 - 0.5, clamped */
    for (int y = 0; y < bh; ++y) {
        float fy = (y + 0.5f) * 0.5f - 0.5f; int sy = (int)floorf(fy); fy -= sy;
        if (sy < 0) { sy = 0; fy = 0.f; }
        int sy1 = sy + 1 < H ? sy + 1 : H - 1;
        if (sy >= H - 1) { sy = H - 1; sy1 = H - 1; fy = 0.f; }
        for (int x = 0; x < bw; ++x) {
            float fx = (x + 0.5f) * 0.5f - 0.5f; int sx = (int)floorf(fx); fx -= sx;
            if (sx < 0) { sx = 0; fx = 0.f; }
            int sx1 = sx + 1 < W ? sx + 1 : W - 1;
            if (sx >= W - 1) { sx = W - 1; sx1 = W - 1; fx = 0.f; }
            float h0 = (float)img[(size_t)sy * W + sx] * (1.f - fx) + (float)img[(size_t)sy * W + sx1] * fx;
            float h1 = (float)img[(size_t)sy1 * W + sx] * (1.f - fx) + (float)img[(size_t)sy1 * W + sx1] * fx;
            up[(size_t)y * bw + x] = h0 * (1.f - fy) + h1 * fy;
        }
    }
    blur_f32(up, gp + g->goff[0], tmp, bw, bh, g->sigma0_k, g->ksize0);
    free(up);
    for (int o = 0; o < g->noct; ++o) {
        int w = g->w[o], h = g->h[o]; size_t n = (size_t)w * h;
        float *t2 = (float *)malloc(sizeof(float) * n);
        if (o > 0) {   /* INTER_NEAREST halving of gaussian[NOL] of the previous octave */
            const float *src = gp + g->goff[o - 1] + (size_t)NOL * g->w[o - 1] * g->h[o - 1];
            float *dst = gp + g->goff[o];
            for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) dst[(size_t)y * w + x] = src[(size_t)(2 * y) * g->w[o - 1] + 2 * x];
        }
        for (int i = 1; i < NG; ++i) blur_f32(gp + g->goff[o] + (i - 1) * n, gp + g->goff[o] + i * n, t2, w, h, g->kern[i], g->ksize[i]);
        free(t2);
    }
}

typedef struct { float x, y, size, angle, response; int octave; } sift_kp;   /* octave = cv packed form */

/* adjustLocalExtrema; dog(o,l) = G(l+1) - G(l) evaluated on the fly */
typedef struct { const float *gp; const sift_geo *g; int o; } dog_ctx;
static float DOG(const dog_ctx *c, int l, int r, int x)
{
    size_t n = (size_t)c->g->w[c->o] * c->g->h[c->o];
    const float *b = c->gp + c->g->goff[c->o];
    size_t i = (size_t)r * c->g->w[c->o] + x;
    return b[(l + 1) * n + i] - b[l * n + i];
}

static int adjust_extremum(const dog_ctx *c, int *layer, int *r, int *x, float *xi_, float *xr_, float *xc_, float *contr_)
{
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
    const int w = c->g->w[c->o], h = c->g->h[c->o];
    float xi = 0, xr = 0, xc = 0;
    int i = 0, l = *layer, rr = *r, cc = *x;
    for (; i < 5; ++i) {
        float dD0 = (DOG(c, l, rr, cc + 1) - DOG(c, l, rr, cc - 1)) * deriv_scale;
        float dD1 = (DOG(c, l, rr + 1, cc) - DOG(c, l, rr - 1, cc)) * deriv_scale;
        float dD2 = (DOG(c, l + 1, rr, cc) - DOG(c, l - 1, rr, cc)) * deriv_scale;
        float v2 = DOG(c, l, rr, cc) * 2;
        float dxx = (DOG(c, l, rr, cc + 1) + DOG(c, l, rr, cc - 1) - v2) * second;
        float dyy = (DOG(c, l, rr + 1, cc) + DOG(c, l, rr - 1, cc) - v2) * second;
        float dss = (DOG(c, l + 1, rr, cc) + DOG(c, l - 1, rr, cc) - v2) * second;
        float dxy = (DOG(c, l, rr + 1, cc + 1) - DOG(c, l, rr + 1, cc - 1) - DOG(c, l, rr - 1, cc + 1) + DOG(c, l, rr - 1, cc - 1)) * cross;
        float dxs = (DOG(c, l + 1, rr, cc + 1) - DOG(c, l + 1, rr, cc - 1) - DOG(c, l - 1, rr, cc + 1) + DOG(c, l - 1, rr, cc - 1)) * cross;
        float dys = (DOG(c, l + 1, rr + 1, cc) - DOG(c, l + 1, rr - 1, cc) - DOG(c, l - 1, rr + 1, cc) + DOG(c, l - 1, rr - 1, cc)) * cross;
        /* X = H^-1 dD by Gaussian elimination with partial pivoting (Matx solve DECOMP_LU), f32 */
        float A[3][4] = {{dxx, dxy, dxs, dD0}, {dxy, dyy, dys, dD1}, {dxs, dys, dss, dD2}};
        int ok = 1;
        for (int p = 0; p < 3; ++p) {
            int piv = p;
            for (int q = p + 1; q < 3; ++q) if (fabsf(A[q][p]) > fabsf(A[piv][p])) piv = q;
            if (fabsf(A[piv][p]) < FLT_EPSILON) { ok = 0; break; }
            if (piv != p) for (int q = 0; q < 4; ++q) { float t = A[p][q]; A[p][q] = A[piv][q]; A[piv][q] = t; }
            float d = -1.f / A[p][p];
            for (int q = p + 1; q < 3; ++q) {
                float al = A[q][p] * d;
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
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return 0;
        cc += cv_round_f(xc); rr += cv_round_f(xr); l += cv_round_f(xi);
        if (l < 1 || l > NOL || cc < SIFT_BORDER || cc >= w - SIFT_BORDER || rr < SIFT_BORDER || rr >= h - SIFT_BORDER) return 0;
    }
    if (i >= 5) return 0;
    {
        float dD0 = (DOG(c, l, rr, cc + 1) - DOG(c, l, rr, cc - 1)) * deriv_scale;
        float dD1 = (DOG(c, l, rr + 1, cc) - DOG(c, l, rr - 1, cc)) * deriv_scale;
        float dD2 = (DOG(c, l + 1, rr, cc) - DOG(c, l - 1, rr, cc)) * deriv_scale;
        float t = (dD0 * xc + dD1 * xr) + dD2 * xi;
        float contr = DOG(c, l, rr, cc) * img_scale + t * 0.5f;
        if (fabsf(contr) * NOL < 0.04f) return 0;
        float v2 = DOG(c, l, rr, cc) * 2.f;
        float dxx = (DOG(c, l, rr, cc + 1) + DOG(c, l, rr, cc - 1) - v2) * second;
        float dyy = (DOG(c, l, rr + 1, cc) + DOG(c, l, rr - 1, cc) - v2) * second;
        float dxy = (DOG(c, l, rr + 1, cc + 1) - DOG(c, l, rr + 1, cc - 1) - DOG(c, l, rr - 1, cc + 1) + DOG(c, l, rr - 1, cc - 1)) * cross;
        float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * 10.f >= (10.f + 1) * (10.f + 1) * det) return 0;
        *contr_ = contr;
    }
    *layer = l; *r = rr; *x = cc; *xi_ = xi; *xr_ = xr; *xc_ = xc;
    return 1;
}

/* calcOrientationHist with 64-lane interleaved partial sums; returns max of smoothed hist */
static float orientation_hist(const float *img, int w, int h, int px, int py, int radius, float sigma, float *hist)
{
    const int n = ORI_BINS;
    float expf_scale = -1.f / (2.f * sigma * sigma);
    float part[ORI_BINS][8];
    memset(part, 0, sizeof(part));
    int k = 0;
    for (int i = -radius; i <= radius; ++i) {
        int y = py + i;
        for (int j = -radius; j <= radius; ++j, ++k) {       /* k = raster index inside the (2r+1)^2 window */
            int x = px + j;
            if (y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1) continue;
            float dx = img[(size_t)y * w + x + 1] - img[(size_t)y * w + x - 1];
            float dy = img[(size_t)(y - 1) * w + x] - img[(size_t)(y + 1) * w + x];
            float wgt = det_expf((float)(i * i + j * j) * expf_scale);
            float ori = orc_fast_atan2(dy, dx);
            float mag = sqrtf(dx * dx + dy * dy);
            int bin = cv_round_f((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            part[bin][k & 7] += wgt * mag;
        }
    }
    float temphist[ORI_BINS + 4], *th = temphist + 2;
    for (int b = 0; b < n; ++b) th[b] = tree8(part[b]);
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    float mx = 0;
    for (int b = 0; b < n; ++b) {
        hist[b] = (th[b - 2] + th[b + 2]) * (1.f / 16.f) + (th[b - 1] + th[b + 1]) * (4.f / 16.f) + th[b] * (6.f / 16.f);
        if (b == 0 || hist[b] > mx) mx = hist[b];
    }
    return mx;
}

/* calcSIFTDescriptor with 64-lane interleaved partial sums */
static void sift_descriptor(const float *img, int w, int h, float ptx, float pty, float ori, float scl, float *dst)
{
    const int d = 4, n = 8;
    int px = cv_round_f(ptx), py = cv_round_f(pty);
    double sn, cs;
    det_sincos((double)(ori * (float)(3.141592653589793238462643383279502884 / 180.0)), &sn, &cs);
    float cos_t = (float)cs, sin_t = (float)sn;
    float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
    int radius = cv_round_f(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    int rmax = (int)sqrt(((double)w) * w + ((double)h) * h);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    const int HL = (d + 2) * (d + 2) * (n + 2);
    float part[360][8];
    memset(part, 0, sizeof(part));
    int k = 0;                                               /* counts the samples that pass the window test, as cv2's first loop does */
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
            float dx = img[(size_t)r * w + c + 1] - img[(size_t)r * w + c - 1];
            float dy = img[(size_t)(r - 1) * w + c] - img[(size_t)(r + 1) * w + c];
            float wgt = det_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float o = orc_fast_atan2(dy, dx);
            float mag = sqrtf(dx * dx + dy * dy) * wgt;
            float obin = (o - ori) * bins_per_rad;
            int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin), o0 = (int)floorf(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            float v111 = v_rc11 * obin, v110 = v_rc11 - v111, v101 = v_rc10 * obin, v100 = v_rc10 - v101;
            float v011 = v_rc01 * obin, v010 = v_rc01 - v011, v001 = v_rc00 * obin, v000 = v_rc00 - v001;
            int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0, L = k & 7;
            part[idx][L] += v000; part[idx + 1][L] += v001;
            part[idx + (n + 2)][L] += v010; part[idx + (n + 3)][L] += v011;
            part[idx + (d + 2) * (n + 2)][L] += v100; part[idx + (d + 2) * (n + 2) + 1][L] += v101;
            part[idx + (d + 3) * (n + 2)][L] += v110; part[idx + (d + 3) * (n + 2) + 1][L] += v111;
            ++k;
        }
    float hist[360];
    for (int b = 0; b < HL; ++b) hist[b] = tree8(part[b]);
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
        int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
        hist[idx] += hist[idx + n]; hist[idx + 1] += hist[idx + n + 1];
        for (int b = 0; b < n; ++b) dst[(i * d + j) * n + b] = hist[idx + b];
    }
    /* normalise: 128 values summed in a fixed tree of two 64-lane halves */
    float sq[64];
    for (int l = 0; l < 64; ++l) sq[l] = dst[l] * dst[l] + dst[l + 64] * dst[l + 64];
    float thr = sqrtf(tree64(sq)) * 0.2f;
    for (int l = 0; l < 128; ++l) dst[l] = dst[l] < thr ? dst[l] : thr;
    for (int l = 0; l < 64; ++l) sq[l] = dst[l] * dst[l] + dst[l + 64] * dst[l + 64];
    float nrm = sqrtf(tree64(sq));
    float f = 512.f / (nrm > FLT_EPSILON ? nrm : FLT_EPSILON);
    for (int l = 0; l < 128; ++l) {
        int v = cv_round_f(dst[l] * f);          /* saturate_cast<uchar>(float) rounds to nearest even */
        dst[l] = (float)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

static int kp_less(const sift_kp *a, const sift_kp *b)   /* KeyPoint_LessThan (keypoint.cpp) */
{
    if (a->x != b->x) return a->x < b->x;
    if (a->y != b->y) return a->y < b->y;
    if (a->size != b->size) return a->size > b->size;
    if (a->angle != b->angle) return a->angle < b->angle;
    if (a->response != b->response) return a->response > b->response;
    return a->octave > b->octave;
}
static int kp_cmp(const void *a, const void *b) { return kp_less((const sift_kp *)a, (const sift_kp *)b) ? -1 : kp_less((const sift_kp *)b, (const sift_kp *)a) ? 1 : 0; }

/* nfeatures <= 0: no cap.  seed_cap: max seeds per image (raster order).  Returns count (<= cap). */
int orc_sift_detect_and_compute(const uint8_t *img, int W, int H, int nfeatures, int seed_cap,
                                orc_sift_keypoint *kps, float *desc, int cap)
{
    return orc_sift_detect_and_compute_ex(img, W, H, nfeatures, seed_cap, kps, desc, cap, NULL);
}

int orc_sift_detect_and_compute_ex(const uint8_t *img, int W, int H, int nfeatures, int seed_cap,
                                   orc_sift_keypoint *kps, float *desc, int cap, uint32_t *flags)
{
    uint32_t ovf = 0;
    sift_geo g;
    sift_geometry(W, H, &g);
    float *gp = (float *)malloc(sizeof(float) * g.gtotal);
    build_gauss(img, W, H, &g, gp);
    int nk = 0, kcap = 4 * seed_cap + 16;
    sift_kp *tmp = (sift_kp *)malloc(sizeof(sift_kp) * (size_t)kcap);
    int nseeds = 0;
    for (int o = 0; o < g.noct; ++o) {
        const int w = g.w[o], h = g.h[o];
        if (w <= 2 * SIFT_BORDER || h <= 2 * SIFT_BORDER) continue;
        dog_ctx dc = {gp, &g, o};
        size_t n = (size_t)w * h;
        int *claimed = (int *)calloc(n * (NOL + 2), sizeof(int));
        for (int i = 1; i <= NOL; ++i)
            for (int r = SIFT_BORDER; r < h - SIFT_BORDER; ++r)
                for (int c = SIFT_BORDER; c < w - SIFT_BORDER; ++c) {
                    float val = DOG(&dc, i, r, c);
                    if (!(fabsf(val) > 1.f)) continue;
                    int ismax = val > 0, ismin = val < 0;
                    for (int dl = -1; dl <= 1 && (ismax || ismin); ++dl)
                        for (int dr = -1; dr <= 1; ++dr)
                            for (int dcx = -1; dcx <= 1; ++dcx) {
                                float v = DOG(&dc, i + dl, r + dr, c + dcx);
                                if (v > val) ismax = 0;
                                if (v < val) ismin = 0;
                            }
                    if (!(ismax || ismin)) continue;
                    if (nseeds >= seed_cap) { ovf |= ORC_OVF_SIFT_SEEDS; continue; }
                    ++nseeds;
                    int l = i, rr = r, cc = c; float xi, xr, xc, contr;
                    if (!adjust_extremum(&dc, &l, &rr, &cc, &xi, &xr, &xc, &contr)) continue;
                    size_t ci = (size_t)l * n + (size_t)rr * w + cc;
                    if (claimed[ci]) continue;             /* exact duplicate of an earlier seed */
                    claimed[ci] = 1;
                    sift_kp kp;
                    kp.x = (cc + xc) * (1 << o); kp.y = (rr + xr) * (1 << o);
                    kp.octave = o + (l << 8) + (cv_round_d((xi + 0.5) * 255) << 16);
                    kp.size = 1.6f * det_exp2f((l + xi) / NOL) * (1 << o) * 2;
                    kp.response = fabsf(contr);
                    float scl_octv = kp.size * 0.5f / (1 << o);
                    float hist[ORI_BINS];
                    float omax = orientation_hist(gp + g.goff[o] + (size_t)l * n, w, h, cc, rr, cv_round_f(4.5f * scl_octv), 1.5f * scl_octv, hist);
                    float mag_thr = omax * 0.8f;
                    for (int j = 0; j < ORI_BINS; ++j) {
                        int lft = j > 0 ? j - 1 : ORI_BINS - 1, rgt = j < ORI_BINS - 1 ? j + 1 : 0;
                        if (hist[j] > hist[lft] && hist[j] > hist[rgt] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[lft] - hist[rgt]) / (hist[lft] - 2 * hist[j] + hist[rgt]);
                            bin = bin < 0 ? ORI_BINS + bin : bin >= ORI_BINS ? bin - ORI_BINS : bin;
                            kp.angle = 360.f - (360.f / ORI_BINS) * bin;
                            if (fabsf(kp.angle - 360.f) < FLT_EPSILON) kp.angle = 0.f;
                            if (nk < kcap) tmp[nk++] = kp; else ovf |= ORC_OVF_SIFT_RAW;
                        }
                    }
                }
        free(claimed);
    }
    /* retainBest(nfeatures): all keypoints with response >= the n-th best */
    if (nfeatures > 0 && nk > nfeatures) {
        ovf |= ORC_OVF_SIFT_CAP;          /* the cap bites: the reference's SIFT_create() keeps all nk */
        float *rs = (float *)malloc(sizeof(float) * (size_t)nk);
        for (int i = 0; i < nk; ++i) rs[i] = tmp[i].response;
        for (int i = 0; i < nfeatures; ++i) { int m = i; for (int j = i + 1; j < nk; ++j) if (rs[j] > rs[m]) m = j; float t = rs[i]; rs[i] = rs[m]; rs[m] = t; }
        float th = rs[nfeatures - 1];
        int m = 0;
        for (int i = 0; i < nk; ++i) if (tmp[i].response >= th) tmp[m++] = tmp[i];
        nk = m; free(rs);
    }
    qsort(tmp, (size_t)nk, sizeof(sift_kp), kp_cmp);
    if (nk > cap) { nk = cap; ovf |= ORC_OVF_SIFT_KEYPOINTS; }
    if (flags) *flags = ovf;
    for (int i = 0; i < nk; ++i) {
        sift_kp kp = tmp[i];
        /* descriptor is computed at the un-halved coordinates (octave index o, scale 1/(1<<o)) */
        int o = kp.octave & 255, l = (kp.octave >> 8) & 255;
        float scale = 1.f / (float)(1 << o);
        float size = kp.size * scale;
        size_t n = (size_t)g.w[o] * g.h[o];
        float angle = 360.f - kp.angle;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        sift_descriptor(gp + g.goff[o] + (size_t)l * n, g.w[o], g.h[o], kp.x * scale, kp.y * scale, angle, size * 0.5f, desc + (size_t)i * 128);
        /* firstOctave = -1: report coordinates of the original image */
        kps[i].x = kp.x * 0.5f; kps[i].y = kp.y * 0.5f; kps[i].size = kp.size * 0.5f;
        kps[i].angle = kp.angle; kps[i].response = kp.response;
        kps[i].octave = (kp.octave & ~255) | ((kp.octave - 1) & 255);
    }
    free(tmp); free(gp);
    return nk;
}

/* gaussian pyramid dump for stage tests: returns float count; out may be NULL to query */
int64_t orc_sift_gauss_pyramid(const uint8_t *img, int W, int H, float *out, int32_t *dims /* noct, then w,h per octave */)
{
    sift_geo g;
    sift_geometry(W, H, &g);
    if (dims) { dims[0] = g.noct; for (int o = 0; o < g.noct; ++o) { dims[1 + 2 * o] = g.w[o]; dims[2 + 2 * o] = g.h[o]; } }
    if (out) build_gauss(img, W, H, &g, out);
    return (int64_t)g.gtotal;
}

float orc_det_expf(float x) { return det_expf(x); }


/* ---------------------------------------------------------- end to end (SIFT + L2) */
int orc_estimate_pose_sift(const uint8_t *img1, const uint8_t *img2, int W, int H, const double *K,
                           int nfeatures, int max_matches, orc_pose_result *out)
{
    memset(out, 0, sizeof(*out));
    int cap = nfeatures > 0 ? nfeatures + 64 : 65536;
    orc_sift_keypoint *k1 = (orc_sift_keypoint *)malloc(sizeof(orc_sift_keypoint) * 2 * (size_t)cap), *k2 = k1 + cap;
    float *d1 = (float *)malloc(sizeof(float) * 256 * (size_t)cap), *d2 = d1 + 128 * (size_t)cap;
    int seed_cap = (int)(((long long)4 * W * H) / 16); if (seed_cap < 16384) seed_cap = 16384;   /* rule of the HIP path */
    uint32_t f1 = 0, f2 = 0;
    int n1 = orc_sift_detect_and_compute_ex(img1, W, H, nfeatures, seed_cap, k1, d1, cap, &f1);
    int n2 = orc_sift_detect_and_compute_ex(img2, W, H, nfeatures, seed_cap, k2, d2, cap, &f2);
    out->n_kp1 = n1; out->n_kp2 = n2; out->overflow = (int32_t)(f1 | f2);
    int mm = max_matches >= 0 ? max_matches : cap;
    int32_t *qi = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)cap), *ti = qi + cap;
    float *di = (float *)malloc(sizeof(float) * (size_t)cap);
    float *p1 = (float *)malloc(sizeof(float) * 4 * (size_t)cap), *p2 = p1 + 2 * (size_t)cap;
    if (n1 == 0 || n2 == 0) { out->status = ORC_NO_DESCRIPTORS; goto done; }
    int M = orc_match_l2(d1, n1, d2, n2, 128, mm, qi, ti, di);
    out->n_matches = M;
    if (M < 5) { out->status = ORC_INSUFFICIENT_MATCHES; goto done; }
    for (int i = 0; i < M; ++i) {
        p1[2 * i] = k1[qi[i]].x; p1[2 * i + 1] = k1[qi[i]].y;
        p2[2 * i] = k2[ti[i]].x; p2[2 * i + 1] = k2[ti[i]].y;
    }
    double E[9];
    {
        const int ne = orc_find_essential(p1, p2, M, K, 0.999, 1.0, 1000, E, NULL, NULL);
        if (!ne) { out->status = ORC_NO_ESSENTIAL; goto done; }
        if (ne > 1) { out->status = ORC_AMBIGUOUS_ESSENTIAL; goto done; }
    }
    out->inliers = orc_recover_pose(E, p1, p2, M, K, out->R, out->t);
    out->status = ORC_OK;
done:
    free(p1); free(di); free(qi); free(d1); free(k1);
    return out->status;
}

#include <pthread.h>
typedef struct { const uint8_t *i1, *i2; int B, W, H; const double *K; int nf, mm; orc_pose_result *out; int tid, nt; } sjob_t;
static void *sift_worker(void *arg)
{
    sjob_t *j = (sjob_t *)arg;
    size_t sz = (size_t)j->W * j->H;
    for (int b = j->tid; b < j->B; b += j->nt)
        orc_estimate_pose_sift(j->i1 + sz * b, j->i2 + sz * b, j->W, j->H, j->K, j->nf, j->mm, &j->out[b]);
    return NULL;
}
void orc_estimate_pose_sift_batch(const uint8_t *imgs1, const uint8_t *imgs2, int B, int W, int H, const double *K,
                                  int nfeatures, int max_matches, orc_pose_result *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; sjob_t jb[256];
    for (int t = 0; t < nthreads; ++t) {
        sjob_t j = {imgs1, imgs2, B, W, H, K, nfeatures, max_matches, out, t, nthreads};
        jb[t] = j;
        pthread_create(&th[t], NULL, sift_worker, &jb[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}
