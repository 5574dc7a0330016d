/*
 * orb_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of
 *   cv2.ORB_create(nfeatures, scaleFactor=1.1, nlevels=12, fastThreshold=15,
 *                  scoreType=HARRIS_SCORE).detectAndCompute(image, None)
 *   (reference src/core/pose_estimator.py:85-91, :108).
 *
 * Follows OpenCV 4.x features2d/orb.cpp (ORB_Impl::detectAndCompute,
 * computeKeyPoints, HarrisResponses, ICAngles, computeOrbDescriptors),
 * fast.cpp / fast_score.cpp (FAST-9/16 + cornerScore + 3x3 NMS), keypoint.cpp
 * (runByImageBorder, retainBest), imgproc resize.cpp (INTER_LINEAR_EXACT,
 * ufixedpoint16) and smooth (fixed-point Gaussian), mathfuncs_core (fastAtan2).
 *
 * Conventions: the defaults of the knobs below are the ones the reference's own 147 result rows pin
 * (tests/test_reference_rows_cpu.py):
 *  - keypoint ORDER inside a level: what cv2's two retainBest calls leave behind, i.e. the C++ runtime's
 *    std::nth_element + std::partition (retain_best.cpp: the real libstdc++ ones; MSVC's and libc++'s restated);
 *  - descriptor blur: sepFilter2D's f32 route with fused multiply-adds (GaussianBlur on a pyramid sub-matrix);
 *  - the 256-pair rBRIEF pattern (bit_pattern_31_), written from memory and confirmed by the rows;
 *  - cos/sin of the keypoint angle from a deterministic f64 kernel (fdlibm polynomials), so that the HIP kernel
 *    reproduces the same f32 values (own convention, invisible in the rows);
 *  - fixed workspace caps (raster corner list per level clamp(w h / 64, 1024, 8192); candidates 4*quota+256;
 *    keypoints per image `cap`), truncating in cv2's own order and flagged (cv2 has no caps).
 */
#include "oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define NLEVELS ORC_ORB_LEVELS
#define EDGE 31
#define HALF_PATCH 15

static int cv_round(double v) { return (int)lrint(v); }

/* Convention knobs.  The DEFAULTS are what the reference's 147 committed answers pin (tools/forensic*.py,
 * tools/agree.py, tests/test_reference_rows_cpu.py) and what the HIP path implements; the other values are the
 * hypotheses that evidence ruled out, kept for the record and for the experiment tools.
 * Key 0, descriptor blur: 3 = sepFilter2D's f32 route with fused multiply-adds (the AVX2-dispatched build of
 *   filter.simd.hpp contracts s += f * x), DEFAULT; 2 = the same unfused; 1 = fixed-point taps [18,34,48,56,...] (sum 256);
 *   0 = fixed-point taps cvRound(256 g) = [18,34,49,55,...] (sum 257, rounds 1-2).
 * Key 1, keypoint order inside a level: 3 = cv2's own on libstdc++ (Linux wheels; the reference's Dockerfile), DEFAULT:
 *   both retainBest calls through the real std::nth_element + std::partition; 5 = the same on the MSVC STL (Windows
 *   wheels: the simulator result file was produced there); 4 = on libc++ (macOS wheels); 0 = raster, 1 = Harris response
 *   descending, 2 = reverse raster (rounds 1-2).
 * Key 2, crossCheck rule: 0 = strict mutual nearest neighbours (OpenCV >= 4.5.x), DEFAULT; 1 = electors only. */
static int g_variant[4] = {3, 3, 0, 0};
int orc_orb_corner_cap(int w, int h) { int c = (int)(((long long)w * h) / 64); return c < 1024 ? 1024 : c > 8192 ? 8192 : c; }
int orc_retain_best(const float *resp, int32_t *ids, int n, int n_points);   /* retain_best.cpp */
int orc_retain_best_llvm(const float *resp, int32_t *ids, int n, int n_points);
int orc_retain_best_msvc(const float *resp, int32_t *ids, int n, int n_points);
void orc_debug_set_variant(int key, int val) { if (key >= 0 && key < 4) g_variant[key] = val; }
int orc_debug_get_variant(int key) { return key >= 0 && key < 4 ? g_variant[key] : 0; }

/* orb.cpp: layer scale (float)pow(scaleFactor, level), size cvRound(cols/scale);
 * per-level feature quota (geometric series, remainder to the last level). */
void orc_orb_layout_init(int W, int H, int nfeatures, orc_orb_layout *L)
{
    double sf = (double)1.1f;
    int64_t off = 0;
    for (int l = 0; l < NLEVELS; ++l) {
        float s = (float)pow(sf, (double)l);
        L->scale[l] = s;
        const float inv_scale = 1.0f / s;                        /* orb.cpp: Size sz(cvRound(image.cols * inv_scale), ...) */
        L->w[l] = cv_round((double)((float)W * inv_scale));
        L->h[l] = cv_round((double)((float)H * inv_scale));
        L->offset[l] = off;
        off += (int64_t)L->w[l] * L->h[l];
    }
    L->total = off;
    float factor = (float)(1.0 / sf);
    float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)NLEVELS));
    int sum = 0;
    for (int l = 0; l < NLEVELS - 1; ++l) {
        L->quota[l] = cv_round((double)nd);
        sum += L->quota[l];
        nd *= factor;
    }
    L->quota[NLEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* INTER_LINEAR_EXACT coefficient table (resize.cpp interpolationLinear<ufixedpoint16>) */
static void lin_coeffs(int src, int dst, int *ofs, int *a1)
{
    double inv_scale = (double)dst / (double)src;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i = (int)floor(f);
        if (i >= 0 && src > 1) {
            if (i < src - 1) { ofs[d] = i; a1[d] = cv_round((f - (double)i) * 256.0); }
            else { ofs[d] = src - 1; a1[d] = 0; }
        } else { ofs[d] = 0; a1[d] = 0; }
    }
}

static void resize_exact(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    int *xo = (int *)malloc(sizeof(int) * 2 * (size_t)(dw + dh));
    int *xa = xo + dw, *yo = xa + dw, *ya = yo + dh;
    lin_coeffs(sw, dw, xo, xa);
    lin_coeffs(sh, dh, yo, ya);
    for (int y = 0; y < dh; ++y) {
        const uint8_t *r0 = src + (size_t)yo[y] * sw;
        const uint8_t *r1 = src + (size_t)(yo[y] + 1 < sh ? yo[y] + 1 : sh - 1) * sw;
        int b1 = ya[y], b0 = 256 - b1;
        for (int x = 0; x < dw; ++x) {
            int o = xo[x], o1 = o + 1 < sw ? o + 1 : sw - 1;
            int a1 = xa[x], a0 = 256 - a1;
            uint32_t h0 = (uint32_t)(a0 * r0[o] + a1 * r0[o1]);
            uint32_t h1 = (uint32_t)(a0 * r1[o] + a1 * r1[o1]);
            dst[(size_t)y * dw + x] = (uint8_t)(((uint32_t)b0 * h0 + (uint32_t)b1 * h1 + 32768u) >> 16);
        }
    }
    free(xo);
}

void orc_orb_build_pyramid(const uint8_t *img, int W, int H, const orc_orb_layout *L, uint8_t *pyr)
{
    memcpy(pyr, img, (size_t)W * H);
    for (int l = 1; l < NLEVELS; ++l)
        resize_exact(pyr + L->offset[l - 1], L->w[l - 1], L->h[l - 1], pyr + L->offset[l], L->w[l], L->h[l]);
}

/* FAST-9/16 (fast.cpp) with cornerScore<16> (fast_score.cpp).  score map value:
 * 0 for non-corners, else the largest threshold for which the pixel is still a
 * corner = max(maxarc min(v-p), maxarc min(p-v)) - 1  (>= thr). */
static const int8_t CIRC[16][2] = {{0,3},{1,3},{2,2},{3,1},{3,0},{3,-1},{2,-2},{1,-3},{0,-3},{-1,-3},{-2,-2},{-3,-1},{-3,0},{-3,1},{-2,2},{-1,3}};

void orc_orb_fast_score_map(const uint8_t *lvl, int w, int h, int thr, uint8_t *score)
{
    memset(score, 0, (size_t)w * h);
    int ofs[16];
    for (int k = 0; k < 16; ++k) ofs[k] = CIRC[k][1] * w + CIRC[k][0];
    for (int y = 3; y < h - 3; ++y) {
        for (int x = 3; x < w - 3; ++x) {
            const uint8_t *p = lvl + (size_t)y * w + x;
            int v = p[0];
            int d0 = v - p[ofs[0]], d8 = v - p[ofs[8]];
            if (abs(d0) <= thr && abs(d8) <= thr) continue;
            int d4 = v - p[ofs[4]], d12 = v - p[ofs[12]];
            if (abs(d4) <= thr && abs(d12) <= thr) continue;
            int d[25];
            for (int k = 0; k < 16; ++k) d[k] = v - p[ofs[k]];
            for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
            int A = -1000, B = -1000;
            for (int k = 0; k < 16; ++k) {
                int mn = d[k], mx = d[k];
                for (int j = 1; j < 9; ++j) { if (d[k + j] < mn) mn = d[k + j]; if (d[k + j] > mx) mx = d[k + j]; }
                if (mn > A) A = mn;
                if (-mx > B) B = -mx;
            }
            int s = A > B ? A : B;
            if (s > thr) score[(size_t)y * w + x] = (uint8_t)(s - 1);
        }
    }
}

/* 3x3 non-maximum suppression (fast.cpp): keep iff score > all 8 neighbours;
 * plus KeyPointsFilter::runByImageBorder(edgeThreshold=31). */
void orc_orb_nms_map(const uint8_t *score, int w, int h, uint8_t *nms)
{
    memset(nms, 0, (size_t)w * h);
    for (int y = EDGE; y < h - EDGE; ++y)
        for (int x = EDGE; x < w - EDGE; ++x) {
            const uint8_t *s = score + (size_t)y * w + x;
            int v = s[0];
            if (!v) continue;
            if (v > s[-1] && v > s[1] && v > s[-w - 1] && v > s[-w] && v > s[-w + 1] &&
                v > s[w - 1] && v > s[w] && v > s[w + 1])
                nms[(size_t)y * w + x] = (uint8_t)v;
        }
}

/* fixed-point separable Gaussian 7x7 sigma 2, BORDER_REFLECT_101 */
static const int GK_TAB[2][7] = {{18, 34, 49, 55, 49, 34, 18}, {18, 34, 48, 56, 48, 34, 18}};
static int refl(int p, int n) { if (p < 0) p = -p; if (p >= n) p = 2 * n - 2 - p; return p; }

/* experiment (knob 0 = 2): sepFilter2D's float route -- f32 kernel = (float)(exp(-x^2 / 8) / sum), row pass accumulates
 * k = 0..6 in f32, column pass centre first then the symmetric pairs, cvRound + saturate */
static void blur_level_float(const uint8_t *src, int w, int h, uint8_t *dst)
{
    float g[7];
    {
        double v[7], sum = 0.;
        for (int i = 0; i < 7; ++i) { double x = (double)(i - 3); v[i] = exp(-0.125 * x * x); sum += v[i]; }
        for (int i = 0; i < 7; ++i) g[i] = (float)(v[i] / sum);
    }
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = g[0] * (float)src[(size_t)y * w + refl(x - 3, w)];
            if (g_variant[0] == 3) for (int k = 1; k < 7; ++k) s = fmaf(g[k], (float)src[(size_t)y * w + refl(x + k - 3, w)], s);
            else for (int k = 1; k < 7; ++k) s += g[k] * (float)src[(size_t)y * w + refl(x + k - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = g[3] * tmp[(size_t)y * w + x];
            if (g_variant[0] == 3) for (int k = 1; k <= 3; ++k) s = fmaf(g[3 + k], tmp[(size_t)refl(y + k, h) * w + x] + tmp[(size_t)refl(y - k, h) * w + x], s);
            else for (int k = 1; k <= 3; ++k) s += g[3 + k] * (tmp[(size_t)refl(y + k, h) * w + x] + tmp[(size_t)refl(y - k, h) * w + x]);
            long r = lrintf(s);
            dst[(size_t)y * w + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    free(tmp);
}

void orc_orb_blur_level(const uint8_t *src, int w, int h, uint8_t *dst)
{
    if (g_variant[0] >= 2) { blur_level_float(src, w, h, dst); return; }
    const int *GK = GK_TAB[g_variant[0] & 1];
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = 0; k < 7; ++k) s += (uint32_t)GK[k] * src[(size_t)y * w + refl(x + k - 3, w)];
            tmp[(size_t)y * w + x] = (uint16_t)s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = 0; k < 7; ++k) s += (uint32_t)GK[k] * tmp[(size_t)refl(y + k - 3, h) * w + x];
            s = (s + 32768u) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(s > 255u ? 255u : s);     /* FixedPtCastEx saturates */
        }
    free(tmp);
}

/* mathfuncs_core atan_f32 */
float orc_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    const int fused = g_variant[3] & 2;                     /* experiment knob 3, bit 1: the polynomial with fused multiply-adds */
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = fused ? fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (fused ? fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c);
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* deterministic sin/cos for x in [0, 2*pi] (fdlibm kernel polynomials) */
static void det_sincos(double x, double *sn, double *cs)
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

/* rBRIEF sampling pattern (orb.cpp bit_pattern_31_), x0,y0,x1,y1 per bit */
static const int8_t PATTERN[256 * 4] = {
#include "orb_pattern.inc"
};
const int8_t *orc_orb_pattern(void) { return PATTERN; }

typedef struct { int x, y; int score; float resp; } cand_t;

static float harris_response(const uint8_t *lvl, int w, int x0, int y0)
{
    /* orb.cpp HarrisResponses, blockSize 7, k = 0.04 */
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *p = lvl + (size_t)(y0 + dy) * w + (x0 + dx);
            int Ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
            int Iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    float fa = (float)a, fb = (float)b, fc = (float)c;
    float t1 = fa * fb, t2 = fc * fc, t3 = t1 - t2;
    float s = fa + fb;
    float t4 = (0.04f * s) * s;
    return (t3 - t4) * scale_sq_sq;
}

static int umax_tab[HALF_PATCH + 2];
static int umax_ready = 0;
static void init_umax(void)
{
    if (umax_ready) return;
    int v, v0, vmax = (int)floor(HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH * sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax_tab[v] = cv_round(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax_tab[v0] == umax_tab[v0 + 1]) ++v0;
        umax_tab[v] = v0;
        ++v0;
    }
    umax_ready = 1;
}

static float ic_angle(const uint8_t *lvl, int w, int x0, int y0)
{
    /* orb.cpp ICAngles */
    const uint8_t *c = lvl + (size_t)y0 * w + x0;
    int m01 = 0, m10 = 0;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * c[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int vs = 0, d = umax_tab[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * w], vm = c[u - v * w];
            vs += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * vs;
    }
    return orc_fast_atan2((float)m01, (float)m10);
}

static float kth_largest_f(const float *v, int n, int k) /* k is 1-based */
{
    float *t = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(t, v, sizeof(float) * (size_t)n);
    /* simple selection by partial sort (n <= a few thousand) */
    for (int i = 0; i < k; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (t[j] > t[m]) m = j;
        float x = t[i]; t[i] = t[m]; t[m] = x;
    }
    float r = t[k - 1];
    free(t);
    return r;
}

int orc_orb_detect_and_compute(const uint8_t *img, int W, int H, int nfeatures,
                               int fast_threshold, orc_keypoint *kps, uint8_t *desc, int cap)
{
    return orc_orb_detect_and_compute_ex(img, W, H, nfeatures, fast_threshold, kps, desc, cap, NULL);
}

int orc_orb_detect_and_compute_ex(const uint8_t *img, int W, int H, int nfeatures,
                                  int fast_threshold, orc_keypoint *kps, uint8_t *desc, int cap, uint32_t *flags)
{
    uint32_t ovf = 0;
    init_umax();
    orc_orb_layout L;
    orc_orb_layout_init(W, H, nfeatures, &L);
    uint8_t *pyr = (uint8_t *)malloc((size_t)L.total * 3);
    uint8_t *smap = pyr + L.total, *nmap = smap + L.total;
    orc_orb_build_pyramid(img, W, H, &L, pyr);
    int nk = 0;
    for (int l = 0; l < NLEVELS; ++l) {
        int w = L.w[l], h = L.h[l], q = L.quota[l];
        if (w <= 2 * EDGE || h <= 2 * EDGE || q <= 0) continue;
        const uint8_t *lv = pyr + L.offset[l];
        uint8_t *sm = smap + L.offset[l], *nm = nmap + L.offset[l];
        orc_orb_fast_score_map(lv, w, h, fast_threshold, sm);
        orc_orb_nms_map(sm, w, h, nm);
        if (g_variant[1] >= 3) {
            /* cv2's own keypoint order (orb.cpp computeKeyPoints): FAST's raster emission -> retainBest(2q) on the FAST
             * score -> Harris -> retainBest(q), both through the C++ runtime's std::nth_element + std::partition
             * (retain_best.cpp).  Workspace capacities of the HIP path, mirrored here so that a truncated image
             * compares equal too (cv2 has none; both raise ORC_OVF_ORB_CANDIDATES): the raster corner list holds
             * orc_orb_corner_cap(w, h) entries (first in raster order), the list after the first retainBest 4q + 256. */
            int (*retain)(const float *, int32_t *, int, int) = g_variant[1] == 5 ? orc_retain_best_msvc : g_variant[1] == 4 ? orc_retain_best_llvm : orc_retain_best;
            const int ccap = orc_orb_corner_cap(w, h), kcap2 = 4 * q + 256;
            float *rs = (float *)malloc(sizeof(float) * (size_t)(ccap + 1));
            int32_t *id = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ccap + 1));
            int n0 = 0;
            for (int y = EDGE; y < h - EDGE; ++y) for (int x = EDGE; x < w - EDGE; ++x) {
                int v = nm[(size_t)y * w + x];
                if (!v) continue;
                if (n0 >= ccap) { ovf |= ORC_OVF_ORB_CANDIDATES; continue; }
                rs[n0] = (float)v; id[n0] = y * w + x; ++n0;
            }
            int n1 = retain(rs, id, n0, 2 * q);
            if (n1 > kcap2) { n1 = kcap2; ovf |= ORC_OVF_ORB_CANDIDATES; }
            for (int i = 0; i < n1; ++i) rs[i] = harris_response(lv, w, id[i] % w, id[i] / w);
            int n2k = retain(rs, id, n1, q);
            for (int i = 0; i < n2k; ++i) {
                if (nk >= cap) { ovf |= ORC_OVF_ORB_KEYPOINTS; break; }
                const int x = id[i] % w, y = id[i] / w;
                orc_keypoint *k = &kps[nk++];
                k->lx = x; k->ly = y; k->octave = l; k->response = harris_response(lv, w, x, y);
                k->angle = ic_angle(lv, w, x, y);
                k->x = (float)x * L.scale[l];
                k->y = (float)y * L.scale[l];
            }
            free(rs); free(id);
            continue;
        }
        /* retainBest(2*quota) on the FAST score: keep every keypoint whose score
         * >= the (2q)-th best score (keypoint.cpp) */
        int hist[256]; memset(hist, 0, sizeof(hist));
        int total = 0;
        for (int y = EDGE; y < h - EDGE; ++y) for (int x = EDGE; x < w - EDGE; ++x) {
            int v = nm[(size_t)y * w + x];
            if (v) { ++hist[v]; ++total; }
        }
        int tau = 1, n2 = 2 * q;
        if (total > n2) { int acc = 0; for (tau = 255; tau > 0; --tau) { acc += hist[tau]; if (acc >= n2) break; } }
        int ccap = 4 * q + 256, nc = 0;
        cand_t *cd = (cand_t *)malloc(sizeof(cand_t) * (size_t)ccap);
        for (int y = EDGE; y < h - EDGE; ++y) for (int x = EDGE; x < w - EDGE; ++x) {
            int v = nm[(size_t)y * w + x];
            if (!(v >= tau && v)) continue;
            if (nc >= ccap) { ovf |= ORC_OVF_ORB_CANDIDATES; continue; }     /* workspace cap: first ccap in raster order */
            cd[nc].x = x; cd[nc].y = y; cd[nc].score = v; cd[nc].resp = harris_response(lv, w, x, y); ++nc;
        }
        /* retainBest(quota) on the Harris response */
        float th = -INFINITY;
        if (nc > q) {
            float *r = (float *)malloc(sizeof(float) * (size_t)nc);
            for (int i = 0; i < nc; ++i) r[i] = cd[i].resp;
            th = kth_largest_f(r, nc, q);
            free(r);
        }
        /* experiment knob 1: emission order inside the level (default raster) */
        int *ord = (int *)malloc(sizeof(int) * (size_t)(nc + 1)), no = 0;
        for (int i = 0; i < nc; ++i) if (cd[i].resp >= th) ord[no++] = i;
        if (g_variant[1] == 1) {
            for (int a = 1; a < no; ++a) {           /* stable insertion sort, response descending */
                int v = ord[a], b = a - 1;
                while (b >= 0 && cd[ord[b]].resp < cd[v].resp) { ord[b + 1] = ord[b]; --b; }
                ord[b + 1] = v;
            }
        } else if (g_variant[1] == 2) {
            for (int a = 0; a < no / 2; ++a) { int t = ord[a]; ord[a] = ord[no - 1 - a]; ord[no - 1 - a] = t; }
        }
        for (int oi = 0; oi < no; ++oi) {
            const int i = ord[oi];
            if (nk >= cap) { ovf |= ORC_OVF_ORB_KEYPOINTS; break; }           /* workspace cap: level-major, raster */
            orc_keypoint *k = &kps[nk++];
            k->lx = cd[i].x; k->ly = cd[i].y; k->octave = l; k->response = cd[i].resp;
            k->angle = ic_angle(lv, w, cd[i].x, cd[i].y);
            k->x = (float)cd[i].x * L.scale[l];
            k->y = (float)cd[i].y * L.scale[l];
        }
        free(ord);
        free(cd);
    }
    /* descriptors on the blurred pyramid (orb.cpp computeOrbDescriptors, WTA_K = 2) */
    uint8_t *blur = smap; /* reuse */
    int done_level = -1;
    for (int i = 0; i < nk; ++i) {
        int l = kps[i].octave, w = L.w[l], h = L.h[l];
        if (l != done_level) { orc_orb_blur_level(pyr + L.offset[l], w, h, blur + L.offset[l]); done_level = l; }
        const uint8_t *bl = blur + L.offset[l];
        float sc = 1.f / L.scale[l];
        float ang = kps[i].angle * (float)(3.141592653589793238462643383279502884 / 180.0);
        double sn, cs;
        det_sincos((double)ang, &sn, &cs);
        float a = (float)cs, b = (float)sn;
        int cx = cv_round((double)(kps[i].x * sc)), cy = cv_round((double)(kps[i].y * sc));
        const uint8_t *center = bl + (size_t)cy * w + cx;
        uint8_t *d = desc + 32 * (size_t)i;
        for (int byte = 0; byte < 32; ++byte) {
            int val = 0;
            for (int bit = 0; bit < 8; ++bit) {
                const int8_t *p = PATTERN + 4 * (byte * 8 + bit);
                float x0 = (float)p[0] * a - (float)p[1] * b, y0 = (float)p[0] * b + (float)p[1] * a;
                float x1 = (float)p[2] * a - (float)p[3] * b, y1 = (float)p[2] * b + (float)p[3] * a;
                int ix0 = cv_round((double)x0), iy0 = cv_round((double)y0);
                int ix1 = cv_round((double)x1), iy1 = cv_round((double)y1);
                int t0 = center[iy0 * w + ix0], t1 = center[iy1 * w + ix1];
                val |= (t0 < t1) << bit;
            }
            d[byte] = (uint8_t)val;
        }
    }
    free(pyr);
    if (flags) *flags = ovf;
    return nk;
}

/* ---------------------------------------------------------- end to end */
/* pts_out (optional): the matched points handed to findEssentialMat, 2 x max_matches x 2 f32
 * (pts1 then pts2) -- lets a test replay RANSAC + recoverPose with other seeds without
 * re-extracting features. */
static int estimate_pose_ex(const uint8_t *img1, const uint8_t *img2, int W, int H,
                            const double *K, int nfeatures, int max_matches,
                            orc_pose_result *out, float *pts_out, int norm, double ratio)
{
    memset(out, 0, sizeof(*out));
    int cap = nfeatures + 64;
    orc_keypoint *k1 = (orc_keypoint *)malloc(sizeof(orc_keypoint) * 2 * (size_t)cap), *k2 = k1 + cap;
    uint8_t *d1 = (uint8_t *)malloc(64 * (size_t)cap), *d2 = d1 + 32 * (size_t)cap;
    uint32_t f1 = 0, f2 = 0;
    int n1 = orc_orb_detect_and_compute_ex(img1, W, H, nfeatures, 15, k1, d1, cap, &f1);
    int n2 = orc_orb_detect_and_compute_ex(img2, W, H, nfeatures, 15, k2, d2, cap, &f2);
    out->n_kp1 = n1; out->n_kp2 = n2; out->overflow = (int32_t)(f1 | f2);
    int mm = max_matches >= 0 ? max_matches : cap;
    int32_t *qi = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)cap), *ti = qi + cap, *di = ti + cap;
    float *p1 = (float *)malloc(sizeof(float) * 4 * (size_t)cap), *p2 = p1 + 2 * (size_t)cap;
    if (n1 == 0 || n2 == 0) { out->status = ORC_NO_DESCRIPTORS; goto done; }
    int M;
    if (norm == 1) {
        /* BFMatcher(NORM_L2) on the ORB bytes (pose_estimator.py:115-131 builds any norm for any extractor) */
        float *fd = (float *)malloc(sizeof(float) * 32 * (size_t)(n1 + n2)), *dist = (float *)malloc(sizeof(float) * (size_t)cap);
        for (size_t e = 0; e < 32 * (size_t)n1; ++e) fd[e] = (float)d1[e];
        for (size_t e = 0; e < 32 * (size_t)n2; ++e) fd[32 * (size_t)n1 + e] = (float)d2[e];
        M = ratio > 0. ? orc_match_l2_ratio(fd, n1, fd + 32 * (size_t)n1, n2, 32, ratio, mm, qi, ti, dist)
                       : orc_match_l2(fd, n1, fd + 32 * (size_t)n1, n2, 32, mm, qi, ti, dist);
        free(fd); free(dist);
    } else
        M = ratio > 0. ? orc_match_hamming_ratio(d1, n1, d2, n2, ratio, mm, qi, ti, di)
                       : orc_match_hamming(d1, n1, d2, n2, mm, qi, ti, di);
    out->n_matches = M;
    if (M < 5) { out->status = ORC_INSUFFICIENT_MATCHES; goto done; }
    for (int i = 0; i < M; ++i) {
        p1[2 * i] = k1[qi[i]].x; p1[2 * i + 1] = k1[qi[i]].y;
        p2[2 * i] = k2[ti[i]].x; p2[2 * i + 1] = k2[ti[i]].y;
    }
    if (pts_out && max_matches >= 0) {
        memcpy(pts_out, p1, sizeof(float) * 2 * (size_t)M);
        memcpy(pts_out + 2 * (size_t)max_matches, p2, sizeof(float) * 2 * (size_t)M);
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
    free(p1); free(qi); free(d1); free(k1);
    return out->status;
}

int orc_estimate_pose(const uint8_t *img1, const uint8_t *img2, int W, int H,
                      const double *K, int nfeatures, int max_matches,
                      orc_pose_result *out)
{
    return estimate_pose_ex(img1, img2, W, H, K, nfeatures, max_matches, out, NULL, 0, 0.);
}

int orc_estimate_pose_orb(const uint8_t *img1, const uint8_t *img2, int W, int H,
                          const double *K, int nfeatures, int max_matches, int norm, orc_pose_result *out)
{
    return estimate_pose_ex(img1, img2, W, H, K, nfeatures, max_matches, out, NULL, norm, 0.);
}

int orc_estimate_pose_orb_ratio(const uint8_t *img1, const uint8_t *img2, int W, int H,
                                const double *K, int nfeatures, int max_matches, int norm, double ratio, orc_pose_result *out)
{
    return estimate_pose_ex(img1, img2, W, H, K, nfeatures, max_matches, out, NULL, norm, ratio);
}

#include <pthread.h>
typedef struct {
    const uint8_t *i1, *i2; int B, W, H; const double *K; int nf, mm; orc_pose_result *out; int tid, nt; float *pts;
} job_t;
static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    size_t sz = (size_t)j->W * j->H;
    for (int b = j->tid; b < j->B; b += j->nt)
        estimate_pose_ex(j->i1 + sz * b, j->i2 + sz * b, j->W, j->H, j->K, j->nf, j->mm, &j->out[b],
                         j->pts ? j->pts + 4 * (size_t)j->mm * b : NULL, 0, 0.);
    return NULL;
}
void orc_estimate_pose_batch_pts(const uint8_t *imgs1, const uint8_t *imgs2, int B, int W, int H,
                                 const double *K, int nfeatures, int max_matches,
                                 orc_pose_result *out, int nthreads, float *pts);
void orc_estimate_pose_batch(const uint8_t *imgs1, const uint8_t *imgs2, int B, int W, int H,
                             const double *K, int nfeatures, int max_matches,
                             orc_pose_result *out, int nthreads)
{
    orc_estimate_pose_batch_pts(imgs1, imgs2, B, W, H, K, nfeatures, max_matches, out, nthreads, NULL);
}
/* pts (optional): B x 2 x max_matches x 2 f32, see estimate_pose_ex */
void orc_estimate_pose_batch_pts(const uint8_t *imgs1, const uint8_t *imgs2, int B, int W, int H,
                                 const double *K, int nfeatures, int max_matches,
                                 orc_pose_result *out, int nthreads, float *pts)
{
    if (nthreads < 1) nthreads = 1;
    pthread_t th[256]; job_t jb[256];
    if (nthreads > 256) nthreads = 256;
    for (int t = 0; t < nthreads; ++t) {
        job_t j = {imgs1, imgs2, B, W, H, K, nfeatures, max_matches, out, t, nthreads, pts};
        jb[t] = j;
        pthread_create(&th[t], NULL, worker, &jb[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}
