// rpe_api.hip -- host side of the C-ABI (include/rpe_amd.h): handle lifecycle,
// HBM workspace, host-built tables (pyramid geometry, resize coefficients,
// RANSAC subset stream and niters table) and stage orchestration on one HIP stream.
#include "rpe_internal.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

void rpe_orb_upload_disc(const signed char *disc, int n);

static std::string g_create_err;

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char b_[512];                                                                       \
            snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (h) (h)->err = b_; else g_create_err = b_;                                       \
            return RPE_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

static int cv_round(double v) { return (int)lrint(v); }
static long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

extern "C" void rpe_default_config(rpe_config *c)
{
    memset(c, 0, sizeof(*c));
    c->abi_version = RPE_ABI_VERSION;
    c->device = 0;
    c->width = 640; c->height = 480;
    c->max_batch = 1;
    c->feature_method = RPE_FEATURE_ORB;   // pose_estimator.py:22
    c->norm_type = RPE_NORM_HAMMING;       // pose_estimator.py:23
    c->max_matches = 500;                  // pose_estimator.py:24
    c->nfeatures = 4000;                   // pose_estimator.py:25
    c->fast_threshold = 15;                // pose_estimator.py:89
    c->ransac_max_iters = 1000;
    c->ransac_prob = 0.999;                // pose_estimator.py:525
    c->ransac_threshold = 1.0;             // pose_estimator.py:526
    c->match_mode = RPE_MATCH_CROSSCHECK;  // pose_estimator.py:131 crossCheck=True
    c->match_ratio = 0.75;
    c->stl_runtime = RPE_STL_LIBSTDCXX;    // the reference's Linux cv2 wheels (Dockerfile: python:3.9-slim)
}

extern "C" int rpe_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *rpe_last_error(const rpe_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }
extern "C" int rpe_keypoint_capacity(const rpe_handle *h) { return h ? h->lay.kcap : 0; }

// ORB pyramid geometry (orb.cpp): scale_l = (float)pow(1.1f, l), size cvRound(dim/scale),
// per-level quota from the geometric series, remainder to the last level.
static void build_layout(rpe_handle *h)
{
    RpeDeviceLayout &L = h->lay;
    const int W = h->cfg.width, H = h->cfg.height;
    // SIFT with nfeatures = 0 is cv2's SIFT_create() default: no retainBest; the arrays hold RPE_SIFT_UNCAPPED_CAPACITY keypoints
    const int nf = (h->cfg.feature_method == RPE_FEATURE_SIFT && h->cfg.nfeatures == 0) ? RPE_SIFT_UNCAPPED_CAPACITY : h->cfg.nfeatures;
    const double sf = (double)1.1f;
    long long off = 0;
    int coef = 0, cand = 0;
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        RpeLevel &v = L.lv[l];
        v.scale = (float)pow(sf, (double)l);
        const float inv_scale = 1.0f / v.scale;                  // orb.cpp: Size sz(cvRound(image.cols * inv_scale), ...)
        v.w = cv_round((double)((float)W * inv_scale));
        v.h = cv_round((double)((float)H * inv_scale));
        v.pitch = (int)align_up(v.w, 16);
        v.off = off;
        off = align_up(off + (long long)v.pitch * v.h, 256);
        v.coef_off = coef;
        coef += 2 * (v.w + v.h);
    }
    L.stride = off;
    float factor = (float)(1.0 / sf);
    float nd = nf * (1 - factor) / (1 - (float)pow((double)factor, (double)RPE_NLEVELS));
    int sum = 0;
    for (int l = 0; l < RPE_NLEVELS - 1; ++l) {
        L.lv[l].quota = cv_round((double)nd);
        sum += L.lv[l].quota;
        nd *= factor;
    }
    L.lv[RPE_NLEVELS - 1].quota = nf - sum > 0 ? nf - sum : 0;
    int corner = 0;
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        // raster corner list: clamp(w h / 64, 1024, 8192) entries (the oracle mirrors this rule: orc_orb_corner_cap)
        const long long c64 = (long long)L.lv[l].w * L.lv[l].h / 64;
        L.lv[l].ccap = (int)(c64 < 1024 ? 1024 : c64 > 8192 ? 8192 : c64);
        L.lv[l].corner_off = corner;
        corner += L.lv[l].ccap;
        L.lv[l].kcap2 = 4 * L.lv[l].quota + 256;
        L.lv[l].cand_off = cand;
        cand += L.lv[l].kcap2;
    }
    L.corner_total = corner;
    L.cand_total = cand;
    L.stl = h->cfg.stl_runtime;
    L.kcap = nf + 64;
    L.fast_thr = h->cfg.fast_threshold;
}

// INTER_LINEAR_EXACT coefficients (resize.cpp interpolationLinear<ufixedpoint16>)
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

// cv::RNG (core/rand.cpp) MWC generator, used by RANSAC's getSubset
static inline uint32_t rng_next(uint64_t &st)
{
    st = (uint64_t)(uint32_t)st * 4164903690ULL + (uint32_t)(st >> 32);
    return (uint32_t)st;
}

template <typename T>
static int dmalloc(rpe_handle *h, T **p, size_t n)
{
    HIPCHK(h, hipMalloc((void **)p, n * sizeof(T)));
    return RPE_OK;
}
#define DM(h, p, n) do { int r_ = dmalloc(h, &(p), (size_t)(n)); if (r_) return r_; } while (0)

static int build_tables(rpe_handle *h)
{
    const RpeDeviceLayout &L = h->lay;
    // tiles (also records each level's run of FAST tiles in the layout)
    std::vector<RpeTile> full, fast;
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        const RpeLevel &v = L.lv[l];
        for (int y = 0; y < v.h; y += 64)
            for (int x = 0; x < v.pitch; x += 64) full.push_back({(short)l, (short)x, (short)y, 0});
        h->lay.lv[l].tile0 = (int)fast.size();
        if (v.w > 2 * RPE_EDGE && v.h > 2 * RPE_EDGE)
            // keypoints survive the border filter on [31, w-31) x [31, h-31) only; x origin dword aligned
            for (int y = RPE_EDGE; y < v.h - RPE_EDGE; y += FAST_TH)
                for (int x = RPE_EDGE & ~3; x < v.w - RPE_EDGE; x += 64) fast.push_back({(short)l, (short)x, (short)y, 0});
        h->lay.lv[l].ntile = (int)fast.size() - h->lay.lv[l].tile0;
    }
    h->n_tiles_full = (int)full.size(); h->n_tiles_fast = (int)fast.size();
    DM(h, h->d_tiles_full, full.size());
    DM(h, h->d_tiles_fast, fast.size() ? fast.size() : 1);
    HIPCHK(h, hipMemcpy(h->d_tiles_full, full.data(), full.size() * sizeof(RpeTile), hipMemcpyHostToDevice));
    if (!fast.empty()) HIPCHK(h, hipMemcpy(h->d_tiles_fast, fast.data(), fast.size() * sizeof(RpeTile), hipMemcpyHostToDevice));
    // resize coefficients
    int ncoef = L.lv[RPE_NLEVELS - 1].coef_off + 2 * (L.lv[RPE_NLEVELS - 1].w + L.lv[RPE_NLEVELS - 1].h);
    std::vector<int> coef((size_t)ncoef, 0);
    for (int l = 1; l < RPE_NLEVELS; ++l) {
        const RpeLevel &S = L.lv[l - 1], &D = L.lv[l];
        int *xo = coef.data() + D.coef_off, *xa = xo + D.w, *yo = xa + D.w, *ya = yo + D.h;
        lin_coeffs(S.w, D.w, xo, xa);
        lin_coeffs(S.h, D.h, yo, ya);
    }
    // resize tiles (PYR_TW x PYR_TH destination pixels) with the origin of their source window, levels 1..11
    {
        std::vector<RpePyrTile> pt;
        for (int l = 1; l < RPE_NLEVELS; ++l) {
            const RpeLevel &S = L.lv[l - 1], &D = L.lv[l];
            h->pyr_tile_off[l] = (int)pt.size();
            for (int y0 = 0; y0 < D.h; y0 += PYR_TH)
                for (int x0 = 0; x0 < D.pitch; x0 += PYR_TW)
                    pt.push_back({(short)x0, (short)y0, (short)(((int)(((long long)x0 * S.w) / D.w)) & ~15), (short)(((long long)y0 * S.h) / D.h)});
            h->pyr_tile_cnt[l] = (int)pt.size() - h->pyr_tile_off[l];
        }
        DM(h, h->d_pyr_tiles, pt.size() ? pt.size() : 1);
        if (!pt.empty()) HIPCHK(h, hipMemcpy(h->d_pyr_tiles, pt.data(), pt.size() * sizeof(RpePyrTile), hipMemcpyHostToDevice));
    }
    // the resize kernel stages a fixed PYR_ROWS-row x (4 PYR_DW)-byte source footprint per PYR_TW x PYR_TH tile (origin 16-B aligned),
    // anchored at floor(scale * tile origin); verify the tables fit it for every tile
    for (int l = 1; l < RPE_NLEVELS; ++l) {
        const RpeLevel &S = L.lv[l - 1], &D = L.lv[l];
        const int *xo = coef.data() + D.coef_off, *yo = xo + 2 * D.w;
        for (int x0 = 0; x0 < D.w; x0 += PYR_TW) {
            int a0 = ((int)(((long long)x0 * S.w) / D.w)) & ~15, xl = x0 + PYR_TW - 1 < D.w ? x0 + PYR_TW - 1 : D.w - 1;
            int hi = xo[xl] + 1 < S.w ? xo[xl] + 1 : S.w - 1;
            if (xo[x0] < a0 || hi - a0 >= PYR_DW * 4) { h->err = "pyramid footprint bound violated (x)"; return RPE_ERR_INVALID; }
        }
        for (int y0 = 0; y0 < D.h; y0 += PYR_TH) {
            int s0 = (int)(((long long)y0 * S.h) / D.h), yl = y0 + PYR_TH - 1 < D.h ? y0 + PYR_TH - 1 : D.h - 1;
            int hi = yo[yl] + 1 < S.h ? yo[yl] + 1 : S.h - 1;
            if (yo[y0] < s0 || hi - s0 >= PYR_ROWS) { h->err = "pyramid footprint bound violated (y)"; return RPE_ERR_INVALID; }
        }
    }
    // device form: (offset, weight) packed into one dword per destination column / row (offset < 65536, weight <= 256).
    // Per level: [align128(w) packed x][align64(h) packed y], padded with the last entry, every run 16-B aligned: a lane
    // of the resize kernel fetches its 4 columns with one 16-B load and its 8 rows with two, without clamps.
    {
        int dtotal = 0;
        for (int l = 1; l < RPE_NLEVELS; ++l) {
            h->lay.lv[l].dcoef_off = dtotal;
            dtotal += ((L.lv[l].w + 127) & ~127) + ((L.lv[l].h + 63) & ~63);
        }
        h->lay.lv[0].dcoef_off = 0;
        std::vector<int> packed((size_t)dtotal + 4, 0);
        for (int l = 1; l < RPE_NLEVELS; ++l) {
            const RpeLevel &D = L.lv[l];
            const int *xo = coef.data() + D.coef_off, *xa = xo + D.w, *yo = xa + D.w, *ya = yo + D.h;
            const int xw = (D.w + 127) & ~127, yh = (D.h + 63) & ~63;
            int *px = packed.data() + D.dcoef_off, *py = px + xw;
            for (int x = 0; x < xw; ++x) { const int k = x < D.w ? x : D.w - 1; px[x] = xo[k] | (xa[k] << 16); }
            for (int y = 0; y < yh; ++y) { const int k = y < D.h ? y : D.h - 1; py[y] = yo[k] | (ya[k] << 16); }
        }
        DM(h, h->d_coef, packed.size());
        HIPCHK(h, hipMemcpy(h->d_coef, packed.data(), sizeof(int) * packed.size(), hipMemcpyHostToDevice));
    }
    // intensity-centroid disc (orb.cpp umax table)
    {
        int umax[RPE_HALF_PATCH + 2];
        int v, v0, vmax = (int)floor(RPE_HALF_PATCH * sqrt(2.f) / 2 + 1);
        int vmin = (int)ceil(RPE_HALF_PATCH * sqrt(2.f) / 2);
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round(sqrt((double)RPE_HALF_PATCH * RPE_HALF_PATCH - v * v));
        for (v = RPE_HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        std::vector<signed char> disc;
        for (int vv = -RPE_HALF_PATCH; vv <= RPE_HALF_PATCH; ++vv) {
            int d = umax[abs(vv)];
            for (int u = -d; u <= d; ++u) { disc.push_back((signed char)u); disc.push_back((signed char)vv); }
        }
        if (disc.size() / 2 > 768) { h->err = "disc table overflow"; return RPE_ERR_INVALID; }
        rpe_orb_upload_disc(disc.data(), (int)(disc.size() / 2));
    }
    // RANSAC subset stream per M (ptsetreg.cpp getSubset; RNG seeded (uint64)-1 per run)
    const int mm = h->cfg.max_matches, iters = h->cfg.ransac_max_iters;
    {
        std::vector<unsigned short> sub((size_t)(mm + 1) * iters * 5, 0);
        for (int M = 6; M <= mm; ++M) {
            uint64_t st = 0xFFFFFFFFFFFFFFFFULL;
            unsigned short *s = sub.data() + (size_t)M * iters * 5;
            for (int it = 0; it < iters; ++it, s += 5)
                for (int i = 0; i < 5; ++i) {
                    int v, dup;
                    do {
                        v = (int)(rng_next(st) % (uint32_t)M);
                        dup = 0;
                        for (int k = 0; k < i; ++k) if (s[k] == v) dup = 1;
                    } while (dup);
                    s[i] = (unsigned short)v;
                }
        }
        DM(h, h->d_subsets, sub.size());
        HIPCHK(h, hipMemcpy(h->d_subsets, sub.data(), sub.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    // RANSACUpdateNumIters(p, ep, 5, niters) terms per (M, goodCount)
    {
        size_t n = (size_t)(mm + 1) * (mm + 2) / 2;
        std::vector<double> den(n, 0.); std::vector<int> rnd(n, 0);
        double p = h->cfg.ransac_prob;
        p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
        double num0 = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
        h->nit_num = log(num0);
        for (int M = 1; M <= mm; ++M)
            for (int g = 0; g <= M; ++g) {
                size_t idx = (size_t)M * (M + 1) / 2 + g;
                double ep = (double)(M - g) / M;
                ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
                double denom = 1. - pow(1. - ep, 5);
                if (denom < DBL_MIN) { den[idx] = 0.; rnd[idx] = -1; continue; }
                denom = log(denom);
                den[idx] = denom;
                rnd[idx] = denom >= 0 ? 0 : cv_round(h->nit_num / denom);
            }
        DM(h, h->d_nit_denom, n); DM(h, h->d_nit_round, n);
        HIPCHK(h, hipMemcpy(h->d_nit_denom, den.data(), n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->d_nit_round, rnd.data(), n * sizeof(int), hipMemcpyHostToDevice));
    }
    return RPE_OK;
}

static int alloc_workspace(rpe_handle *h)
{
    const RpeDeviceLayout &L = h->lay;
    const size_t NI = (size_t)h->n_img_cap, B = (size_t)h->cfg.max_batch, mm = (size_t)h->cfg.max_matches;
    const size_t NIo = h->cfg.feature_method == RPE_FEATURE_SIFT ? 1 : NI;   // ORB pyramid buffers are unused by SIFT handles
    DM(h, h->d_pyr, NIo * L.stride); DM(h, h->d_bufA, L.stride);
    HIPCHK(h, hipMemset(h->d_bufA, 0, L.stride));
    HIPCHK(h, hipMemset(h->d_pyr, 0, NIo * L.stride));
    const size_t ntf = h->n_tiles_fast > 0 ? (size_t)h->n_tiles_fast : 1;
    DM(h, h->d_tile_cnt, NIo * ntf); DM(h, h->d_tile_list, NIo * ntf * RPE_FAST_TILE_CAP);
    HIPCHK(h, hipMemset(h->d_tile_cnt, 0, NIo * ntf * sizeof(int)));
    const size_t img = (size_t)h->cfg.width * h->cfg.height;
    DM(h, h->d_stage1, (B + 1) * img); DM(h, h->d_stage2, B * img);   // +1: a stream of max_batch pairs has max_batch + 1 frames
    DM(h, h->d_hist, NI * RPE_NLEVELS * 256);
    DM(h, h->d_cand_xy, NI * L.cand_total); DM(h, h->d_cand_resp, NI * L.cand_total);
    DM(h, h->d_cand_count, NI * RPE_NLEVELS);
    DM(h, h->d_corner, NIo * L.corner_total); DM(h, h->d_corner_count, NI * RPE_NLEVELS); DM(h, h->d_kp_lvl_count, NI * RPE_NLEVELS);
    DM(h, h->d_kp_xy, NI * L.kcap); DM(h, h->d_kp_resp, NI * L.kcap); DM(h, h->d_kp_angle, NI * L.kcap);
    DM(h, h->d_kp_pt, NI * L.kcap); DM(h, h->d_kp_cs, NI * L.kcap); DM(h, h->d_kp_count, NI);
    DM(h, h->d_ovf, NI);
    HIPCHK(h, hipMemset(h->d_ovf, 0, NI * sizeof(unsigned)));
    DM(h, h->d_desc, NI * L.kcap * h->desc_bytes);
    HIPCHK(h, hipMemset(h->d_kp_pt, 0, NI * L.kcap * sizeof(float2)));
    HIPCHK(h, hipMemset(h->d_kp_count, 0, NI * sizeof(int)));
    DM(h, h->d_m_q, B * mm); DM(h, h->d_m_t, B * mm); DM(h, h->d_m_d, B * mm);
    // results of a batch in ONE block [R 9B f64 | t 3B f64 | inliers B | status B | n_matches B]: rpe_fetch_results is one
    // device-to-host copy into pinned memory (five copies into pageable memory left the GPU idle for ~100 us per step)
    DM(h, h->d_resblk, (size_t)B * RPE_RESULT_BYTES);
    h->d_R = (double *)h->d_resblk; h->d_t = h->d_R + (size_t)B * 9;
    h->d_inliers = (int *)(h->d_t + (size_t)B * 3); h->d_status = h->d_inliers + B; h->d_m_n = h->d_status + B;
    HIPCHK(h, hipHostMalloc((void **)&h->h_resblk, (size_t)B * RPE_RESULT_BYTES));
    DM(h, h->d_pts1, B * mm); DM(h, h->d_pts2, B * mm);
    if (h->cfg.norm_type == RPE_NORM_L2) { DM(h, h->d_m_best, B * L.kcap); DM(h, h->d_m_best2, B * L.kcap); DM(h, h->d_m_norm, 2 * NI * L.kcap); }
    else {
        // small batches (and, when the fused matcher's LDS would not fit, all batches) keep the election words in HBM
        const bool big = (size_t)L.kcap * 8 + 32768 > 65536;
        const size_t bs = big || B < RPE_MATCH_SPLIT_PAIRS ? B : RPE_MATCH_SPLIT_PAIRS;
        DM(h, h->d_hm_best, bs * L.kcap); DM(h, h->d_hm_row, bs * L.kcap);
    }
    DM(h, h->d_n1, B * mm); DM(h, h->d_n2, B * mm);
    DM(h, h->d_rstate, B); DM(h, h->d_found, B);
    DM(h, h->d_models, B * RPE_RANSAC_MAXCHUNK * RPE_MAX_MODELS * 9);
    DM(h, h->d_hyp, B * (RPE_RANSAC_MAXCHUNK / 64) * 88 * 64);
    DM(h, h->d_nmodels, B * RPE_RANSAC_MAXCHUNK);
    DM(h, h->d_counts, B * RPE_RANSAC_MAXCHUNK * RPE_MAX_MODELS);
    DM(h, h->d_mask, B * mm);
    DM(h, h->d_E, B * 9);
    DM(h, h->d_K, 9);
    return RPE_OK;
}

extern "C" int rpe_create(const rpe_config *cfg, rpe_handle **out)
{
    if (!cfg || !out) { g_create_err = "null argument"; return RPE_ERR_INVALID; }
    *out = nullptr;
    if (cfg->abi_version != RPE_ABI_VERSION) { g_create_err = "ABI version mismatch"; return RPE_ERR_INVALID; }
    const bool is_sift = cfg->feature_method == RPE_FEATURE_SIFT;
    if (cfg->feature_method != RPE_FEATURE_ORB && !is_sift) { g_create_err = "unknown feature method"; return RPE_ERR_INVALID; }
    if (cfg->norm_type != RPE_NORM_HAMMING && cfg->norm_type != RPE_NORM_L2) { g_create_err = "unknown norm type"; return RPE_ERR_INVALID; }
    if (is_sift && cfg->norm_type == RPE_NORM_HAMMING) {
        // cv2 builds this matcher but its match() rejects float descriptors (batchDistance: NORM_HAMMING needs CV_8U)
        g_create_err = "NORM_HAMMING needs 8-bit descriptors: SIFT descriptors are float (cv2 raises in match())"; return RPE_ERR_INVALID;
    }
    if (cfg->match_mode != RPE_MATCH_CROSSCHECK && cfg->match_mode != RPE_MATCH_RATIO) { g_create_err = "unknown match mode"; return RPE_ERR_INVALID; }
    if (cfg->norm_type == RPE_NORM_L2 && cfg->nfeatures > RPE_SIFT_UNCAPPED_CAPACITY) {
        // the L2 matcher sorts next_pow2(nfeatures + 64) 64-bit keys in LDS: 16384 keys = 128 KB of the CU's 160
        g_create_err = "NORM_L2: nfeatures must be <= 16320"; return RPE_ERR_INVALID;
    }
    if (cfg->stl_runtime != RPE_STL_LIBSTDCXX && cfg->stl_runtime != RPE_STL_MSVC) { g_create_err = "unknown stl_runtime"; return RPE_ERR_INVALID; }
    if (cfg->match_mode == RPE_MATCH_RATIO && !(cfg->match_ratio > 0. && cfg->match_ratio <= 1.)) { g_create_err = "match_ratio must be in (0, 1]"; return RPE_ERR_INVALID; }
    if (is_sift && (cfg->nfeatures > RPE_SIFT_UNCAPPED_CAPACITY || cfg->nfeatures < 0 || cfg->width > 4000 || cfg->height > 4000)) {
        g_create_err = "SIFT: nfeatures must be 0 (no cap: cv2's SIFT_create()) or a cap <= 16320, and the image <= 4000 px"; return RPE_ERR_INVALID;
    }
    if (cfg->width < 96 || cfg->height < 96 || cfg->width > 4095 || cfg->height > 4095 || cfg->max_batch < 1 ||
        (!is_sift && (cfg->nfeatures < 1 || cfg->nfeatures > 8000)) || cfg->max_matches < 5 || cfg->max_matches > 8064 ||
        cfg->ransac_max_iters < 1 || cfg->ransac_max_iters > 4096 || cfg->fast_threshold < 1 || cfg->fast_threshold > 254) {
        g_create_err = "configuration out of supported range"; return RPE_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device available: the MI355X path has no CPU fallback"; return RPE_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "bad device ordinal"; return RPE_ERR_INVALID; }
    rpe_handle *h = new rpe_handle();
    h->cfg = *cfg;
    h->n_img_cap = 2 * cfg->max_batch;
    h->desc_bytes = is_sift ? 128 : 32;
    int rc = RPE_OK;
    do {
        if (hipSetDevice(cfg->device) != hipSuccess) { h->err = "hipSetDevice failed"; rc = RPE_ERR_HIP; break; }
        if (hipStreamCreate(&h->stream) != hipSuccess) { h->err = "hipStreamCreate failed"; rc = RPE_ERR_HIP; break; }
        build_layout(h);
        if ((rc = build_tables(h)) != RPE_OK) break;
        if ((rc = alloc_workspace(h)) != RPE_OK) break;
        if (is_sift && (rc = rpe_sift_create(h)) != RPE_OK) break;
        for (int i = 0; i <= RPE_STAGE_COUNT; ++i)
            if (hipEventCreate(&h->ev[i]) != hipSuccess) { h->err = "hipEventCreate failed"; rc = RPE_ERR_HIP; break; }
    } while (0);
    if (rc != RPE_OK) { g_create_err = h->err; rpe_destroy(h); return rc; }
    *out = h;
    return RPE_OK;
}

extern "C" void rpe_destroy(rpe_handle *h)
{
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    rpe_sift_destroy(h);
    for (auto &g : h->graphs) { hipGraphExecDestroy(g.exec); hipGraphDestroy(g.graph); }
    void *ptrs[] = {h->d_tiles_full, h->d_tiles_fast, h->d_coef, h->d_pyr_tiles, h->d_pyr, h->d_bufA, h->d_tile_list, h->d_tile_cnt, h->d_stage1, h->d_stage2,
                    h->d_hist, h->d_cand_xy, h->d_cand_resp, h->d_cand_count, h->d_kp_xy, h->d_kp_resp, h->d_kp_angle,
                    h->d_kp_pt, h->d_kp_cs, h->d_kp_count, h->d_desc, h->d_m_q, h->d_m_t, h->d_m_d, h->d_resblk, h->d_pts1, h->d_pts2,
                    h->d_subsets, h->d_nit_denom, h->d_nit_round, h->d_rstate, h->d_n1, h->d_n2, h->d_found, h->d_models, h->d_hyp, h->d_counts,
                    h->d_nmodels, h->d_mask, h->d_E, h->d_K, h->d_m_best, h->d_m_best2, h->d_m_norm, h->d_hm_best, h->d_hm_row, h->d_ovf, h->d_corner, h->d_corner_count, h->d_kp_lvl_count};
    for (void *p : ptrs) if (p) hipFree(p);
    if (h->h_resblk) hipHostFree(h->h_resblk);
    for (void *p : h->user_allocs) hipFree(p);
    for (int i = 0; i <= RPE_STAGE_COUNT; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
    for (int c = 0; c < 8; ++c) if (h->ev_up[c]) hipEventDestroy(h->ev_up[c]);
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

// ------------------------------------------------------------ device buffers
extern "C" int rpe_device_malloc(rpe_handle *h, size_t bytes, void **d_ptr)
{
    if (!h || !d_ptr) return RPE_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMalloc(d_ptr, bytes));
    h->user_allocs.push_back(*d_ptr);
    return RPE_OK;
}
extern "C" int rpe_device_free(rpe_handle *h, void *d_ptr)
{
    if (!h) return RPE_ERR_INVALID;
    for (size_t i = 0; i < h->user_allocs.size(); ++i)
        if (h->user_allocs[i] == d_ptr) { h->user_allocs.erase(h->user_allocs.begin() + i); HIPCHK(h, hipFree(d_ptr)); return RPE_OK; }
    h->err = "rpe_device_free: unknown pointer";
    return RPE_ERR_INVALID;
}
extern "C" int rpe_host_alloc(rpe_handle *h, size_t bytes, void **h_ptr)
{
    if (!h || !h_ptr) return RPE_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return RPE_OK;
}
extern "C" int rpe_host_free(rpe_handle *h, void *h_ptr)
{
    if (!h) return RPE_ERR_INVALID;
    HIPCHK(h, hipHostFree(h_ptr));
    return RPE_OK;
}
extern "C" int rpe_host_register(rpe_handle *h, void *h_ptr, size_t bytes)
{
    if (!h || !h_ptr) return RPE_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return RPE_OK;
}
extern "C" int rpe_host_unregister(rpe_handle *h, void *h_ptr)
{
    if (!h || !h_ptr) return RPE_ERR_INVALID;
    HIPCHK(h, hipHostUnregister(h_ptr));
    return RPE_OK;
}
extern "C" int rpe_memcpy_h2d(rpe_handle *h, void *d, const void *s, size_t n)
{
    if (!h) return RPE_ERR_INVALID;
    HIPCHK(h, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RPE_OK;
}
extern "C" int rpe_memcpy_d2h(rpe_handle *h, void *d, const void *s, size_t n)
{
    if (!h) return RPE_ERR_INVALID;
    HIPCHK(h, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RPE_OK;
}
extern "C" int rpe_synchronize(rpe_handle *h)
{
    if (!h) return RPE_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RPE_OK;
}

// ------------------------------------------------------------- orchestration

// copies level 0 of every image into the pyramid buffer (device to device)
static int load_level0(rpe_handle *h, const uint8_t *d_a, const uint8_t *d_b, int na, int nb)
{
    const int W = h->cfg.width, H = h->cfg.height;
    const RpeLevel &v = h->lay.lv[0];
    for (int i = 0; i < na + nb; ++i) {
        const uint8_t *src = i < na ? d_a + (size_t)i * W * H : d_b + (size_t)(i - na) * W * H;
        uint8_t *dst = h->d_pyr + (size_t)i * h->lay.stride + v.off;
        HIPCHK(h, hipMemcpy2DAsync(dst, v.pitch, src, W, W, H, hipMemcpyDeviceToDevice, h->stream));
    }
    return RPE_OK;
}

// strided batch copy kernel-free variant: when pitch == W one 2-D copy moves all images
static int load_level0_fast(rpe_handle *h, const uint8_t *d_a, const uint8_t *d_b, int na, int nb)
{
    const int W = h->cfg.width, H = h->cfg.height;
    const RpeLevel &v = h->lay.lv[0];
    if (v.pitch != W) return load_level0(h, d_a, d_b, na, nb);
    const size_t img = (size_t)W * H;
    if (na) HIPCHK(h, hipMemcpy2DAsync(h->d_pyr + v.off, h->lay.stride, d_a, img, img, na, hipMemcpyDeviceToDevice, h->stream));
    if (nb) HIPCHK(h, hipMemcpy2DAsync(h->d_pyr + (size_t)na * h->lay.stride + v.off, h->lay.stride, d_b, img, img, nb,
                                       hipMemcpyDeviceToDevice, h->stream));
    return RPE_OK;
}

static int run_orb(rpe_handle *h, const uint8_t *d_a, const uint8_t *d_b, int na, int nb)
{
    const int n = na + nb;
    MARK(h, RPE_STAGE_PYRAMID);
    HIPCHK(h, hipMemsetAsync(h->d_ovf, 0, sizeof(unsigned) * (size_t)n, h->stream));
    // Level 0 is the input itself: when its pitch equals the image width and the batches are 16-B aligned the kernels read
    // it in place (rpe_level_base) -- the device-to-device copy into the pyramid buffer was 1.26 GB of HBM traffic and
    // 0.22 ms per 1024 VGA pairs.  Other widths / unaligned batches take the copy.
    const bool direct = h->lay.lv[0].pitch == h->cfg.width && (((uintptr_t)d_a | (uintptr_t)(nb ? d_b : d_a)) & 15) == 0;
    h->lay.in_a = direct ? d_a : nullptr; h->lay.in_b = direct ? d_b : nullptr;
    h->lay.in_na = na; h->lay.in_img = h->cfg.width * h->cfg.height;
    h->level0_slots = n;
    if (!direct) {
        int rc = load_level0_fast(h, d_a, d_b, na, nb);
        if (rc) return rc;
    }
    rpe_launch_pyramid(h, n);
    MARK(h, RPE_STAGE_FAST);      rpe_launch_fast(h, n);
    MARK(h, RPE_STAGE_NMS);       rpe_launch_nms(h, n);
    MARK(h, RPE_STAGE_SELECT);    rpe_launch_select(h, n);
    MARK(h, RPE_STAGE_HARRIS);    rpe_launch_harris(h, n);
    MARK(h, RPE_STAGE_KEYPOINTS); rpe_launch_keypoints(h, n);
    MARK(h, RPE_STAGE_ANGLE);     rpe_launch_angle(h, n);
    MARK(h, RPE_STAGE_BLUR);      // fused into the per-keypoint kernel (the whole-level blur runs on demand in rpe_orb_debug_fetch)
    MARK(h, RPE_STAGE_DESCRIBE);  rpe_launch_describe(h, n);
    MARK(h, RPE_STAGE_MATCH);
    HIPCHK(h, hipGetLastError());
    return RPE_OK;
}

static int set_K(rpe_handle *h, const double K[9])
{
    if (h->K_valid && memcmp(h->K_last, K, sizeof(double) * 9) == 0) return RPE_OK;      // same camera as the last batch: already resident
    memcpy(h->K_last, K, sizeof(double) * 9);
    h->K_valid = false;
    HIPCHK(h, hipMemcpyAsync(h->d_K, h->K_last, sizeof(double) * 9, hipMemcpyHostToDevice, h->stream));
    h->K_valid = true;
    return RPE_OK;
}

// feature extraction + matching + geometry of `pairs` pairs whose images sit in slots (p, img2_base + p)
static int run_pairs(rpe_handle *h, const uint8_t *d_a, const uint8_t *d_b, int na, int nb, int pairs, int img2_base)
{
    struct Guard {                          // the launchers read h->img2_base; never leave a stream's value behind
        rpe_handle *h; ~Guard() { h->img2_base = 0; }
    } guard{h};
    h->img2_base = img2_base;
    h->last_pairs = pairs; h->last_img2_base = img2_base;
    h->last_chunked = false;
    int rc;
    if (h->cfg.feature_method == RPE_FEATURE_SIFT) {
        if ((rc = rpe_sift_run(h, d_a, d_b, na, nb)) != RPE_OK) return rc;             // records PYRAMID .. DESCRIBE
        MARK(h, RPE_STAGE_MATCH);
    } else {
        if ((rc = run_orb(h, d_a, d_b, na, nb)) != RPE_OK) return rc;
    }
    if (h->cfg.norm_type == RPE_NORM_L2) rpe_launch_match_l2(h, pairs);
    else rpe_launch_match(h, pairs);
    MARK(h, RPE_STAGE_RANSAC);
    rpe_launch_ransac(h, pairs, false);
    MARK(h, RPE_STAGE_POSE);
    rpe_launch_pose(h, pairs, true);
    if (h->profiling) { hipEventRecord(h->ev[RPE_STAGE_COUNT], h->stream); h->ev_valid = true; }
    HIPCHK(h, hipGetLastError());
    return RPE_OK;
}

extern "C" int rpe_enqueue_batch_device(rpe_handle *h, const uint8_t *d_imgs1, const uint8_t *d_imgs2, int B, const double K[9])
{
    if (!h || !d_imgs1 || !d_imgs2 || !K || B < 1) return RPE_ERR_INVALID;
    if (B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = set_K(h, K);
    if (rc) return rc;
    // Small batches are launch-bound on the HOST (one pair: ~50 launches for 0.6 ms of GPU work): the sequence is captured
    // once per (input buffers, B) and replayed as a hipGraph.  Everything in it is stream-ordered kernels and memsets whose
    // arguments depend only on the handle, the two buffers and B; K travels outside (set_K above).  Not while profiling (the
    // stage events are host-side records), not for SIFT (its launch sequence reads back counts), RPE_NO_GRAPH=1 turns it off.
    static const bool no_graph = getenv("RPE_NO_GRAPH") != nullptr;
    if (B > RPE_GRAPH_MAX_PAIRS || h->profiling || h->cfg.feature_method != RPE_FEATURE_ORB || no_graph)
        return run_pairs(h, d_imgs1, d_imgs2, B, B, B, B);
    for (auto &g : h->graphs)
        if (g.a == d_imgs1 && g.b == d_imgs2 && g.B == B) {
            h->last_pairs = B; h->last_img2_base = B; h->last_chunked = false;
            h->lay.in_na = B; h->level0_slots = 2 * B;
            const bool direct = h->lay.lv[0].pitch == h->cfg.width && (((uintptr_t)d_imgs1 | (uintptr_t)d_imgs2) & 15) == 0;
            h->lay.in_a = direct ? d_imgs1 : nullptr; h->lay.in_b = direct ? d_imgs2 : nullptr;
            HIPCHK(h, hipGraphLaunch(g.exec, h->stream));
            return RPE_OK;
        }
    rpe_handle::GraphEntry e{d_imgs1, d_imgs2, B, nullptr, nullptr};
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return run_pairs(h, d_imgs1, d_imgs2, B, B, B, B);
    }
    rc = run_pairs(h, d_imgs1, d_imgs2, B, B, B, B);
    const hipError_t ce = hipStreamEndCapture(h->stream, &e.graph);
    if (rc != RPE_OK || ce != hipSuccess || !e.graph || hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (e.graph) hipGraphDestroy(e.graph);
        return rc != RPE_OK ? rc : run_pairs(h, d_imgs1, d_imgs2, B, B, B, B);       // capture unavailable: plain launches
    }
    if (h->graphs.size() >= 4) {                                                    // a handful of (buffers, B) combinations at most
        hipGraphExecDestroy(h->graphs.front().exec); hipGraphDestroy(h->graphs.front().graph);
        h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back(e);
    HIPCHK(h, hipGraphLaunch(e.exec, h->stream));
    return RPE_OK;
}

// Consecutive-frame stream (reference batch_processor.py:71-109): F frames -> F-1
// pairs (i, i+1).  Features are extracted ONCE per frame (the reference extracts every interior
// frame twice, batch_processor.py:79,92); pair p reads image slots p and p+1.
extern "C" int rpe_enqueue_stream_device(rpe_handle *h, const uint8_t *d_frames, int F, const double K[9])
{
    if (!h || !d_frames || !K || F < 2) return RPE_ERR_INVALID;
    if (F > h->n_img_cap || F - 1 > h->cfg.max_batch) { h->err = "stream longer than the handle capacity (frames <= 2*max_batch, pairs <= max_batch)"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = set_K(h, K);
    if (rc) return rc;
    return run_pairs(h, d_frames, d_frames, F, 0, F - 1, 1);
}

extern "C" int rpe_estimate_stream(rpe_handle *h, const uint8_t *h_frames, int F, const double K[9],
                                   double *R, double *t, int32_t *inliers, int32_t *n_matches, int32_t *status)
{
    if (!h || !h_frames || F < 2) return RPE_ERR_INVALID;
    if (F > h->n_img_cap || F - 1 > h->cfg.max_batch) { h->err = "stream longer than the handle capacity (frames <= 2*max_batch, pairs <= max_batch)"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t img = (size_t)h->cfg.width * h->cfg.height;
    // d_stage1 holds max_batch + 1 frames: every legal stream fits the persistent staging buffer
    HIPCHK(h, hipMemcpyAsync(h->d_stage1, h_frames, img * F, hipMemcpyHostToDevice, h->stream));
    int rc = rpe_enqueue_stream_device(h, h->d_stage1, F, K);
    if (rc) return rc;
    return rpe_fetch_results(h, F - 1, R, t, inliers, n_matches, status);
}

// ---------------------------------------------------------------- image ingest
// cv2.cvtColor(BGR2GRAY) (reference src/utils/image_loader.py:27-28), 8-bit path of OpenCV's color_rgb:
// (B*3735 + G*19235 + R*9798 + (1 << 14)) >> 15.  HBM-bound streaming kernel: a lane turns 16 pixels
// (three 16-B loads) into one 16-B store; 4 algorithmic bytes per pixel.
__global__ __launch_bounds__(256) void bgr_to_gray_kernel(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray, size_t n_pixels,
                                                          unsigned w0, unsigned w1, unsigned w2)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;        // group of 16 pixels
    const size_t p0 = g * 16;
    if (p0 >= n_pixels) return;
    if (p0 + 16 <= n_pixels) {
        const uint4 *src = (const uint4 *)(bgr + p0 * 3);
        const uint4 a = src[0], b = src[1], c = src[2];
        const unsigned wd[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        unsigned out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 3 * i;
            const unsigned c0 = (wd[o >> 2] >> (8 * (o & 3))) & 255u;
            const unsigned c1 = (wd[(o + 1) >> 2] >> (8 * ((o + 1) & 3))) & 255u;
            const unsigned c2 = (wd[(o + 2) >> 2] >> (8 * ((o + 2) & 3))) & 255u;
            out[i >> 2] |= ((c0 * w0 + c1 * w1 + c2 * w2 + 16384u) >> 15) << (8 * (i & 3));
        }
        *(uint4 *)(gray + p0) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
        for (size_t p = p0; p < n_pixels; ++p)
            gray[p] = (uint8_t)((bgr[3 * p] * w0 + bgr[3 * p + 1] * w1 + bgr[3 * p + 2] * w2 + 16384u) >> 15);
    }
}

extern "C" int rpe_bgr_to_gray_device(rpe_handle *h, const uint8_t *d_bgr, size_t n_pixels, int order, uint8_t *d_gray)
{
    if (!h || !d_bgr || !d_gray || (order != RPE_ORDER_BGR && order != RPE_ORDER_RGB)) return RPE_ERR_INVALID;
    if (((uintptr_t)d_bgr | (uintptr_t)d_gray) & 15) { h->err = "rpe_bgr_to_gray_device: buffers must be 16-byte aligned"; return RPE_ERR_INVALID; }
    if (n_pixels == 0) return RPE_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const unsigned wb = 3735u, wg = 19235u, wr = 9798u;
    const size_t groups = (n_pixels + 15) / 16;
    hipLaunchKernelGGL(bgr_to_gray_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, h->stream, d_bgr, d_gray, n_pixels,
                       order == RPE_ORDER_BGR ? wb : wr, wg, order == RPE_ORDER_BGR ? wr : wb);
    HIPCHK(h, hipGetLastError());
    return RPE_OK;
}

extern "C" int rpe_bgr_to_gray(rpe_handle *h, const uint8_t *h_bgr, size_t n_pixels, int order, uint8_t *h_gray)
{
    if (!h || !h_bgr || !h_gray) return RPE_ERR_INVALID;
    if (n_pixels == 0) return RPE_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    void *d_in = nullptr, *d_out = nullptr;
    HIPCHK(h, hipMalloc(&d_in, n_pixels * 3));
    if (hipMalloc(&d_out, n_pixels) != hipSuccess) { hipFree(d_in); h->err = "hipMalloc failed"; return RPE_ERR_HIP; }
    int rc = RPE_OK;
    if (hipMemcpyAsync(d_in, h_bgr, n_pixels * 3, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = RPE_ERR_HIP;
    if (!rc) rc = rpe_bgr_to_gray_device(h, (const uint8_t *)d_in, n_pixels, order, (uint8_t *)d_out);
    if (!rc && hipMemcpyAsync(h_gray, d_out, n_pixels, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = RPE_ERR_HIP;
    if (hipStreamSynchronize(h->stream) != hipSuccess && !rc) rc = RPE_ERR_HIP;
    hipFree(d_in); hipFree(d_out);
    if (rc == RPE_ERR_HIP && h->err.empty()) h->err = "rpe_bgr_to_gray: HIP failure";
    return rc;
}

extern "C" int rpe_fetch_results(rpe_handle *h, int B, double *R, double *t, int32_t *inliers, int32_t *n_matches, int32_t *status)
{
    if (!h || B < 1 || B > h->cfg.max_batch) return RPE_ERR_INVALID;
    const size_t MB = (size_t)h->cfg.max_batch;
    // the used part of every section when the batch is small, the whole block in one piece otherwise
    if ((size_t)B * 4 < MB) {
        const uint8_t *d = h->d_resblk; uint8_t *o = h->h_resblk;
        const size_t off[5] = {0, MB * 72, MB * 96, MB * 100, MB * 104}, len[5] = {(size_t)B * 72, (size_t)B * 24, (size_t)B * 4, (size_t)B * 4, (size_t)B * 4};
        for (int k = 0; k < 5; ++k) HIPCHK(h, hipMemcpyAsync(o + off[k], d + off[k], len[k], hipMemcpyDeviceToHost, h->stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(h->h_resblk, h->d_resblk, MB * RPE_RESULT_BYTES, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint8_t *o = h->h_resblk;
    if (R) memcpy(R, o, (size_t)B * 72);
    if (t) memcpy(t, o + MB * 72, (size_t)B * 24);
    if (inliers) memcpy(inliers, o + MB * 96, (size_t)B * 4);
    if (status) memcpy(status, o + MB * 100, (size_t)B * 4);
    if (n_matches) memcpy(n_matches, o + MB * 104, (size_t)B * 4);
    return RPE_OK;
}

extern "C" int rpe_fetch_overflow(rpe_handle *h, int n_pairs, uint32_t *flags)
{
    if (!h || !flags || n_pairs < 1 || n_pairs > h->cfg.max_batch) return RPE_ERR_INVALID;
    if (h->last_chunked) {
        // a chunked host batch reuses the per-image flag words chunk after chunk: they were collected per pair as the chunks finished
        if ((size_t)n_pairs > h->ovf_pairs.size()) { h->err = "rpe_fetch_overflow: more pairs than the last batch had"; return RPE_ERR_INVALID; }
        memcpy(flags, h->ovf_pairs.data(), sizeof(uint32_t) * (size_t)n_pairs);
        return RPE_OK;
    }
    if (n_pairs > h->last_pairs) { h->err = "rpe_fetch_overflow: more pairs than the last batch had"; return RPE_ERR_INVALID; }
    const int nimg = h->last_img2_base + n_pairs;
    std::vector<unsigned> ov((size_t)nimg);
    HIPCHK(h, hipMemcpyAsync(ov.data(), h->d_ovf, sizeof(unsigned) * (size_t)nimg, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int p = 0; p < n_pairs; ++p) flags[p] = ov[p] | ov[h->last_img2_base + p];
    return RPE_OK;
}

extern "C" int rpe_estimate_batch_device(rpe_handle *h, const uint8_t *d_imgs1, const uint8_t *d_imgs2, int B, const double K[9],
                                         double *R, double *t, int32_t *inliers, int32_t *n_matches, int32_t *status)
{
    int rc = rpe_enqueue_batch_device(h, d_imgs1, d_imgs2, B, K);
    if (rc) return rc;
    return rpe_fetch_results(h, B, R, t, inliers, n_matches, status);
}

extern "C" int rpe_estimate_batch(rpe_handle *h, const uint8_t *h_imgs1, const uint8_t *h_imgs2, int B, const double K[9],
                                  double *R, double *t, int32_t *inliers, int32_t *n_matches, int32_t *status)
{
    if (!h || !h_imgs1 || !h_imgs2 || B < 1) return RPE_ERR_INVALID;
    if (B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t img = (size_t)h->cfg.width * h->cfg.height;
    // Large host batches run in chunks: all uploads are queued on a copy stream, chunk c's kernels wait for its
    // 'resident' event only, so the PCIe transfer of the later chunks hides behind the kernels of the earlier ones
    // (1024 VGA pairs: 629 MB = 12 ms of copy in front of 16 ms of kernels when done in one piece).
    const int nchunks = (B >= 512 && img * (size_t)B >= ((size_t)64 << 20)) ? 4 : 1;    // 2/3/4/6/8 chunks measured: 23.1/22.4/22.1/24.1/26.7 ms
    if (nchunks == 1) {
        HIPCHK(h, hipMemcpyAsync(h->d_stage1, h_imgs1, img * B, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_stage2, h_imgs2, img * B, hipMemcpyHostToDevice, h->stream));
        return rpe_estimate_batch_device(h, h->d_stage1, h->d_stage2, B, K, R, t, inliers, n_matches, status);
    }
    if (!h->copy_stream) {
        HIPCHK(h, hipStreamCreate(&h->copy_stream));
        for (int c = 0; c < 8; ++c) HIPCHK(h, hipEventCreateWithFlags(&h->ev_up[c], hipEventDisableTiming));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));                 // the staging buffers are free again
    const int per = (B + nchunks - 1) / nchunks;
    auto upload = [&](int c) -> int {
        const int off = c * per, n = B - off < per ? B - off : per;
        if (n <= 0) return RPE_OK;
        HIPCHK(h, hipMemcpyAsync(h->d_stage1 + (size_t)off * img, h_imgs1 + (size_t)off * img, img * n, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(h, hipMemcpyAsync(h->d_stage2 + (size_t)off * img, h_imgs2 + (size_t)off * img, img * n, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(h, hipEventRecord(h->ev_up[c], h->copy_stream));
        return RPE_OK;
    };
    // The chunks run back to back on the compute stream without a host round trip in between: chunk c's results (and capacity
    // flags) are copied device-to-device into a whole-batch block right behind its kernels, the next upload is issued while
    // they run, and the host waits ONCE, at the end (a fetch per chunk cost four stream drains per batch).  Copies from
    // pageable host memory still block the calling thread while they are staged -- after the kernels of the chunk before
    // have been queued; page-locked batches (rpe_host_alloc / rpe_host_register) do not block at all.
    if (!h->d_resall) {
        const size_t MBb = (size_t)h->cfg.max_batch;
        HIPCHK(h, hipMalloc((void **)&h->d_resall, MBb * RPE_RESULT_BYTES));
        HIPCHK(h, hipMalloc((void **)&h->d_ovfall, MBb * 2 * sizeof(unsigned)));
        h->user_allocs.push_back(h->d_resall); h->user_allocs.push_back(h->d_ovfall);
    }
    const size_t MB = (size_t)h->cfg.max_batch;
    const size_t soff[5] = {0, MB * 72, MB * 96, MB * 100, MB * 104}, esz[5] = {72, 24, 4, 4, 4};
    int rc = upload(0);
    if (rc) return rc;
    for (int c = 0; c < nchunks; ++c) {
        const int off = c * per, n = B - off < per ? B - off : per;
        if (n <= 0) break;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_up[c], 0));
        if ((rc = rpe_enqueue_batch_device(h, h->d_stage1 + (size_t)off * img, h->d_stage2 + (size_t)off * img, n, K)) != RPE_OK) return rc;
        for (int k = 0; k < 5; ++k)
            HIPCHK(h, hipMemcpyAsync(h->d_resall + soff[k] + (size_t)off * esz[k], h->d_resblk + soff[k], (size_t)n * esz[k], hipMemcpyDeviceToDevice, h->stream));
        // the chunk's capacity flags (image slots p and img2_base + p of this launch), before the next chunk overwrites them
        HIPCHK(h, hipMemcpyAsync(h->d_ovfall + off, h->d_ovf, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_ovfall + MB + off, h->d_ovf + h->last_img2_base, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
        if (c + 1 < nchunks && (rc = upload(c + 1)) != RPE_OK) return rc;
    }
    std::vector<unsigned> ov(2 * MB);
    HIPCHK(h, hipMemcpyAsync(h->h_resblk, h->d_resall, MB * RPE_RESULT_BYTES, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ov.data(), h->d_ovfall, sizeof(unsigned) * 2 * MB, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint8_t *o = h->h_resblk;
    if (R) memcpy(R, o, (size_t)B * 72);
    if (t) memcpy(t, o + MB * 72, (size_t)B * 24);
    if (inliers) memcpy(inliers, o + MB * 96, (size_t)B * 4);
    if (status) memcpy(status, o + MB * 100, (size_t)B * 4);
    if (n_matches) memcpy(n_matches, o + MB * 104, (size_t)B * 4);
    h->ovf_pairs.assign((size_t)B, 0u);
    for (int p2 = 0; p2 < B; ++p2) h->ovf_pairs[(size_t)p2] = ov[(size_t)p2] | ov[MB + (size_t)p2];
    h->last_chunked = true;
    return RPE_OK;
}

extern "C" int rpe_fetch_matched_points(rpe_handle *h, int B, float *pts1, float *pts2)
{
    if (!h || B < 1 || B > h->cfg.max_batch) return RPE_ERR_INVALID;
    if (h->last_chunked) { h->err = "the last host batch ran in chunks: matched points are kept for device-resident batches (rpe_estimate_batch_device) only"; return RPE_ERR_INVALID; }
    const size_t n = sizeof(float2) * (size_t)B * h->cfg.max_matches;
    if (pts1) HIPCHK(h, hipMemcpyAsync(pts1, h->d_pts1, n, hipMemcpyDeviceToHost, h->stream));
    if (pts2) HIPCHK(h, hipMemcpyAsync(pts2, h->d_pts2, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RPE_OK;
}

// ---------------------------------------------------------------- stage API
extern "C" int rpe_orb_detect_and_compute(rpe_handle *h, const uint8_t *h_imgs, int n_images,
                                          rpe_keypoint *kps, uint8_t *desc, int32_t *counts)
{
    if (!h || !h_imgs || n_images < 1) return RPE_ERR_INVALID;
    if (h->cfg.feature_method != RPE_FEATURE_ORB) { h->err = "handle was not created for ORB"; return RPE_ERR_INVALID; }
    if (n_images > h->n_img_cap) { h->err = "n_images exceeds 2*max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t img = (size_t)h->cfg.width * h->cfg.height;
    const int na = n_images < h->cfg.max_batch ? n_images : h->cfg.max_batch, nb = n_images - na;
    HIPCHK(h, hipMemcpyAsync(h->d_stage1, h_imgs, img * na, hipMemcpyHostToDevice, h->stream));
    if (nb) HIPCHK(h, hipMemcpyAsync(h->d_stage2, h_imgs + img * na, img * nb, hipMemcpyHostToDevice, h->stream));
    int rc = run_orb(h, h->d_stage1, h->d_stage2, na, nb);
    if (rc) return rc;
    const int kcap = h->lay.kcap;
    std::vector<unsigned> xy((size_t)n_images * kcap);
    std::vector<float> resp((size_t)n_images * kcap), ang((size_t)n_images * kcap);
    std::vector<float2> pt((size_t)n_images * kcap);
    std::vector<int> cnt(n_images);
    HIPCHK(h, hipMemcpyAsync(xy.data(), h->d_kp_xy, xy.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(resp.data(), h->d_kp_resp, resp.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ang.data(), h->d_kp_angle, ang.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(pt.data(), h->d_kp_pt, pt.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(cnt.data(), h->d_kp_count, cnt.size() * 4, hipMemcpyDeviceToHost, h->stream));
    if (desc) HIPCHK(h, hipMemcpyAsync(desc, h->d_desc, (size_t)n_images * kcap * 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n_images; ++i) {
        if (counts) counts[i] = cnt[i];
        if (!kps) continue;
        for (int k = 0; k < cnt[i]; ++k) {
            size_t g = (size_t)i * kcap + k;
            rpe_keypoint &o = kps[g];
            o.x = pt[g].x; o.y = pt[g].y; o.angle = ang[g]; o.response = resp[g];
            o.octave = (int)(xy[g] >> 24); o.lx = (int)(xy[g] & 0xFFF); o.ly = (int)((xy[g] >> 12) & 0xFFF);
        }
    }
    return RPE_OK;
}

extern "C" int64_t rpe_orb_pyramid_pixels(const rpe_handle *h)
{
    if (!h) return 0;
    int64_t n = 0;
    for (int l = 0; l < RPE_NLEVELS; ++l) n += (int64_t)h->lay.lv[l].w * h->lay.lv[l].h;
    return n;
}

extern "C" int rpe_orb_debug_fetch(rpe_handle *h, int index, int which, uint8_t *h_out)
{
    if (!h || !h_out || index < 0 || index >= h->n_img_cap) return RPE_ERR_INVALID;
    if (h->cfg.feature_method != RPE_FEATURE_ORB) { h->err = "rpe_orb_debug_fetch: handle was not created for ORB"; return RPE_ERR_INVALID; }
    if (which != 0 && which != 2 && which != 3) { h->err = "rpe_orb_debug_fetch: which must be 0 (pyramid), 2 (NMS map) or 3 (blurred pyramid)"; return RPE_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    std::vector<uint8_t> tmp((size_t)h->lay.stride, 0);
    if (which == 2) {
        // the NMS map is not materialised any more: rebuild it from the tile lists of the image
        const size_t nt = (size_t)h->n_tiles_fast;
        std::vector<int> cnt(nt ? nt : 1);
        std::vector<unsigned> lst((nt ? nt : 1) * RPE_FAST_TILE_CAP);
        if (nt) {
            HIPCHK(h, hipMemcpyAsync(cnt.data(), h->d_tile_cnt + (size_t)index * nt, nt * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(lst.data(), h->d_tile_list + (size_t)index * nt * RPE_FAST_TILE_CAP, nt * RPE_FAST_TILE_CAP * sizeof(unsigned),
                                     hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        for (int l = 0; l < RPE_NLEVELS; ++l) {
            const RpeLevel &v = h->lay.lv[l];
            for (int t = v.tile0; t < v.tile0 + v.ntile; ++t)
                for (int k = 0; k < cnt[t] && k < RPE_FAST_TILE_CAP; ++k) {
                    const unsigned e = lst[(size_t)t * RPE_FAST_TILE_CAP + k];
                    tmp[v.off + (size_t)((e >> 12) & 0xFFF) * v.pitch + (e & 0xFFF)] = (uint8_t)(e >> 24);
                }
        }
    } else {
        const uint8_t *src = h->d_pyr + (size_t)index * h->lay.stride;
        if (h->lay.in_a) {
            // level 0 of the last run was read in place: bring this image's copy into the pyramid-shaped buffer (the
            // batches handed to the last enqueue must still be alive, as they are for the host-buffer entry points)
            if (index >= h->level0_slots) { h->err = "rpe_orb_debug_fetch: image slot was not part of the last run"; return RPE_ERR_INVALID; }
            const size_t img = (size_t)h->lay.in_img;
            const uint8_t *in = index < h->lay.in_na ? h->lay.in_a + (size_t)index * img : h->lay.in_b + (size_t)(index - h->lay.in_na) * img;
            HIPCHK(h, hipMemcpyAsync(h->d_pyr + (size_t)index * h->lay.stride + h->lay.lv[0].off, in, img, hipMemcpyDeviceToDevice, h->stream));
        }
        if (which == 3) { rpe_launch_blur(h, index); HIPCHK(h, hipGetLastError()); src = h->d_bufA; }
        HIPCHK(h, hipMemcpyAsync(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    uint8_t *o = h_out;
    for (int l = 0; l < RPE_NLEVELS; ++l) {
        const RpeLevel &v = h->lay.lv[l];
        for (int y = 0; y < v.h; ++y, o += v.w) memcpy(o, tmp.data() + v.off + (size_t)y * v.pitch, v.w);
    }
    return RPE_OK;
}

extern "C" int rpe_match_hamming(rpe_handle *h, const uint8_t *h_desc1, const int32_t *n1, const uint8_t *h_desc2,
                                 const int32_t *n2, int B, int32_t *qidx, int32_t *tidx, int32_t *dist, int32_t *n_matches)
{
    if (!h || !h_desc1 || !h_desc2 || !n1 || !n2 || B < 1) return RPE_ERR_INVALID;
    if (B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t per = (size_t)h->lay.kcap * 32, mm = h->cfg.max_matches;
    for (int i = 0; i < B; ++i) if (n1[i] < 0 || n2[i] < 0 || n1[i] > h->lay.kcap || n2[i] > h->lay.kcap) {
        h->err = "descriptor count exceeds keypoint capacity"; return RPE_ERR_INVALID;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_desc, h_desc1, per * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_desc + per * B, h_desc2, per * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_kp_count, n1, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_kp_count + B, n2, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    rpe_launch_match(h, B);
    HIPCHK(h, hipGetLastError());
    if (qidx) HIPCHK(h, hipMemcpyAsync(qidx, h->d_m_q, sizeof(int) * mm * B, hipMemcpyDeviceToHost, h->stream));
    if (tidx) HIPCHK(h, hipMemcpyAsync(tidx, h->d_m_t, sizeof(int) * mm * B, hipMemcpyDeviceToHost, h->stream));
    if (dist) HIPCHK(h, hipMemcpyAsync(dist, h->d_m_d, sizeof(int) * mm * B, hipMemcpyDeviceToHost, h->stream));
    if (n_matches) HIPCHK(h, hipMemcpyAsync(n_matches, h->d_m_n, sizeof(int) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RPE_OK;
}

static int upload_points(rpe_handle *h, const float *p1, const float *p2, const int32_t *m, int B)
{
    const size_t mm = h->cfg.max_matches;
    for (int i = 0; i < B; ++i) if (m[i] < 0 || m[i] > (int)mm) { h->err = "match count exceeds max_matches"; return RPE_ERR_INVALID; }
    HIPCHK(h, hipMemcpyAsync(h->d_pts1, p1, sizeof(float2) * mm * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_pts2, p2, sizeof(float2) * mm * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_m_n, m, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    return RPE_OK;
}

extern "C" int rpe_find_essential(rpe_handle *h, const float *h_pts1, const float *h_pts2, const int32_t *m, int B,
                                  const double K[9], double *E, uint8_t *mask, int32_t *found, int32_t *info)
{
    if (!h || !h_pts1 || !h_pts2 || !m || !K || B < 1) return RPE_ERR_INVALID;
    if (B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = upload_points(h, h_pts1, h_pts2, m, B);
    if (rc) return rc;
    if ((rc = set_K(h, K)) != RPE_OK) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_E, 0, sizeof(double) * 9 * B, h->stream));
    rpe_launch_ransac(h, B, true);
    HIPCHK(h, hipGetLastError());
    std::vector<RpeRansacState> st(B);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->d_rstate, sizeof(RpeRansacState) * B, hipMemcpyDeviceToHost, h->stream));
    if (mask) HIPCHK(h, hipMemcpyAsync(mask, h->d_mask, (size_t)h->cfg.max_matches * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < B; ++i) {
        if (E) memcpy(E + 9 * i, st[i].E, sizeof(double) * 9);
        if (found) found[i] = st[i].found;
        if (info) { info[4 * i] = st[i].best_count; info[4 * i + 1] = st[i].best_iter; info[4 * i + 2] = st[i].best_model; info[4 * i + 3] = st[i].iters_run; }
    }
    return RPE_OK;
}

extern "C" int rpe_recover_pose(rpe_handle *h, const double *h_E, const float *h_pts1, const float *h_pts2, const int32_t *m,
                                int B, const double K[9], double *R, double *t, int32_t *inliers)
{
    if (!h || !h_E || !h_pts1 || !h_pts2 || !m || !K || B < 1) return RPE_ERR_INVALID;
    if (B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = upload_points(h, h_pts1, h_pts2, m, B);
    if (rc) return rc;
    if ((rc = set_K(h, K)) != RPE_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_E, h_E, sizeof(double) * 9 * B, hipMemcpyHostToDevice, h->stream));
    rpe_launch_pose(h, B, false);
    HIPCHK(h, hipGetLastError());
    return rpe_fetch_results(h, B, R, t, inliers, nullptr, nullptr);
}

// ---------------------------------------------------------------- profiling
static const char *kStageNames[RPE_STAGE_COUNT] = {"pyramid", "fast", "nms", "select", "harris", "keypoints",
                                                   "angle", "blur", "describe", "match", "ransac", "pose"};
extern "C" const char *rpe_stage_name(int s) { return (s >= 0 && s < RPE_STAGE_COUNT) ? kStageNames[s] : "?"; }
extern "C" int rpe_set_profiling(rpe_handle *h, int enable)
{
    if (!h) return RPE_ERR_INVALID;
    h->profiling = enable != 0; h->ev_valid = false;
    return RPE_OK;
}
extern "C" int rpe_get_stage_ms(rpe_handle *h, float *ms)
{
    if (!h || !ms) return RPE_ERR_INVALID;
    if (!h->ev_valid) { h->err = "no profiled batch recorded"; return RPE_ERR_INVALID; }
    HIPCHK(h, hipEventSynchronize(h->ev[RPE_STAGE_COUNT]));
    for (int i = 0; i < RPE_STAGE_COUNT; ++i) HIPCHK(h, hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    return RPE_OK;
}

// ------------------------------------------------------------ roofline calibration
// The hot path is bound by vector-instruction ISSUE, not by HBM, so bench.py prices the
// dominant kernel and the matcher against a MEASURED issue rate: each kernel below runs a long stream of one
// instruction kind (inline asm: the count is exact, nothing is folded away) over independent register chains,
// at 1, 2, 4 or 8 resident waves per SIMD on every CU.  MI355X_MICROARCH.md: a wave64 VALU instruction takes
// 2 cycles on the 32-wide SIMD when >= 2 waves feed it, 4 cycles for one wave alone; f64 and transcendental
// instructions take longer.  Kinds: the instructions the ORB / matcher / RANSAC inner loops are made of.
#define CALIB_UNROLL 16
template <int KIND>
__global__ __launch_bounds__(256) void valu_calib_kernel(unsigned *sink, int iters)
{
    unsigned a[8];
    double d[8];
    float f[8];
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t f2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = threadIdx.x * 2654435761u + u * 40503u + blockIdx.x; d[u] = 1.0 + 1e-9 * (double)(a[u] & 1023u);
        f[u] = 1.0f + 1e-6f * (float)(a[u] & 1023u); f2[u].x = f[u]; f2[u].y = f[u] * 0.5f;
    }
    const float fk = 1.0000001f;
    const f2_t fk2 = {1.0000001f, 0.9999999f};
    const unsigned k0 = 0x9E3779B9u ^ threadIdx.x, k1 = 0x01010101u;
    const double dk = 1.0000000001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < CALIB_UNROLL / 8; ++r) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0)       // the Hamming inner loop: v_xor_b32 + v_bcnt_u32_b32 (2 instructions)
                    asm volatile("v_xor_b32 %0, %0, %1\n\tv_bcnt_u32_b32 %0, %0, %2" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 1)  // FAST pair test: packed 16-bit min / max (2 instructions)
                    asm volatile("v_pk_min_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 2)  // byte gather: v_perm_b32 (1 instruction)
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 3)  // packed-u8 dot product: v_dot4_u32_u8 (1 instruction)
                    asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 4)  // FAST ring score: v_min3_i32 + v_max3_i32 (2 instructions)
                    asm volatile("v_min3_i32 %0, %0, %1, %2\n\tv_max3_i32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 5)  // resize / blur taps: v_mad_u32_u24 (1 instruction)
                    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(k1), "v"(k0));
                else if (KIND == 6)  // RANSAC / pose: v_mul_f64 + v_add_f64 (2 instructions; the library compiles without contraction)
                    asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1" : "+v"(d[u]) : "v"(dk));
                else if (KIND == 7)  // v_fma_f64 (1 instruction), for reference
                    asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[u]) : "v"(dk));
                else if (KIND == 8)  // v_fma_f32 (1 instruction): the guide's 2-cycle instruction
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[u]) : "v"(fk));
                else if (KIND == 9)  // v_pk_fma_f32 (1 instruction, 2 FMAs per lane): the 157 TFLOP/s f32 vector peak
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(f2[u]) : "v"(fk2));
                else if (KIND == 10) // pyramid / descriptor taps: v_dot2_u32_u16
                    asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 11) // byte phase: v_alignbyte_b32
                    asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(k0), "v"(k1));
                else if (KIND == 12) // 32-bit multiply: v_mul_lo_u32
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(k0));
                else if (KIND == 13) // 64-bit multiply-add (what 32-bit index arithmetic often compiles to): v_mad_u64_u32
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[u]) : "v"(k0), "v"(k1) : "vcc");
                else if (KIND == 14) // SDWA operand select: v_mul_u32_u24_sdwa
                    asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[u]) : "v"(k0));
                else                 // packed 16-bit multiply-add: v_pk_mad_u16
                    asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[u]) : "v"(k0), "v"(k1));
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= a[u] ^ (unsigned)__double_as_longlong(d[u]) ^ __float_as_uint(f[u]) ^ __float_as_uint(f2[u].x) ^ __float_as_uint(f2[u].y);
    if (acc == 0x12345679u) *sink = acc;
}

static const int kCalibInstPerSlot[16] = {2, 2, 1, 1, 2, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1};
static const char *kCalibNames[16] = {"v_xor_b32+v_bcnt_u32_b32", "v_pk_min_i16+v_pk_max_i16", "v_perm_b32", "v_dot4_u32_u8",
                                      "v_min3_i32+v_max3_i32", "v_mad_u32_u24", "v_mul_f64+v_add_f64", "v_fma_f64", "v_fma_f32", "v_pk_fma_f32",
                                      "v_dot2_u32_u16", "v_alignbyte_b32", "v_mul_lo_u32", "v_mad_u64_u32", "v_mul_u32_u24_sdwa", "v_pk_mad_u16"};
extern "C" const char *rpe_calibrate_valu_name(int kind) { return (kind >= 0 && kind < 16) ? kCalibNames[kind] : "?"; }

extern "C" int rpe_calibrate_valu(rpe_handle *h, int kind, int waves_per_simd, double *wave_insts_per_s)
{
    if (!h || !wave_insts_per_s || kind < 0 || kind > 15 || waves_per_simd < 1 || waves_per_simd > 8) return RPE_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device));
    const int ncu = prop.multiProcessorCount;
    // one 256-thread block = 4 waves = one wave per SIMD of a CU; waves_per_simd blocks per CU
    const int blocks = ncu * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {                    // first repetition warms up
        HIPCHK(h, hipEventRecord(e0, h->stream));
        switch (kind) {
#define CALIB_CASE(K) case K: hipLaunchKernelGGL((valu_calib_kernel<K>), dim3(blocks), dim3(256), 0, h->stream, (unsigned *)h->d_hist, iters); break;
            CALIB_CASE(0) CALIB_CASE(1) CALIB_CASE(2) CALIB_CASE(3) CALIB_CASE(4) CALIB_CASE(5) CALIB_CASE(6) CALIB_CASE(7)
            CALIB_CASE(8) CALIB_CASE(9) CALIB_CASE(10) CALIB_CASE(11) CALIB_CASE(12) CALIB_CASE(13) CALIB_CASE(14) CALIB_CASE(15)
#undef CALIB_CASE
        }
        HIPCHK(h, hipEventRecord(e1, h->stream));
        HIPCHK(h, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    const double insts = (double)blocks * 4.0 * (double)iters * CALIB_UNROLL * kCalibInstPerSlot[kind];
    *wave_insts_per_s = insts / ((double)best * 1e-3);
    return RPE_OK;
}

// HBM streaming rate of this device: 16-B-per-lane read of the handle's pyramid buffer (>= 256 MiB so the
// Infinity Cache cannot serve it) -- the "achievable" figure next to the 8 TB/s spec peak in the bench line.
__global__ __launch_bounds__(256) void calib_read16_kernel(const uint4 *__restrict__ p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345679u) *sink = acc;
}
extern "C" int rpe_calibrate_hbm(rpe_handle *h, double *bytes_per_s)
{
    if (!h || !bytes_per_s) return RPE_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t NIo = h->cfg.feature_method == RPE_FEATURE_SIFT ? 1 : (size_t)h->n_img_cap;
    const size_t bytes = NIo * (size_t)h->lay.stride;
    if (bytes < ((size_t)256 << 20)) { h->err = "rpe_calibrate_hbm: the handle's pyramid buffer is smaller than the 256 MiB Infinity Cache"; return RPE_ERR_INVALID; }
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        HIPCHK(h, hipEventRecord(e0, h->stream));
        hipLaunchKernelGGL(calib_read16_kernel, dim3(8192), dim3(256), 0, h->stream, (const uint4 *)h->d_pyr, bytes / 16, (unsigned *)h->d_hist);
        HIPCHK(h, hipEventRecord(e1, h->stream));
        HIPCHK(h, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    *bytes_per_s = (double)bytes / ((double)best * 1e-3);
    return RPE_OK;
}


// ------------------------------------------------------------------ SIFT stage API
extern "C" int rpe_sift_detect_and_compute(rpe_handle *h, const uint8_t *h_imgs, int n_images,
                                           rpe_sift_keypoint *kps, float *desc, int32_t *counts)
{
    if (!h || !h_imgs || n_images < 1) return RPE_ERR_INVALID;
    if (h->cfg.feature_method != RPE_FEATURE_SIFT) { h->err = "handle was not created for SIFT"; return RPE_ERR_INVALID; }
    if (n_images > h->n_img_cap) { h->err = "n_images exceeds 2*max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t img = (size_t)h->cfg.width * h->cfg.height;
    const int na = n_images < h->cfg.max_batch ? n_images : h->cfg.max_batch, nb = n_images - na;
    HIPCHK(h, hipMemcpyAsync(h->d_stage1, h_imgs, img * na, hipMemcpyHostToDevice, h->stream));
    if (nb) HIPCHK(h, hipMemcpyAsync(h->d_stage2, h_imgs + img * na, img * nb, hipMemcpyHostToDevice, h->stream));
    int rc = rpe_sift_run(h, h->d_stage1, h->d_stage2, na, nb);
    if (rc) return rc;
    const int kcap = h->lay.kcap;
    std::vector<float> fin((size_t)n_images * kcap * 6);
    std::vector<int> cnt(n_images);
    std::vector<uint8_t> d8(desc ? (size_t)n_images * kcap * 128 : 0);
    if ((rc = rpe_sift_fetch(h, n_images, fin.data(), cnt.data())) != RPE_OK) return rc;
    if (desc) {
        HIPCHK(h, hipMemcpy(d8.data(), h->d_desc, d8.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < d8.size(); ++i) desc[i] = (float)d8[i];       // cv2 returns SIFT descriptors as f32
    }
    for (int i = 0; i < n_images; ++i) {
        if (counts) counts[i] = cnt[i];
        if (!kps) continue;
        for (int k = 0; k < cnt[i]; ++k) {
            const float *f = &fin[((size_t)i * kcap + k) * 6];
            rpe_sift_keypoint &o = kps[(size_t)i * kcap + k];
            int oct; memcpy(&oct, &f[5], 4);
            o.x = f[0] * 0.5f; o.y = f[1] * 0.5f; o.size = f[2] * 0.5f; o.angle = f[3]; o.response = f[4];
            o.octave = (oct & ~255) | ((oct - 1) & 255);                          // firstOctave = -1 (sift.dispatch.cpp)
        }
    }
    return RPE_OK;
}

extern "C" int64_t rpe_sift_debug_gauss(rpe_handle *h, int index, float *out)
{
    if (!h || !h->sift) return 0;
    if (out && rpe_sift_fetch_gauss(h, index, out) != RPE_OK) return -1;
    return rpe_sift_gauss_floats(h);
}

extern "C" int rpe_match_l2(rpe_handle *h, const float *h_desc1, const int32_t *n1, const float *h_desc2, const int32_t *n2, int B,
                            int32_t *qidx, int32_t *tidx, float *dist, int32_t *n_matches)
{
    if (!h || !h_desc1 || !h_desc2 || !n1 || !n2 || B < 1) return RPE_ERR_INVALID;
    if (h->cfg.norm_type != RPE_NORM_L2) { h->err = "handle was not created for NORM_L2"; return RPE_ERR_INVALID; }
    if (B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return RPE_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t dim = (size_t)h->desc_bytes;               // 128 (SIFT) or 32 (ORB descriptors under NORM_L2)
    const size_t per = (size_t)h->lay.kcap * dim, mm = h->cfg.max_matches;
    std::vector<uint8_t> u(2 * per * B, 0);
    for (int s = 0; s < 2; ++s) {
        const float *src = s ? h_desc2 : h_desc1; const int32_t *cn = s ? n2 : n1;
        for (int i = 0; i < B; ++i) {
            if (cn[i] < 0 || cn[i] > h->lay.kcap) { h->err = "descriptor count exceeds keypoint capacity"; return RPE_ERR_INVALID; }
            for (size_t e = 0; e < (size_t)cn[i] * dim; ++e) {
                float v = src[(size_t)i * per + e];
                if (!(v >= 0.f && v <= 255.f) || v != (float)(int)v) { h->err = "NORM_L2 path expects byte-valued descriptors (SIFT's integer-valued 0..255 floats, or ORB bytes)"; return RPE_ERR_INVALID; }
                u[(size_t)s * per * B + (size_t)i * per + e] = (uint8_t)v;
            }
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->d_desc, u.data(), u.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_kp_count, n1, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_kp_count + B, n2, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    rpe_launch_match_l2(h, B);
    HIPCHK(h, hipGetLastError());
    if (qidx) HIPCHK(h, hipMemcpyAsync(qidx, h->d_m_q, sizeof(int) * mm * B, hipMemcpyDeviceToHost, h->stream));
    if (tidx) HIPCHK(h, hipMemcpyAsync(tidx, h->d_m_t, sizeof(int) * mm * B, hipMemcpyDeviceToHost, h->stream));
    if (dist) HIPCHK(h, hipMemcpyAsync(dist, h->d_m_d, sizeof(float) * mm * B, hipMemcpyDeviceToHost, h->stream));
    if (n_matches) HIPCHK(h, hipMemcpyAsync(n_matches, h->d_m_n, sizeof(int) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RPE_OK;
}
