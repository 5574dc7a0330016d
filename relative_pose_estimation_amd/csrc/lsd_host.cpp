// lsd_host.cpp -- line-segment detector for the VP-refinement post-step.
//
// Replaces cv2.createLineSegmentDetector(cv2.LSD_REFINE_STD).detect(gray)
// (reference src/core/pose_estimator.py:160-175).  The reference runs this step on the CPU after the
// pose has been estimated; it is not part of the GPU hot path here either: region growing is a
// sequential flood fill over a gradient-ordered seed list.  Algorithm = LSD (Grompone von Gioi,
// Jakubowicz, Morel, Randall, IPOL 2012) with OpenCV's defaults (imgproc/src/lsd.cpp: scale 0.8,
// sigma_scale 0.6, quant 2.0, ang_th 22.5, density_th 0.7, 1024 bins, REFINE_STD = density refinement,
// no NFA step), restated from the published algorithm:
//   1. Gaussian 7x7 (sigma 0.6/0.8, 8-bit fixed-point taps) + 0.8x INTER_LINEAR_EXACT down-scale
//   2. 2x2 gradient: norm, level-line angle (fastAtan2), pixels with norm <= 2/sin(22.5 deg) undefined
//   3. seeds in descending order of the gradient norm quantised to 1024 bins
//   4. region growing (8-neighbourhood, angle tolerance 22.5 deg, running mean direction)
//   5. rectangle from the norm-weighted centroid and the smallest-eigenvalue axis of the inertia matrix
//   6. density refinement: tighter angle tolerance from the seed neighbourhood, then shrinking radius
// Parity with cv2 is UNPINNED: cv2 is not installable here and the reference holds no line fixtures;
// cv2 orders equal-bin seeds with an unstable std::sort, this file uses raster order inside a bin.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rpe_amd.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kNotDef = -1024.0;
constexpr double kDegToRad = kPi / 180.0;

// cv::fastAtan2 (degrees in [0, 360)), same polynomial as rpe_devmath.h
float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / kPi);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

inline int refl101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; if (p >= n) p = 2 * n - 2 - p; } return p; }
inline int round_half_even(double v) { return (int)std::nearbyint(v); }

struct RegionPoint { int x, y; double angle, modgrad; };
struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy; };

struct Lsd {
    int w = 0, h = 0;
    std::vector<uint8_t> img, used;
    std::vector<double> angles, modgrad;

    // ---- 1. Gaussian (taps of sigma 0.75 in 8.8 fixed point: 4 56 136 56 4) and 0.8x bilinear
    void prepare(const uint8_t *gray, int W, int H)
    {
        static const int taps[7] = {0, 4, 56, 136, 56, 4, 0};
        std::vector<uint16_t> tmp((size_t)W * H);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                unsigned s = 0;
                for (int k = -3; k <= 3; ++k) s += (unsigned)taps[k + 3] * gray[(size_t)y * W + refl101(x + k, W)];
                tmp[(size_t)y * W + x] = (uint16_t)s;
            }
        std::vector<uint8_t> blur((size_t)W * H);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                unsigned s = 0;
                for (int k = -3; k <= 3; ++k) s += (unsigned)taps[k + 3] * tmp[(size_t)refl101(y + k, H) * W + x];
                blur[(size_t)y * W + x] = (uint8_t)((s + 32768u) >> 16);
            }
        w = round_half_even(W * 0.8); h = round_half_even(H * 0.8);
        const double scale = 1.0 / 0.8;
        std::vector<int> xo(w), xa(w), yo(h), ya(h);
        auto coeffs = [&](int src, int dst, std::vector<int> &o, std::vector<int> &a) {
            for (int d = 0; d < dst; ++d) {
                const double f = scale * ((double)d + 0.5) - 0.5;
                const int i = (int)std::floor(f);
                if (i >= 0 && src > 1) {
                    if (i < src - 1) { o[d] = i; a[d] = round_half_even((f - (double)i) * 256.0); }
                    else { o[d] = src - 1; a[d] = 0; }
                } else { o[d] = 0; a[d] = 0; }
            }
        };
        coeffs(W, w, xo, xa); coeffs(H, h, yo, ya);
        img.assign((size_t)w * h, 0);
        for (int y = 0; y < h; ++y) {
            const uint8_t *r0 = &blur[(size_t)yo[y] * W], *r1 = &blur[(size_t)std::min(yo[y] + 1, H - 1) * W];
            for (int x = 0; x < w; ++x) {
                const int o0 = xo[x], o1 = std::min(o0 + 1, W - 1);
                const unsigned h0 = (256 - xa[x]) * r0[o0] + xa[x] * r0[o1], h1 = (256 - xa[x]) * r1[o0] + xa[x] * r1[o1];
                img[(size_t)y * w + x] = (uint8_t)(((256 - ya[y]) * h0 + ya[y] * h1 + 32768u) >> 16);
            }
        }
    }

    // ---- 2 + 3. gradient, level-line angles, seed order
    void gradient(double threshold, int n_bins, std::vector<int> &order)
    {
        angles.assign((size_t)w * h, kNotDef); modgrad.assign((size_t)w * h, 0.0);
        double max_grad = -1;
        for (int y = 0; y < h - 1; ++y)
            for (int x = 0; x < w - 1; ++x) {
                const int DA = img[(size_t)(y + 1) * w + x + 1] - img[(size_t)y * w + x];
                const int BC = img[(size_t)y * w + x + 1] - img[(size_t)(y + 1) * w + x];
                const int gx = DA + BC, gy = DA - BC;
                const double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
                modgrad[(size_t)y * w + x] = norm;
                if (norm > threshold) {
                    angles[(size_t)y * w + x] = (double)fast_atan2_deg((float)gx, (float)-gy) * kDegToRad;
                    if (norm > max_grad) max_grad = norm;
                }
            }
        const double bin_coef = max_grad > 0 ? (double)(n_bins - 1) / max_grad : 0;
        std::vector<int> count(n_bins + 1, 0);
        const int nx = w - 1, ny = h - 1;
        std::vector<int> bin((size_t)nx * ny);
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const int b = (int)(modgrad[(size_t)y * w + x] * bin_coef);
                bin[(size_t)y * nx + x] = b; ++count[b];
            }
        // descending bins, raster order inside a bin (counting sort)
        std::vector<int> start(n_bins + 1, 0);
        int acc = 0;
        for (int b = n_bins - 1; b >= 0; --b) { start[b] = acc; acc += count[b]; }
        order.assign((size_t)nx * ny, 0);
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) order[start[bin[(size_t)y * nx + x]]++] = y * w + x;
    }

    bool aligned(int x, int y, double theta, double prec) const
    {
        const double a = angles[(size_t)y * w + x];
        if (a == kNotDef) return false;
        double n = theta - a;
        if (n < 0) n = -n;
        if (n > 1.5 * kPi) { n -= 2 * kPi; if (n < 0) n = -n; }
        return n <= prec;
    }

    // ---- 4. region growing
    void grow(int sx, int sy, std::vector<RegionPoint> &reg, double &reg_angle, double prec)
    {
        reg.clear();
        reg_angle = angles[(size_t)sy * w + sx];
        reg.push_back({sx, sy, reg_angle, modgrad[(size_t)sy * w + sx]});
        float sumdx = (float)std::cos(reg_angle), sumdy = (float)std::sin(reg_angle);
        used[(size_t)sy * w + sx] = 1;
        for (size_t i = 0; i < reg.size(); ++i) {
            const int px = reg[i].x, py = reg[i].y;
            const int x0 = std::max(px - 1, 0), x1 = std::min(px + 1, w - 1), y0 = std::max(py - 1, 0), y1 = std::min(py + 1, h - 1);
            for (int yy = y0; yy <= y1; ++yy)
                for (int xx = x0; xx <= x1; ++xx) {
                    uint8_t &u = used[(size_t)yy * w + xx];
                    if (u != 1 && aligned(xx, yy, reg_angle, prec)) {
                        const double a = angles[(size_t)yy * w + xx];
                        u = 1;
                        reg.push_back({xx, yy, a, modgrad[(size_t)yy * w + xx]});
                        sumdx += std::cos((float)a); sumdy += std::sin((float)a);
                        reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * kDegToRad;
                    }
                }
        }
    }

    static double angle_diff_signed(double a, double b)
    {
        a -= b;
        while (a <= -kPi) a += 2 * kPi;
        while (a > kPi) a -= 2 * kPi;
        return a;
    }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }

    // ---- 5. rectangle
    bool to_rect(const std::vector<RegionPoint> &reg, double reg_angle, double prec, Rect &rc) const
    {
        double x = 0, y = 0, sum = 0;
        for (const RegionPoint &p : reg) { x += p.x * p.modgrad; y += p.y * p.modgrad; sum += p.modgrad; }
        if (!(sum > 0)) return false;
        x /= sum; y /= sum;
        double Ixx = 0, Iyy = 0, Ixy = 0;
        for (const RegionPoint &p : reg) {
            const double rx = p.x - x, ry = p.y - y;
            Ixx += ry * ry * p.modgrad; Iyy += rx * rx * p.modgrad; Ixy -= rx * ry * p.modgrad;
        }
        if (Ixx == 0 && Iyy == 0 && Ixy == 0) return false;
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = std::fabs(Ixx) > std::fabs(Iyy) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                                        : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
        theta *= kDegToRad;
        if (std::fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += kPi;
        const double dx = std::cos(theta), dy = std::sin(theta);
        double lmin = 0, lmax = 0, wmin = 0, wmax = 0;
        for (const RegionPoint &p : reg) {
            const double rx = p.x - x, ry = p.y - y;
            const double l = rx * dx + ry * dy, wv = -rx * dy + ry * dx;
            lmin = std::min(lmin, l); lmax = std::max(lmax, l); wmin = std::min(wmin, wv); wmax = std::max(wmax, wv);
        }
        rc = {x + lmin * dx, y + lmin * dy, x + lmax * dx, y + lmax * dy, std::max(wmax - wmin, 1.0), x, y, theta, dx, dy};
        return true;
    }

    double density(const std::vector<RegionPoint> &reg, const Rect &rc) const
    {
        return (double)reg.size() / (dist(rc.x1, rc.y1, rc.x2, rc.y2) * rc.width);
    }

    // ---- 6. density refinement (LSD_REFINE_STD)
    bool refine(std::vector<RegionPoint> &reg, double &reg_angle, double prec, Rect &rc, double density_th)
    {
        if (density(reg, rc) >= density_th) return true;
        const double xc = reg[0].x, yc = reg[0].y, ang_c = reg[0].angle;
        double sum = 0, s_sum = 0; int n = 0;
        for (const RegionPoint &p : reg) {
            used[(size_t)p.y * w + p.x] = 0;
            if (dist(xc, yc, p.x, p.y) < rc.width) { const double d = angle_diff_signed(p.angle, ang_c); sum += d; s_sum += d * d; ++n; }
        }
        if (n == 0) return false;
        const double mean = sum / n;
        const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean * sum) / n + mean * mean);
        grow(reg[0].x, reg[0].y, reg, reg_angle, tau);
        if (reg.size() < 2) return false;
        if (!to_rect(reg, reg_angle, prec, rc)) return false;
        double den = density(reg, rc);
        if (den >= density_th) return true;
        // shrink the region radius to 75 % until it is dense enough
        const double x0 = reg[0].x, y0 = reg[0].y;
        const double r1 = (x0 - rc.x1) * (x0 - rc.x1) + (y0 - rc.y1) * (y0 - rc.y1), r2 = (x0 - rc.x2) * (x0 - rc.x2) + (y0 - rc.y2) * (y0 - rc.y2);
        double rad2 = std::max(r1, r2);
        while (den < density_th) {
            rad2 *= 0.75 * 0.75;
            for (size_t i = 0; i < reg.size(); ++i) {
                const double dx = reg[i].x - x0, dy = reg[i].y - y0;
                if (dx * dx + dy * dy > rad2) {
                    used[(size_t)reg[i].y * w + reg[i].x] = 0;
                    std::swap(reg[i], reg.back());
                    reg.pop_back();
                    --i;
                }
            }
            if (reg.size() < 2) return false;
            if (!to_rect(reg, reg_angle, prec, rc)) return false;
            den = density(reg, rc);
        }
        return true;
    }

    int detect(const uint8_t *gray, int W, int H, float *lines, int cap)
    {
        const double ang_th = 22.5, quant = 2.0, density_th = 0.7;
        const int n_bins = 1024;
        const double prec = kPi * ang_th / 180.0, p = ang_th / 180.0, rho = quant / std::sin(prec);
        prepare(gray, W, H);
        std::vector<int> order;
        gradient(rho, n_bins, order);
        const double log_nt = 5.0 * (std::log10((double)w) + std::log10((double)h)) / 2.0 + std::log10(11.0);
        const size_t min_reg = (size_t)(-log_nt / std::log10(p));
        used.assign((size_t)w * h, 0);
        std::vector<RegionPoint> reg;
        int n = 0;
        for (int idx : order) {
            if (used[idx] || angles[idx] == kNotDef) continue;
            double reg_angle;
            grow(idx % w, idx / w, reg, reg_angle, prec);
            if (reg.size() < min_reg) continue;
            Rect rc;
            if (!to_rect(reg, reg_angle, prec, rc)) continue;
            if (!refine(reg, reg_angle, prec, rc, density_th)) continue;
            if (n < cap) {
                lines[4 * n + 0] = (float)((rc.x1 + 0.5) / 0.8); lines[4 * n + 1] = (float)((rc.y1 + 0.5) / 0.8);
                lines[4 * n + 2] = (float)((rc.x2 + 0.5) / 0.8); lines[4 * n + 3] = (float)((rc.y2 + 0.5) / 0.8);
            }
            ++n;
        }
        return n;
    }
};

}  // namespace

extern "C" int rpe_lsd_detect(const uint8_t *h_gray, int width, int height, float *h_lines, int capacity, int32_t *n_lines)
{
    if (!h_gray || !h_lines || !n_lines || width < 8 || height < 8 || capacity < 0) return RPE_ERR_INVALID;
    Lsd d;
    *n_lines = d.detect(h_gray, width, height, h_lines, capacity);
    return RPE_OK;
}
