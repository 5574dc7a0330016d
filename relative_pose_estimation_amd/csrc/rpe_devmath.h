// rpe_devmath.h -- deterministic device math shared by the ORB and SIFT kernels.
// Every function has a line-for-line twin in the CPU oracle (same operations, same order,
// compiled with -ffp-contract=off), so f32/f64 results compare bit for bit.
#pragma once
#include <hip/hip_runtime.h>

// cv::fastAtan2 (core/mathfuncs_core: atan_f32), degrees in [0, 360)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// sin/cos for x in [0, 2*pi] (fdlibm kernel polynomials, f64)
__device__ __forceinline__ void det_sincos(double x, double &sn, double &cs)
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
    case 0: sn = s;  cs = c;  break;
    case 1: sn = c;  cs = -s; break;
    case 2: sn = -s; cs = -c; break;
    default: sn = -c; cs = s; break;
    }
}

// exp(x), |x| < 700: n = rint(x/ln2), degree-11 Taylor on the remainder, f64
__device__ __forceinline__ double det_exp_core(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00;
    double nf = rint(x * INV_LN2);
    double r = (x - nf * LN2_HI) - nf * LN2_LO;
    double p = 1.0 + r * (1.0 + r * (0.5 + r * (1.0 / 6 + r * (1.0 / 24 + r * (1.0 / 120 + r * (1.0 / 720 + r * (1.0 / 5040 +
               r * (1.0 / 40320 + r * (1.0 / 362880 + r * (1.0 / 3628800 + r * (1.0 / 39916800)))))))))));
    int n = (int)nf;
    return p * __longlong_as_double((long long)(1023 + n) << 52);
}
__device__ __forceinline__ float det_expf(float xf) { return xf < -87.0f ? 0.f : (float)det_exp_core((double)xf); }
__device__ __forceinline__ float det_exp2f(float t) { return (float)det_exp_core((double)t * 0.69314718055994530942); }
