// Floating-point steps of the frontend written so that the device and the host
// produce identical bits: every operation rounds on its own (no fused
// multiply-add), the transcendental functions are explicit polynomials.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

namespace sa {

#if defined(__HIP_DEVICE_COMPILE__)
#define SA_FMUL(a, b) __fmul_rn((a), (b))
#define SA_FADD(a, b) __fadd_rn((a), (b))
#define SA_FSUB(a, b) __fsub_rn((a), (b))
#define SA_FDIV(a, b) __fdiv_rn((a), (b))
#define SA_DMUL(a, b) __dmul_rn((a), (b))
#define SA_DADD(a, b) __dadd_rn((a), (b))
#define SA_DSUB(a, b) __dsub_rn((a), (b))
#define SA_RINT_I(a) __float2int_rn(a)
#else
// The host build passes -ffp-contract=off, so plain operators round separately.
#define SA_FMUL(a, b) ((a) * (b))
#define SA_FADD(a, b) ((a) + (b))
#define SA_FSUB(a, b) ((a) - (b))
#define SA_FDIV(a, b) ((a) / (b))
#define SA_DMUL(a, b) ((a) * (b))
#define SA_DADD(a, b) ((a) + (b))
#define SA_DSUB(a, b) ((a) - (b))
#define SA_RINT_I(a) ((int)lrintf(a))
#endif

// atan2 in degrees by the degree-7 odd polynomial OpenCV's fastAtan2 uses
// (the angle ORB stores in cv::KeyPoint::angle), range [0, 360].
__host__ __device__ inline float fast_atan2_deg(float y, float x) {
    // Coefficients as products of two floats, folded at compile time in IEEE float.
    const float deg = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * deg;
    const float p3 = -0.3258083974640975f * deg;
    const float p5 = 0.1555786518463281f * deg;
    const float p7 = -0.04432655554792128f * deg;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = SA_FDIV(ay, SA_FADD(ax, eps));
        c2 = SA_FMUL(c, c);
        a = SA_FMUL(SA_FADD(SA_FMUL(SA_FADD(SA_FMUL(SA_FADD(SA_FMUL(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = SA_FDIV(ax, SA_FADD(ay, eps));
        c2 = SA_FMUL(c, c);
        a = SA_FSUB(90.f, SA_FMUL(SA_FADD(SA_FMUL(SA_FADD(SA_FMUL(SA_FADD(SA_FMUL(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = SA_FSUB(180.f, a);
    if (y < 0) a = SA_FSUB(360.f, a);
    return a;
}

// cos and sin of a float angle in [0, 7] radians, evaluated in double and
// rounded to float: quadrant reduction with a two-part pi/2, then the classic
// minimax kernels for |r| <= pi/4.  The double result is within 1 ulp of the
// true value, so the float rounding agrees with a correctly rounded libm
// except where the true value lies within ~1e-16 (relative) of a float
// rounding boundary.
__host__ __device__ inline void sincos_f32(float angle, float* c_out, float* s_out) {
    const double x = (double)angle;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
    const double n = rint(SA_DMUL(x, two_over_pi));
    const int q = (int)n;
    double r = SA_DSUB(x, SA_DMUL(n, pio2_hi));
    r = SA_DSUB(r, SA_DMUL(n, pio2_lo));
    const double z = SA_DMUL(r, r);
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = SA_DADD(S5, SA_DMUL(z, S6));
    ps = SA_DADD(S4, SA_DMUL(z, ps));
    ps = SA_DADD(S3, SA_DMUL(z, ps));
    ps = SA_DADD(S2, SA_DMUL(z, ps));
    ps = SA_DADD(S1, SA_DMUL(z, ps));
    const double sn = SA_DADD(r, SA_DMUL(SA_DMUL(r, z), ps));
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = SA_DADD(C5, SA_DMUL(z, C6));
    pc = SA_DADD(C4, SA_DMUL(z, pc));
    pc = SA_DADD(C3, SA_DMUL(z, pc));
    pc = SA_DADD(C2, SA_DMUL(z, pc));
    pc = SA_DADD(C1, SA_DMUL(z, pc));
    const double hz = SA_DMUL(0.5, z);
    const double w = SA_DSUB(1.0, hz);
    // 1 - z/2 + z*z*pc, with the rounding error of (1 - z/2) compensated
    const double cs = SA_DADD(w, SA_DADD(SA_DSUB(SA_DSUB(1.0, w), hz), SA_DMUL(SA_DMUL(z, z), pc)));
    double c, s;
    switch (q & 3) {
        case 0: c = cs; s = sn; break;
        case 1: c = -sn; s = cs; break;
        case 2: c = -cs; s = -sn; break;
        default: c = sn; s = -cs; break;
    }
    *c_out = (float)c;
    *s_out = (float)s;
}

}  // namespace sa
