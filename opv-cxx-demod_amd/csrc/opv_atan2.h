// opv_atan2.h — fp64 atan2 for the AFC phase detector (reference src/opv-demod.cpp:299,
// std::arg), written for a WAVE-UNIFORM argument: one divide, a 32-interval table of degree-9
// Taylor coefficients picked by a scalar index (on the GPU the coefficients arrive through the
// scalar cache into SGPRs and feed v_fma_f64 directly), nine FMAs and the quadrant fix-up —
// about 1/3 of the instructions of the generic libm routine (19-term polynomial + fix-ups).
// Accuracy vs glibc atan2 over 2e7 random arguments: max abs 4.4e-16 (1 ulp of pi), max
// relative 2.9e-16 (tests/test_atan2_host.py, host build of this same header).
// Not handled here (callers do): x == y == 0 and non-finite inputs.
//
// Hooks (define before including to specialise for the device):
//   OPV_ATAN_DIV(n, d)      n / d            (default: IEEE divide)
//   OPV_ATAN_FMAC(p, h, c)  p * h + c, c a table coefficient (default: fma)
//   OPV_ATAN_UNI(k)         make the interval index wave-uniform (default: identity)
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define OPV_HD __host__ __device__
#else
#define OPV_HD
#endif
#ifndef OPV_ATAN_DIV
#define OPV_ATAN_DIV(n, d) ((n) / (d))
#endif
#ifndef OPV_ATAN_FMAC
#define OPV_ATAN_FMAC(p, h, c) __builtin_fma((p), (h), (c))
#endif
#ifndef OPV_ATAN_UNI
#define OPV_ATAN_UNI(k) (k)
#endif

#ifdef __HIP_DEVICE_COMPILE__
__constant__
#else
static const
#endif
double kOpvAtanTab[32][10] = {
#include "opv_atan_table.inc"
};

OPV_HD inline double opv_atan2(double y, double x) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
    const double r = OPV_ATAN_DIV(mn, mx);              // in [0, 1]
    int k = (int)(r * 32.0);
    k = k > 31 ? 31 : k;
    k = OPV_ATAN_UNI(k);
    const double h = r - (k ? ((double)k + 0.5) * (1.0 / 32.0) : 0.0);  // interval 0 is expanded at 0
    const double* t = kOpvAtanTab[k];
    double p = t[9];
    p = OPV_ATAN_FMAC(p, h, t[8]);
    p = OPV_ATAN_FMAC(p, h, t[7]);
    p = OPV_ATAN_FMAC(p, h, t[6]);
    p = OPV_ATAN_FMAC(p, h, t[5]);
    p = OPV_ATAN_FMAC(p, h, t[4]);
    p = OPV_ATAN_FMAC(p, h, t[3]);
    p = OPV_ATAN_FMAC(p, h, t[2]);
    p = OPV_ATAN_FMAC(p, h, t[1]);
    p = OPV_ATAN_FMAC(p, h, t[0]);
    if (ay > ax) p = 1.57079632679489661923 - p;
    if (x < 0) p = 3.14159265358979323846 - p;
    return y < 0 ? -p : p;
}
