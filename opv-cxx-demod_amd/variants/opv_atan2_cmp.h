// variants/opv_atan2_cmp.h — COMPARISON BUILD ONLY (make variants, -DOPV_WITH_COMPARISON_MAPPINGS): the angle routine of the
// round-1 front-end body (variants/k_frontend_cmp_symbol.inc): one divide, a 33-row table of degree-8 Taylor coefficients
// around k/32 picked by round(32 r), eight FMAs and the octant fix-up. Accuracy vs glibc atan2 over 4e6 random arguments: max
// abs < 5e-16 (1 ulp of pi), max relative < 4e-16 (tests/test_atan2_host.py, host build of this same header).
#pragma once
#include "../csrc/opv_atan2.h"

#if !defined(__HIP_DEVICE_COMPILE__) || defined(OPV_WITH_COMPARISON_MAPPINGS)
#ifdef __HIP_DEVICE_COMPILE__
__constant__
#else
static const
#endif
double kOpvAtanTab[33][10] = {
#include "opv_atan_table.inc"
};

OPV_HD inline double opv_atan2(double y, double x) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
    const double r = mn / mx;                            // in [0, 1]
    const double kd = __builtin_rint(r * 32.0);          // nearest expansion point k/32
    const double h = __builtin_fma(kd, -1.0 / 32.0, r);  // |h| <= 1/64, exact
    const double* t = kOpvAtanTab[(int)kd];
    double p = t[8];
    p = __builtin_fma(p, h, t[7]);
    p = __builtin_fma(p, h, t[6]);
    p = __builtin_fma(p, h, t[5]);
    p = __builtin_fma(p, h, t[4]);
    p = __builtin_fma(p, h, t[3]);
    p = __builtin_fma(p, h, t[2]);
    p = __builtin_fma(p, h, t[1]);
    p = __builtin_fma(p, h, t[0]);
    if (ay > ax) p = 1.57079632679489661923 - p;
    if (x < 0) p = 3.14159265358979323846 - p;
    return y < 0 ? -p : p;
}
#endif

