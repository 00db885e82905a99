// opv_atan2.h — fp64 atan2 for the AFC phase detector (reference src/opv-demod.cpp:299,
// std::arg), written for a WAVE-UNIFORM argument: one divide, a 33-row table of degree-8
// Taylor coefficients around k/32 picked by round(32 r), eight FMAs and the octant fix-up —
// about 1/3 of the instructions of the generic libm routine (19-term polynomial + fix-ups).
// Accuracy vs glibc atan2 over 4e6 random arguments: max abs < 5e-16 (1 ulp of pi), max
// relative < 4e-16 (tests/test_atan2_host.py, host build of this same header).
// Not handled here (callers do): x == y == 0 and non-finite inputs.
//
// The device kernel (k_frontend.hip) restates these steps inline on an LDS copy of the table.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define OPV_HD __host__ __device__
#else
#define OPV_HD
#endif
// Device images carry only what a kernel of that build reads: the 33-row table belongs to the comparison mappings
// (-DOPV_WITH_COMPARISON_MAPPINGS), the 257-row and the first 1025-row table are host references (tests/test_atan2_host.py;
// k_frontend_x4.hip keeps its own image of the 257-row one); the product's one-wave kernels read kOpvAtanTabQ3R.
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

// ---- the same angle without the octant fix-up (k_frontend.hip's row-broadcast body, k_frontend_x4.hip) -----------------
// atan(|y| / |x|) = pi/4 + atan(q), q = (|y| - |x|) / (|y| + |x|) in [-1, 1]: 257 rows (k/128, |h| <= 1/256) of a degree-5
// polynomial f(0) + h g(h), g a degree-4 near-minimax fit (tools/gen_atan_table.py): five FMAs, no compare / select for
// the octant. ABSOLUTE accuracy like opv_atan2 (max abs error vs glibc < 5e-16); the relative accuracy of tiny angles is
// that of an angle near pi/4 (q near -1 cancels against the table's constant term) - the AFC integrates the angle, so the
// absolute error counts. An argument on the positive x axis gives exactly 0 (row 0 starts with an exact 0 and h = 0).
#ifndef __HIP_DEVICE_COMPILE__
static const double kOpvAtanTabQ[257][6] = {
#include "opv_atan_table_q.inc"
};

OPV_HD inline double opv_atan2_q(double y, double x) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double q = (ay - ax) / (ay + ax);              // in [-1, 1]
    const double kd = __builtin_rint(q * 128.0);         // nearest expansion point k/128
    const double h = __builtin_fma(kd, -1.0 / 128.0, q); // |h| <= 1/256, exact
    const double* t = kOpvAtanTabQ[(int)kd + 128];
    double p = t[5];
    p = __builtin_fma(p, h, t[4]);
    p = __builtin_fma(p, h, t[3]);
    p = __builtin_fma(p, h, t[2]);
    p = __builtin_fma(p, h, t[1]);
    p = __builtin_fma(p, h, t[0]);
    if (x < 0) p = 3.14159265358979323846 - p;
    return y < 0 ? -p : p;
}

// ---- the same angle from a finer grid and a cubic (k_frontend.hip's row-broadcast body from round 3 on) -----------------------
// 1025 rows (k/512, |h| <= 1/1024) of f(0) + h g(h), g of degree 2: three FMAs and two 16-byte LDS reads instead of five and
// three. Approximation error <= 3e-14 rad (tests/test_atan2_host.py asserts 1e-13 against glibc) - deliberately not the 4e-16 of
// the tables above: the AFC loop turns an angle error e into a steady-state frequency error of ~8600 e Hz (3e-10 Hz), and a soft
// symbol moves by ~2e-8 of its size per Hz, i.e. by 1e-17: nothing the 1e-5 contract, the 1e-9 the tests assert or a quantiser
// boundary can see. An argument on the positive x axis still gives exactly 0.
static const double kOpvAtanTabQ3[1025][4] = {
#include "opv_atan_table_q3.inc"
};

OPV_HD inline double opv_atan2_q3(double y, double x) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double q = (ay - ax) / (ay + ax);              // in [-1, 1]
    const double kd = __builtin_rint(q * 512.0);         // nearest expansion point k/512
    const double h = __builtin_fma(kd, -1.0 / 512.0, q); // |h| <= 1/1024, exact
    const double* t = kOpvAtanTabQ3[(int)kd + 512];
    double p = t[3];
    p = __builtin_fma(p, h, t[2]);
    p = __builtin_fma(p, h, t[1]);
    p = __builtin_fma(p, h, t[0]);
    if (x < 0) p = 3.14159265358979323846 - p;
    return y < 0 ? -p : p;
}
#endif  // !__HIP_DEVICE_COMPILE__

// ---- and with the cubics written in the argument itself (what k_frontend.hip's row-broadcast body evaluates) --------------
// Row k of kOpvAtanTabQ3R is row k of kOpvAtanTabQ3 re-expanded around 0: the same function values up to Horner's roundings
// (~3e-16), without forming k as a double and h = q - k/512: two instructions per symbol. The axis is no longer exact
// (opv_atan2_q3r(0, x > 0) ~ 1e-16): 1e-15 Hz per symbol on a frequency state compared to 1e-7.
#ifdef __HIP_DEVICE_COMPILE__
__constant__
#else
static const
#endif
double kOpvAtanTabQ3R[1025][4] = {
#include "opv_atan_table_q3r.inc"
};

OPV_HD inline double opv_atan2_q3r(double y, double x) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double q = (ay - ax) / (ay + ax);              // in [-1, 1]
    const double* t = kOpvAtanTabQ3R[(int)__builtin_rint(q * 512.0) + 512];
    double p = __builtin_fma(t[3], q, t[2]);
    p = __builtin_fma(p, q, t[1]);
    p = __builtin_fma(p, q, t[0]);
    if (x < 0) p = 3.14159265358979323846 - p;
    return y < 0 ? -p : p;
}
