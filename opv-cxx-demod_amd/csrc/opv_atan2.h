// opv_atan2.h — fp64 atan2 for the AFC phase detector (reference src/opv-demod.cpp:299, std::arg), written for a
// WAVE-UNIFORM argument and without the octant fix-up: atan(|y| / |x|) = pi/4 + atan(q), q = (|y| - |x|) / (|y| + |x|) in [-1, 1],
// from a table of short polynomials picked by round(q x rows) - a fraction of the instructions of the generic libm routine
// (19-term polynomial + fix-ups). Two tables ship: 257 rows of degree 5 (k_frontend_x4.hip / k_frontend_x16.hip, max abs
// error vs glibc < 5e-16) and 1025 cubics written in the argument itself (k_frontend.hip's one-wave kernels, <= 3e-14 rad:
// below says why that is plenty). The routines here are the HOST statements of what the kernels evaluate inline on LDS /
// constant-memory copies of the tables (tests/test_atan2_host.py builds this header for the host, against glibc).
// Not handled here (callers do): x == y == 0 and non-finite inputs.
// Elsewhere: the 1025-row table in (k, h) form that the shipped one was derived from lives with the host tests
// (tests/atan/opv_atan2_q3.h).
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define OPV_HD __host__ __device__
#else
#define OPV_HD
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

#endif  // !__HIP_DEVICE_COMPILE__

// ---- the same angle from a finer grid and a cubic written in the argument itself (k_frontend.hip's row-broadcast body) ----
// 1025 rows (k/512, |q - k/512| <= 1/1024) of a cubic in q: three FMAs and two 16-byte LDS reads instead of five and three, and
// neither k as a double nor h = q - k/512 is formed. Approximation error <= 3e-14 rad (tests/test_atan2_host.py asserts 1e-13
// against glibc) - deliberately not the 4e-16 of the table above: the AFC loop turns an angle error e into a steady-state
// frequency error of ~8600 e Hz (3e-10 Hz), and a soft symbol moves by ~2e-8 of its size per Hz, i.e. by 1e-17: nothing the
// 1e-5 contract, the 1e-9 the tests assert or a quantiser boundary can see. Row k is row k of the (k, h) form of the same fit
// (tests/atan/opv_atan2_q3.h) re-expanded around 0: the same values up to Horner's roundings (~3e-16; the tests compare the
// two). The axis is not exact in this form (opv_atan2_q3r(0, x > 0) ~ 1e-16): 1e-15 Hz per symbol on a frequency state
// compared to 1e-7.
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
