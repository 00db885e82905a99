// tests/atan/opv_atan2_q3.h — TEST INFRASTRUCTURE: the 1025-row angle table in its (k, h) form, f(0) + h g(h) around k/512, that
// the shipped table (csrc/opv_atan_table_q3r.inc: the same cubics re-expanded in the argument) was derived from
// (opv-cxx-demod_amd/tools/gen_atan_table.py writes both). tests/test_atan2_host.py checks this form against glibc and the
// shipped form against this one. An argument on the positive x axis gives exactly 0 here.
#pragma once
#include "opv_atan2.h"

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
