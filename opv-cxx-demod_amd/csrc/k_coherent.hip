// k_coherent.hip — the batch coherent demodulator (`opv-demod -c`): Costas loop + AFC on a fixed 40-sample symbol
// grid, one wavefront per stream, lane i = sample i of the symbol.
//
// Replaces CoherentMSKDemodulator::demodulate (reference src/opv-demod.cpp:455-543; batch driver :1144-1161,
// SURVEY.md §8f row 4). Per symbol: every sample is de-rotated by the carrier phase (advancing by loop_freq per
// SAMPLE, :484) and correlated with the two tone oscillators (:470-481); soft = Re c2 - Re c1 (:502-507); the
// dominant tone's correlation drives a second-order loop with the phase error Im / |.| (:512-530) and the AFC with
// arg(dom conj(prev)) (:535-543).
//
// PARITY STATUS: prefix only, by the nature of the reference. Its loop does not lock (on the reference's own clean
// loopback the AFC runs to the +/-2000 Hz clamp and 4 garbage frames come out of 10) and its trajectory is chaotic:
// 1e-15 rad on the initial carrier phase grows to 3e-13 after 2000 symbols and to O(1) after ~12 000
// (tests/test_oracle_golden.py::test_coherent_loop_is_chaotic). This kernel forms the per-sample phases in closed
// form (phase + i * increment instead of i sequential additions) and uses the device's sincos / atan2 / hypot, i.e.
// it differs from the reference in the last place of those - and therefore follows the reference's soft symbols to
// <= 1e-9 for the first ~2000 symbols and departs from them, like ANY implementation that is not the reference's own
// libm, within a few frames (tests/test_gpu_parity.py::test_coherent_prefix_parity). It exists so that `-c` runs;
// it is not tuned (a stream is serial, ~300 issued instructions per symbol).
#include <hip/hip_runtime.h>
#include <math.h>

#include "opv_device.h"

namespace {
constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kFs = 2168000.0;
constexpr double kFdev = 13550.0;
constexpr double kSymRate = kFs / 40.0;

__device__ inline double wave_sum_all(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline double wrap_pi(double p) {  // ref :488-493
    // (the reference's two loops never end for an infinite phase and take 1e5+ turns per symbol beyond this bound - a `-p` of
    // 1e9 Hz does that; a wave that never finishes hangs the GPU, so such a phase is folded in one step instead)
    if (!(fabs(p) < 1.0e6)) return p - kTwoPi * rint(p / kTwoPi);
    while (p > kPi) p -= kTwoPi;
    while (p < -kPi) p += kTwoPi;
    return p;
}
__device__ inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }
}  // namespace

extern "C" __global__ __launch_bounds__(64) void k_coherent_frontend(OpvStream* __restrict__ streams, double pll_alpha,
                                                                      double pll_beta) {
    OpvStream& st = streams[blockIdx.x];
    const int lane = threadIdx.x;
    if (!st.eof || st.tail_done) return;                  // batch: the whole capture, once (ref :1132-1135)
    const uint64_t n = st.n_avail;
    const uint64_t nsym = n / OPV_SPS;                    // ref :462
    if (nsym > st.cap_soft) { if (lane == 0) st.overflow = 1; return; }
    double fo = st.freq_offset;                           // set_freq_offset(estimate) (ref :1148-1149)
    double ph1 = 0.0, ph2 = 0.0, carrier = 0.0, loop_freq = 0.0, prev_re = 0.0, prev_im = 0.0;
    double inc1 = kTwoPi * (-kFdev + fo) / kFs, inc2 = kTwoPi * (+kFdev + fo) / kFs;   // ref :459-460
    const int* iq = reinterpret_cast<const int*>(st.iq);
    const double li = (double)lane;
    for (uint64_t sym = 0; sym < nsym; ++sym) {
        double c1r = 0, c1i = 0, c2r = 0, c2i = 0;
        if (lane < OPV_SPS) {
            const int w = iq[sym * OPV_SPS + lane];
            const double sr = (double)(int)(short)(w & 0xFFFF), si = (double)(w >> 16);
            double sn, cs;
            sincos(carrier + li * loop_freq, &sn, &cs);   // phase_rot = (cos, -sin) (ref :470)
            const double xr = sr * cs + si * sn, xi = si * cs - sr * sn;
            double s1, k1, s2, k2;
            sincos(ph1 + li * inc1, &s1, &k1);
            sincos(ph2 + li * inc2, &s2, &k2);
            c1r = xr * k1 + xi * s1; c1i = xi * k1 - xr * s1;          // corrected * conj(lo) (ref :477-478)
            c2r = xr * k2 + xi * s2; c2i = xi * k2 - xr * s2;
        }
        c1r = wave_sum_all(c1r); c1i = wave_sum_all(c1i);
        c2r = wave_sum_all(c2r); c2i = wave_sum_all(c2i);
        ph1 = wrap_pi(ph1 + 40.0 * inc1);
        ph2 = wrap_pi(ph2 + 40.0 * inc2);
        carrier = wrap_pi(carrier + 40.0 * loop_freq);
        const double en1 = c1r * c1r + c1i * c1i, en2 = c2r * c2r + c2i * c2i;   // ref :496-497
        if (lane == 0) st.soft[sym & (st.cap_soft - 1)] = c2r - c1r;            // ref :502-507
        const double dr = en1 > en2 ? c1r : c2r, di = en1 > en2 ? c1i : c2i;     // ref :512
        const double mag = hypot(dr, di);                                        // ref :515
        const double pe = mag > 1e-10 ? di / mag : 0.0;                          // ref :517-522
        loop_freq += pll_beta * pe;                                              // ref :526-527
        carrier += pll_alpha * pe;
        loop_freq = clampd(loop_freq, -0.1, 0.1);                                // ref :530
        if (sym > 0) {                                                           // ref :535-543
            const double zr = dr * prev_re + di * prev_im, zi = di * prev_re - dr * prev_im;
            fo = clampd(fo + st.afc_alpha * (atan2(zi, zr) * kSymRate / kTwoPi), -2000.0, 2000.0);
            inc1 = kTwoPi * (-kFdev + fo) / kFs;
            inc2 = kTwoPi * (+kFdev + fo) / kFs;
        }
        prev_re = dr; prev_im = di;                                              // ref :545
    }
    if (lane == 0) {
        double* c = st.chunk_log + 5 * (size_t)(st.n_chunks % st.cap_chunks);
        c[0] = fo; c[1] = loop_freq; c[2] = carrier; c[3] = (double)(n - nsym * OPV_SPS); c[4] = (double)nsym;
        st.freq_offset = fo;
        st.n_soft = nsym; st.total_samples = n; st.origin = n;
        st.n_chunks += 1; st.tail_done = 1;
    }
}
