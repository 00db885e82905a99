// k_offset_search.hip — coarse carrier-offset search, one 256-thread workgroup per stream.
//
// Replaces MSKDemodulatorAFC::estimate_offset (reference src/opv-demod.cpp:131-202):
// 121 coarse candidates (-1500..+1500 step 25 Hz, :135) then 13 fine ones (best-30..best+30
// step 5, :169), energy = sum over the first <=1000 fixed 40-sample symbol windows of
// |sum s conj(lo1)|^2 + |sum s conj(lo2)|^2 (:143-158); strict '>' so the first maximum wins
// (:161, :195).
//
// MI355X mapping. The reference accumulates the LO phase over all 40 000 samples; a
// symbol's energy does not depend on the phase at the start of its window (|.|^2 removes a
// common rotation), so each candidate needs only a 40-entry phasor table per tone,
// exp(j i inc). The table is built in LDS by 80 lanes (fp64 sincos), then the 256 threads
// each own symbols t, t+256, ... and run the two 40-tap complex correlations from L2-resident
// int16 IQ (160 KB per stream, 16-byte loads). fp64 throughout: neighbouring candidates
// differ by ~3e-7 (coarse) / ~1e-8 (fine) relative in energy, far above the
// ~1e-13 re-association error of the block reduction but below fp32 resolution.
//
// Roofline: compute-trivial (134 x 1000 x 80 cMAC = 43 MFMA-free fp64 FMAs x4 per stream);
// runs once per stream. Algorithmic bytes: 160 000 B read per stream.
#include <hip/hip_runtime.h>
#include <math.h>

#include "opv_device.h"

namespace {

constexpr double kTwoPi = 2.0 * 3.14159265358979323846;  // ref :43-44
constexpr double kFs = 2168000.0;                        // ref :40
constexpr double kFdev = 13550.0;                        // ref :42

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace

extern "C" __global__ __launch_bounds__(256) void k_offset_search(OpvStream* __restrict__ streams,
                                                                   OpvGlobalCfg cfg) {
    OpvStream& st = streams[blockIdx.x];
    if (st.first_chunk_done) return;

    uint64_t n;
    bool run;
    if (cfg.streaming) {
        // main() runs the search on the first FULL chunk only (ref :1026-1037); a capture that
        // never fills one chunk is demodulated without it (ref :1088-1090).
        if (st.n_avail < OPV_CHUNK) return;
        n = OPV_CHUNK;
        run = !cfg.have_init_offset;  // ref :1031
    } else {
        if (!st.eof) return;  // batch mode slurps all of stdin first (ref :1132-1135)
        n = st.n_avail;
        run = true;  // ref :1166 (batch ignores -o)
    }
    const int tid = threadIdx.x;
    if (!run) {
        if (tid == 0) st.first_chunk_done = 1;
        return;
    }

    const uint64_t test = n < (uint64_t)OPV_SPS * 1000u ? n : (uint64_t)OPV_SPS * 1000u;  // ref :141
    const int nsym = (int)(test / OPV_SPS);

    __shared__ double2 tab[2][OPV_SPS];  // exp(j i inc_t), t = tone
    __shared__ double part[4];
    __shared__ double s_best_e, s_best, s_fine;

    if (tid == 0) { s_best_e = 0.0; s_best = 0.0; s_fine = 0.0; }
    const int4* iq4 = reinterpret_cast<const int4*>(st.iq);

    for (int c = 0; c < 134; ++c) {
        __syncthreads();
        double offset;
        if (c < 121) offset = -1500.0 + 25.0 * c;           // exact in fp64, as the += 25 loop
        else offset = (s_best - 30.0) + 5.0 * (c - 121);     // ref :169
        if (tid < 2 * OPV_SPS) {
            const int tone = tid / OPV_SPS, i = tid % OPV_SPS;
            const double inc = kTwoPi * ((tone ? kFdev : -kFdev) + offset) / kFs;  // ref :137-138
            double sn, cs;
            sincos((double)i * inc, &sn, &cs);
            tab[tone][i] = make_double2(cs, sn);
        }
        __syncthreads();

        double acc = 0.0;
        for (int sym = tid; sym < nsym; sym += 256) {
            double a1r = 0, a1i = 0, a2r = 0, a2i = 0;
            const int4* p = iq4 + (size_t)sym * (OPV_SPS / 4);
#pragma unroll
            for (int q = 0; q < OPV_SPS / 4; ++q) {
                const int4 v = p[q];
                const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double xr = (double)(int)(short)(w[k] & 0xFFFF);
                    const double xi = (double)(w[k] >> 16);
                    const double2 t1 = tab[0][4 * q + k], t2 = tab[1][4 * q + k];
                    // x * conj(lo)  (ref :151-152)
                    a1r = fma(xr, t1.x, fma(xi, t1.y, a1r));
                    a1i = fma(xi, t1.x, fma(-xr, t1.y, a1i));
                    a2r = fma(xr, t2.x, fma(xi, t2.y, a2r));
                    a2i = fma(xi, t2.x, fma(-xr, t2.y, a2i));
                }
            }
            acc += (a1r * a1r + a1i * a1i) + (a2r * a2r + a2i * a2i);  // ref :158
        }
        acc = wave_sum(acc);
        if ((tid & 63) == 0) part[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            const double e = (part[0] + part[1]) + (part[2] + part[3]);
            st.energies[c] = e;
            if (e > s_best_e) {  // strict: first maximum wins (ref :161, :195)
                s_best_e = e;
                if (c < 121) s_best = offset; else s_fine = offset;
            }
            if (c == 120) s_fine = s_best;  // ref :168
        }
    }
    __syncthreads();
    if (tid == 0) {
        st.est_offset = s_fine;
        st.freq_offset = s_fine;  // demod.set_freq_offset(est) (ref :1033 / :1167)
        st.first_chunk_done = 1;
    }
}
