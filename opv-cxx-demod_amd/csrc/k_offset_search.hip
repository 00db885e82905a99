// k_offset_search.hip — coarse carrier-offset search, one 256-thread workgroup per stream, ONE pass over
// the samples for all 134 candidates.
//
// Replaces MSKDemodulatorAFC::estimate_offset (reference src/opv-demod.cpp:131-202):
// 121 coarse candidates (-1500..+1500 step 25 Hz, :135) then 13 fine ones (best-30..best+30
// step 5, :169), energy = sum over the first <=1000 fixed 40-sample symbol windows of
// |sum s conj(lo1)|^2 + |sum s conj(lo2)|^2 (:143-158); strict '>' so the first maximum wins
// (:161, :195).
//
// The reference evaluates 134 x 2 correlations per window. All of them are ONE function of the
// candidate: with theta = 2 pi o / Fs and u_i = i - 19.5 (window centre),
//     c_t(o) = sum_i x_i conj(lo_t[i])  =  (unit phasor) * sum_i y_t,i exp(-j u_i theta),
//     y_t,i = x_i exp(-/+ j 2 pi i / 160)          (13 550 * 40 = Fs / 4: a fixed 40-entry table),
// |u theta| <= 0.0865 over the whole +/-1530 Hz span, so the exponential is its Taylor series to
// machine precision after ten terms (the eleventh is 6e-18):
//     c_t(o) ~ sum_{k<10} (-j theta)^k m_t,k ,    m_t,k = sum_i y_t,i u_i^k / k!   (ten complex moments),
// and the window's energy |c_t|^2 is a polynomial of degree 18 in theta whose coefficients are sums of
// products m_k conj(m_l). Summed over windows and tones that is 19 real numbers per stream - after which
// every candidate, coarse or fine, costs one Horner evaluation. Work per stream: 40 000 samples x 40 FMA
// + 1000 windows x 220 FMA = 1.8 M FMA instead of 134 x 40 000 x 8 = 43 M, and the fine pass needs no
// second look at the samples. The moments' weights (cos / sin table x u^k / k!) are wave-uniform: they
// arrive through scalar loads and are scalar operands of the FMAs.
//
// Exactness. The energies agree with the reference's to ~1e-13 relative (its own phase accumulation
// over 40 000 samples is no better); candidates in the tested captures are 1e-8 ... 3e-7 apart. Where the
// winner is NOT clear - another candidate within kTieRel = 1e-11 relative - the contenders are re-evaluated the
// reference's way (`exact_energy`: LO phases accumulated sample by sample from zero, products and sums in
// the reference's order, windows summed in order; only libm's sin / cos are the device's), which shrinks
// the undecidable band from 1e-13 to the last-place differences of sin / cos. The count of such
// re-evaluations is reported (opv_stream_state.offset_ties).
// That last band is closed on the HOST: a stream on which two RE-EVALUATED energies are still within kHostRel = 2e-13 of each
// other (exact ties included: a real-valued capture) puts its index on `tie_list` and leaves its 19 polynomial coefficients in
// OpvStream.est_poly. Behind the search, in stream order and without a host wait (opv_capi.hip: opv_process), k_tie_collect
// copies what the decision needs (coefficients, window count, the first <= 40 000 samples) into pinned host memory, a host
// function (hipLaunchHostFunc) repeats the decision with the contenders evaluated by opv_offset_candidate_energy
// (opv_offset_host.cpp) - the reference's loop on the reference's own libm - and k_tie_apply carries the estimate, the energies
// tap and the tie count into the stream's context before the front-end reads them. Re-evaluated contenders further apart than
// that are decided here: sin / cos cannot move them past each other (bound at kHostRel). Why not every guarded stream: a
// candidate costs the host 160 000 sin / cos, and a context of 32 768 streams has a dozen guarded ones. The device's own
// decision stands throughout when the host's libm does not reproduce the pinned energy (tie_list == nullptr then).
//
// Both bands are RELATIVE TO WHAT THE ERRORS SCALE WITH, which is not the energy alone: a correlation c = sum x conj(lo) over a
// window carries an absolute error ~ eps sum |x|, so the energy E = sum |c|^2 carries ~ 2 eps sqrt(E) sqrt(40 P), P = sum |x|^2
// over the samples used (Cauchy-Schwarz over taps and windows). For a signal the tones match, E ~ 40 P and that IS eps E; for
// an input whose correlation is weak against its power (an out-of-band tone, an interferer, noise on a DC offset) it is
// larger by sqrt(40 P / E). `near` below therefore accepts a gap d when d <= rel x E or d^2 <= rel^2 x 40 P E - products and
// one subtraction, no square root, so that the host's restatement (opv_offset_host.cpp) reaches the same verdicts bit for bit.
// P is an exact integer in fp64 (<= 80 000 x 2^31 < 2^53), whatever the order of summation.
//
// Roofline: 160 000 B read per stream, once. Compute: see above; no MFMA (the contraction is 40 x 40 per
// window with weights that differ per tap - a GEMM only in name, and fp64).
#include <hip/hip_runtime.h>
#include <math.h>

#include "opv_device.h"

namespace {

constexpr double kTwoPi = 2.0 * 3.14159265358979323846;  // ref :43-44
constexpr double kFs = 2168000.0;                        // ref :40
constexpr double kFdev = 13550.0;                        // ref :42
constexpr int kK = OPV_OFFS_TERMS;                       // Taylor terms (moments) per tone
constexpr double kTieRel = 1e-11;                        // "not clearly below the best": 100x the agreement with the reference
// Second level: after the contenders have been re-evaluated in the reference's order of operations, their energies differ from
// the reference's only through the last places of sin / cos - a relative 1e-14 at worst (each LO sample within ~2 ulp, 40
// products per window, the windows' errors added up without cancellation credit). Contenders that are still within kHostRel of
// the best re-evaluated energy - 20x that bound - go to the host, whose libm IS the reference's; all others are decided here.
constexpr double kHostRel = 2e-13;

// is `e` within `rel` of `top` (>= e) on the scale the errors of both have? (see the header: power-scaled band)
__device__ inline bool near(double top, double e, double rel, double power) {
    const double d = top - e;
    return d <= rel * top || d * d <= (rel * rel * 40.0) * power * top;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// The reference's own evaluation of one candidate (ref :143-159, restated like the oracle's
// candidate_energy): phases from zero, one addition per sample, never wrapped.
// Workgroup-wide; `ph0` (2 x 1000 doubles) and `esym` (1000 doubles) are LDS scratch.
__device__ double exact_energy(const int16_t* __restrict__ iq, int nsym, double offset, double* ph0, double* esym) {
    const int tid = threadIdx.x;
    const double i1 = kTwoPi * (-kFdev + offset) / kFs;   // ref :137
    const double i2 = kTwoPi * (+kFdev + offset) / kFs;   // ref :138
    __syncthreads();
    if (tid == 0 || tid == 64) {                          // one lane per tone (two waves): 40 sequential adds per window
        const double inc = tid ? i2 : i1;
        double* out = ph0 + (tid ? 1000 : 0);
        double p = 0.0;
        for (int s = 0; s < nsym; ++s) {
            out[s] = p;
#pragma unroll 8
            for (int i = 0; i < OPV_SPS; ++i) p += inc;
        }
    }
    __syncthreads();
    for (int s = tid; s < nsym; s += 256) {
        double p1 = ph0[s], p2 = ph0[1000 + s];
        double a1r = 0, a1i = 0, a2r = 0, a2i = 0;
        const int16_t* x = iq + (size_t)s * (2 * OPV_SPS);
        for (int i = 0; i < OPV_SPS; ++i) {
            const double xr = (double)x[2 * i], xi = (double)x[2 * i + 1];
            double s1, c1, s2, c2;
            sincos(p1, &s1, &c1);
            sincos(p2, &s2, &c2);
            a1r += xr * c1 + xi * s1;                     // x conj(lo), contraction off (ref :151-152)
            a1i += xi * c1 - xr * s1;
            a2r += xr * c2 + xi * s2;
            a2i += xi * c2 - xr * s2;
            p1 += i1;
            p2 += i2;
        }
        esym[s] = (a1r * a1r + a1i * a1i) + (a2r * a2r + a2i * a2i);   // ref :158
    }
    __syncthreads();
    double total = 0.0;
    for (int s = 0; s < nsym; ++s) total += esym[s];      // every thread, same order as the reference's loop
    return total;
}

}  // namespace

// wtab: [40 taps][2 (cos, sin)][kK] doubles = cos(pi i/80) u^k/k!, sin(pi i/80) u^k/k!, u = i - 19.5 (host-made, opv_create)
// tie_list (device, may be null): [0] = number of streams that re-evaluated a candidate in this launch, [1 + i] = their indices
extern "C" __global__ __launch_bounds__(256) void k_offset_search(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                   const double* __restrict__ wtab, uint32_t* __restrict__ tie_list) {
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

    __shared__ double s_red[4][2 * kK - 1];
    __shared__ double s_poly[2 * kK - 1];
    __shared__ double s_e[134];
    __shared__ double s_scratch[3000];       // exact_energy: 2 x 1000 window-start phases + 1000 window energies
    __shared__ double s_best_e, s_best, s_fine, s_power;
    __shared__ double s_pw[4];
    __shared__ int s_ties, s_host;
    __shared__ unsigned char s_play[134];

    // ---- one pass: moments per window, products accumulated into the 19 polynomial coefficients ----------
    double ed[kK], eo[2 * kK - 1];           // sum |m_k|^2 (-> theta^2k) and sum over k > l of Re(rho m_k conj m_l) (-> theta^(k+l))
#pragma unroll
    for (int k = 0; k < kK; ++k) ed[k] = 0.0;
#pragma unroll
    for (int p = 0; p < 2 * kK - 1; ++p) eo[p] = 0.0;
    double pw = 0.0;                         // sum |x|^2 over this thread's windows (exact: integers below 2^53)
    const int4* iq4 = reinterpret_cast<const int4*>(st.iq);
    for (int sym = tid; sym < nsym; sym += 256) {
        // A = sum xr cos w_k, B = sum xi cos w_k, C = sum xi sin w_k, D = sum xr sin w_k
        double A[kK], B[kK], C[kK], D[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k) A[k] = B[k] = C[k] = D[k] = 0.0;
        const int4* p = iq4 + (size_t)sym * (OPV_SPS / 4);
        for (int q = 0; q < OPV_SPS / 4; ++q) {
            const int4 v = p[q];
            const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double xr = (double)(int)(short)(w[j] & 0xFFFF);
                const double xi = (double)(w[j] >> 16);
                pw = fma(xr, xr, fma(xi, xi, pw));
                const double* wt = wtab + (size_t)(4 * q + j) * (2 * kK);   // wave-uniform address: scalar loads
#pragma unroll
                for (int k = 0; k < kK; ++k) {
                    const double wc = wt[k], ws = wt[kK + k];
                    A[k] = fma(xr, wc, A[k]);
                    B[k] = fma(xi, wc, B[k]);
                    C[k] = fma(xi, ws, C[k]);
                    D[k] = fma(xr, ws, D[k]);
                }
            }
        }
        // tone 1 (-13550 + o): y = x exp(+j pi i/80) -> m = (A - C, D + B); tone 2: y = x exp(-j pi i/80) -> m = (A + C, B - D)
#pragma unroll
        for (int tone = 0; tone < 2; ++tone) {
            double mr[kK], mi[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                mr[k] = tone ? A[k] + C[k] : A[k] - C[k];
                mi[k] = tone ? B[k] - D[k] : D[k] + B[k];
            }
            // |sum_k (-j theta)^k m_k|^2 = sum_k theta^2k |m_k|^2 + 2 sum_{k>l} theta^(k+l) Re((-j)^(k-l) m_k conj m_l)
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                ed[k] = fma(mr[k], mr[k], fma(mi[k], mi[k], ed[k]));
#pragma unroll
                for (int l = 0; l < k; ++l) {
                    const int r = (k - l) & 3;            // (-j)^r = 1, -j, -1, j
                    if (r == 0) eo[k + l] = fma(mr[k], mr[l], fma(mi[k], mi[l], eo[k + l]));          // + Re g
                    else if (r == 1) eo[k + l] = fma(mi[k], mr[l], fma(-mr[k], mi[l], eo[k + l]));    // + Im g
                    else if (r == 2) eo[k + l] = fma(-mr[k], mr[l], fma(-mi[k], mi[l], eo[k + l]));   // - Re g
                    else eo[k + l] = fma(-mi[k], mr[l], fma(mr[k], mi[l], eo[k + l]));                // - Im g
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 2 * kK - 1; ++p) {
        double v = 2.0 * eo[p];
        if ((p & 1) == 0) v += ed[p / 2];
        v = wave_sum(v);
        if ((tid & 63) == 0) s_red[tid >> 6][p] = v;
    }
    pw = wave_sum(pw);
    if ((tid & 63) == 0) s_pw[tid >> 6] = pw;
    if (tid == 0) { s_best_e = 0.0; s_best = 0.0; s_fine = 0.0; s_ties = 0; s_host = 0; }
    __syncthreads();
    if (tid < 2 * kK - 1) st.est_poly[tid] = s_poly[tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
    if (tid == 0) st.est_power = s_power = (s_pw[0] + s_pw[1]) + (s_pw[2] + s_pw[3]);
    __syncthreads();
    const double power = s_power;
    auto poly_energy = [&](double offset) {
        const double th = kTwoPi * offset / kFs;
        double e = s_poly[2 * kK - 2];
#pragma unroll
        for (int p = 2 * kK - 3; p >= 0; --p) e = fma(e, th, s_poly[p]);
        return e;
    };
    // Decide among candidates [c0, c1) the way the reference's loop does (strict '>' against the running best),
    // after re-evaluating in its own order every candidate that is within kTieRel (1e-11, power-scaled: `near`) of the
    // best energy in play (the running best included) unless the winner is clear.
    auto decide = [&](int c0, int c1, double base_offset, double step, bool fine) {
        if (tid >= c0 && tid < c1) s_e[tid] = poly_energy(base_offset + step * (double)(tid - c0));
        __syncthreads();
        double top = s_best_e;
        for (int c = c0; c < c1; ++c) top = fmax(top, s_e[c]);
        if (top > 0.0) {
            // (the fine candidate AT the coarse winner's offset repeats its evaluation: identical by construction,
            // here as in the reference, and never '>' - it is no contender)
            auto in_play = [&](int c) { return near(top, s_e[c], kTieRel, power) && !(fine && base_offset + step * (double)(c - c0) == s_best); };
            const bool defend = fine && near(top, s_best_e, kTieRel, power);   // the coarse winner defends its energy
            int contenders = defend ? 1 : 0;
            for (int c = c0; c < c1; ++c) contenders += in_play(c);
            if (contenders > 1) {                         // workgroup-uniform: every thread sees the same LDS values
                for (int c = c0; c < c1; ++c) {
                    const bool play = in_play(c);
                    if (tid == 0) s_play[c] = play ? 1 : 0;
                    if (!play) continue;
                    const double e = exact_energy(st.iq, nsym, base_offset + step * (double)(c - c0), s_scratch, s_scratch + 2000);
                    __syncthreads();
                    if (tid == 0) { s_e[c] = e; ++s_ties; }
                    __syncthreads();
                }
                if (defend) {
                    const double e = exact_energy(st.iq, nsym, s_best, s_scratch, s_scratch + 2000);
                    __syncthreads();
                    if (tid == 0) { s_best_e = e; ++s_ties; }
                }
                __syncthreads();
                if (tid == 0) {                           // second level: are two re-evaluated energies within what sin / cos could move?
                    double m = defend ? s_best_e : 0.0;
                    for (int c = c0; c < c1; ++c)
                        if (s_play[c]) m = fmax(m, s_e[c]);
                    int close = (defend && near(m, s_best_e, kHostRel, power)) ? 1 : 0;
                    for (int c = c0; c < c1; ++c) close += (s_play[c] && near(m, s_e[c], kHostRel, power)) ? 1 : 0;
                    if (close > 1) s_host = 1;
                }
                __syncthreads();
            }
        }
        if (tid == 0) {
            for (int c = c0; c < c1; ++c) {
                st.energies[c] = s_e[c];
                // the fine candidate at the coarse winner's own offset: a bit-identical repeat in the reference, never
                // '>' there - here its polynomial value may not be compared with a re-evaluated (exact) best
                if (fine && base_offset + step * (double)(c - c0) == s_best) continue;
                if (s_e[c] > s_best_e) {                  // strict: first maximum wins (ref :161, :195)
                    s_best_e = s_e[c];
                    if (fine) s_fine = base_offset + step * (double)(c - c0);
                    else s_best = base_offset + step * (double)(c - c0);
                }
            }
            if (!fine) s_fine = s_best;                   // ref :168
        }
        __syncthreads();
    };
    decide(0, 121, -1500.0, 25.0, false);                 // exact in fp64, as the reference's += 25 loop (:135)
    decide(121, 134, s_best - 30.0, 5.0, true);           // ref :169
    if (tid == 0) {
        st.est_offset = s_fine;
        st.freq_offset = s_fine;  // demod.set_freq_offset(est) (ref :1033 / :1167)
        st.est_ties = (uint32_t)s_ties;
        st.est_nsym = (uint32_t)nsym;
        st.first_chunk_done = 1;
        if (s_host && tie_list) tie_list[1 + atomicAdd(&tie_list[0], 1u)] = blockIdx.x;   // (capacity: one entry per stream)
    }
}

// ---- the host's tie decision, in stream order (see the header; opv_capi.hip enqueues collect -> host function -> apply) -------
// Pass `pass` serves entries [pass * slots, (pass + 1) * slots) of the tie list: one workgroup per slot copies the stream's
// polynomial, power, window count and first nsym x 40 samples into pinned host memory (16 B per lane over PCIe). The last pass
// of a round also reports how many listed streams lie beyond it (their device decision stands; counted by the host function).
extern "C" __global__ __launch_bounds__(256) void k_tie_collect(const OpvStream* __restrict__ streams, const uint32_t* __restrict__ tie_list,
                                                                 uint32_t pass, uint32_t slots, uint32_t last_pass, OpvTieStage* __restrict__ stage) {
    const uint32_t listed = tie_list[0], first = pass * slots;
    const uint32_t n = listed > first ? (listed - first < slots ? listed - first : slots) : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stage->n = n; stage->listed = listed; stage->beyond = (last_pass && listed > first + n) ? listed - first - n : 0u; }
    if (blockIdx.x >= n) return;
    const uint32_t s = tie_list[1 + first + blockIdx.x];
    const OpvStream& st = streams[s];
    OpvTieSlot& sl = stage->slot[blockIdx.x];
    const uint32_t nsym = st.est_nsym <= 1000u ? st.est_nsym : 1000u;
    if (threadIdx.x == 0) { sl.stream = s; sl.nsym = nsym; sl.power = st.est_power; sl.ties = 0; sl.est = st.est_offset; }
    if (threadIdx.x < 2 * kK - 1) sl.poly[threadIdx.x] = st.est_poly[threadIdx.x];
    const int4* src = reinterpret_cast<const int4*>(st.iq);
    int4* dst = reinterpret_cast<int4*>(sl.iq);
    for (uint32_t i = threadIdx.x; i < nsym * (OPV_SPS / 4); i += 256) dst[i] = src[i];
}
// What the host function left in the slots: estimate, tie count and the energies tap of each decided stream (ties == 0: the
// host found nothing to re-evaluate - the device's decision stands).
extern "C" __global__ __launch_bounds__(192) void k_tie_apply(OpvStream* __restrict__ streams, const OpvTieStage* __restrict__ stage, uint32_t slots, int n_streams) {
    const uint32_t n = stage->n < slots ? stage->n : slots;
    if (blockIdx.x >= n) return;
    const OpvTieSlot& sl = stage->slot[blockIdx.x];
    if (sl.ties == 0 || sl.stream >= (uint32_t)n_streams) return;
    OpvStream& st = streams[sl.stream];
    if (threadIdx.x < 134) st.energies[threadIdx.x] = sl.energies[threadIdx.x];
    if (threadIdx.x == 0) {
        st.freq_offset = st.est_offset = sl.est;      // demod.set_freq_offset(est) (ref :1033 / :1167)
        st.est_ties = sl.ties;
    }
}
