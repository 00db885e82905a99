// k_frontend.hip — MSK front-end: dual-tone correlator bank + early-late-gate timing loop +
// AFC, one wavefront (64 lanes) per IQ stream, serial over symbols, parallel inside a symbol.
//
// Replaces MSKDemodulatorAFC::demodulate (+interp) (reference src/opv-demod.cpp:206-329,
// :122-128) and the streaming chunker / batch driver of main() (:1012-1113 / :1132-1173).
//
// -------------------------------------------------------------------------------------------
// Algorithm restatement (what is computed per symbol, ref line in brackets)
//   samples:   L(p) = s[floor p](1-f) + s[floor p + 1] f                      [:122-128]
//   on-time:   c_t = sum_{i<40} L(pos+i)      conj(lo_t[i])                   [:236,:243-244]
//   early:     e_t = sum_{i<40} L(pos+i-10)   conj(lo_t[i])  (s[0] if <0)     [:237,:245-246]
//   late:      l_t = sum_{i<40} L(pos+i+10)   conj(lo_t[i])                   [:238,:247-248]
//   lo_t[i] = exp(j(ph_t + i inc_t)), inc_t = 2pi(-/+13550 + fo)/Fs, ph_t += 40 inc_t [:240-251]
//   soft = |c_2|^2 - |c_1|^2                                                  [:264-268]
//   ted on the dominant tone, 2nd-order loop on pos                           [:271-286,:313]
//   AFC: fo += alpha * arg(c_dom conj(c_dom_prev)) * 54200/2pi, not on the first
//        symbol of a demodulate() call                                        [:289-306]
//
// MI355X mapping
//   * Lane j (0..59) owns ONE interpolated sample  Lam_j = L(pos + j - 10). The three gates
//     are the same 60 values under three shifts: early uses lanes 0..39, on-time 10..49, late
//     20..59, so the 120 interpolations of the reference become 60, one per lane. (Row-broadcast body: 15 samples per
//     row of 16 lanes, lane 16 r + n <-> sample 15 r + n; a row's last lane carries no weight.)
//   * The LO is factored  lo_t[i] = E_t * T_t[i] * X[i],  E_t = exp(j ph_t) (a rotation that
//     every use below is invariant to, so it is never formed and no phase is tracked),
//     T_t[i] = exp(-/+ j 2pi i/160) a per-lane CONSTANT (13550*40 = Fs/4), and
//     X[m] = exp(j m d), d = 2pi fo/Fs, |m d| <= 0.29 rad: a short polynomial pair per lane
//     instead of two libm sincos per sample. T_2 = conj(T_1), so both tones share four real
//     products per gate:  P1=sum Zr a, P2=sum Zi b, P3=sum Zi a, P4=sum Zr b with
//     Z = Lam conj(X):  C_1 = (P1+P2, P3-P4), C_2 = (P1-P2, P3+P4).
//   * |.|^2 of early/late is invariant to the gate's constant rotation; for the AFC the
//     previous on-time correlation is advanced by the LO rotation of one symbol,
//     P_t = S_t * (-/+ j) * X[40]  (T_t[40] = -/+ j exactly), so that
//     arg(c(k) conj(c(k-1))) = arg(S(k) conj(P(k-1))).
//   * Sums over the wave, current body (`symbol_r`, kernels k_msk_frontend_rb / _rb_wg4): the DP-ALU DPP form
//     v_fmac_f64_dpp acc, src0 row_newbcast:n, src1 lets a row of 16 lanes form 16 differently weighted sums of its
//     samples; with 15 samples per row, 2 x 15 FMACs give ALL window sums of the symbol (on-time / early / late
//     correlations of both tones + the on-time sums P1..P4) as row partials in lane t = lane & 15, one all-reduce over
//     the four rows completes them, v_mov_b64_dpp row_newbcast hands out what the loop filters need, the energies and
//     the dominant-tone select are lane-parallel. 153 VALU instructions per symbol.
//     The scalar loop filters run redundantly on all lanes (wave-uniform, no divergence).
//   * A LONE wave on a SIMD issues one instruction of ANY kind (VALU, SALU, LDS, s_nop, branch)
//     every ~4.7 cycles and gains nothing from independent chains (scripts/microbench): the
//     symbol rate is set by the instruction COUNT of the loop body. Hence: no per-symbol
//     bookkeeping (chunk end and tile events are tested once per batch of symbols that provably
//     need neither), stateless power-of-two ring addressing, sign choices folded into FMA
//     multipliers and bit-field inserts, one shared reciprocal for the two divides, symbols
//     processed in pairs with alternating "previous" registers, the first symbol of a call (no
//     AFC) as its own instantiation, hazard slots of swaps / DPP reads filled with useful work.
//   * int16 IQ is staged HBM -> LDS in 2048-sample tiles (8 KiB) with direct-to-LDS 16-byte
//     loads (global_load_lds_dwordx4, 1 KiB per wave instruction), two tile slots forming a
//     4096-sample ring (slot = sample index & 4095) plus a 4-sample guard that mirrors the head
//     of the even tile for a lane's second interpolation tap; the next tile is requested one
//     whole tile (~51 symbols) before its first use. Lanes read their taps straight from the
//     int16 ring (ds_read2_b32) and widen in registers.
//   * fp64 everywhere: the 1e-5 soft contract does not need it, bit-exact quantiser/sync
//     decisions on noisy input do. No MFMA: the per-symbol contraction is
//     3x4x60 with a serial dependence between symbols.
//
// Roofline: HBM-bound on paper (4 B/sample in, 8 B/symbol out => 4.2 B/sample) but actually
// issue-bound by the per-symbol feedback recurrence; see DESIGN.md and profiles/.
#include <hip/hip_runtime.h>
#include <math.h>

#include <type_traits>

#include "opv_device.h"

namespace {

constexpr double kPi = 3.14159265358979323846;  // ref :43
constexpr double kTwoPi = 2.0 * kPi;            // ref :44
constexpr double kFs = 2168000.0;               // ref :40
constexpr double kSymRate = 2168000.0 / 40.0;   // ref :41
constexpr double kDeltaPerHz = kTwoPi / kFs;    // d = 2 pi fo / Fs (ref :210-211, :305-306)

constexpr uint32_t kTile = OPV_TILE_SAMPLES;    // 2048 samples
constexpr uint32_t kRing = 2 * kTile;           // 4096 samples, two tile slots, slot = index & 4095
constexpr uint32_t kRingBytes = kRing * 4;      // 16384
constexpr uint32_t kGuardBytes = 16;            // mirror of the even slot's first 16 B
constexpr uint32_t kBack = 11;                  // lowest tap is floor(pos) - 10, one spare
constexpr uint32_t kAhead = 56;                 // highest tap is floor(pos) + 54, one spare
static_assert((kRing & (kRing - 1)) == 0, "ring must be a power of two");

// LDS map (bytes): WPB x (ring | guard), then ONE atan table shared by the workgroup's waves:
//   1025 rows x 4 doubles (opv_atan_table_q3r.inc: pi/4 + atan(q) around q = k/512, a cubic in q itself, 4e-14 rad:
//   opv_atan2.h says why that is plenty) = 32 800 B -> 49 200 B for one wave, 98 400 B for four
constexpr uint32_t kTabOff = kRingBytes + kGuardBytes;   // 16400
static_assert(kTabOff % 16 == 0, "16-byte LDS alignment");

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) unsigned char gbyte;

__device__ inline int dlo(double v) { return __double2loint(v); }
__device__ inline int dhi(double v) { return __double2hiint(v); }
__device__ inline double mkd(int hi, int lo) { return __hiloint2double(hi, lo); }

// A-values end in lanes 0..31, B-values in lanes 32..63: returns A[l]+A[l+32] | B[l-32]+B[l]
__device__ inline double swap32_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)dlo(a), (unsigned)dlo(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)dhi(a), (unsigned)dhi(b), false, false);
    return mkd((int)hi[0], (int)lo[0]) + mkd((int)hi[1], (int)lo[1]);
}
// rows of 16: even rows get A[l]+A[l+16], odd rows B[l-16]+B[l]
__device__ inline double swap16_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)dlo(a), (unsigned)dlo(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)dhi(a), (unsigned)dhi(b), false, false);
    return mkd((int)hi[0], (int)lo[0]) + mkd((int)hi[1], (int)lo[1]);
}
// v_mov_b64_dpp row_newbcast:N - lane N of every row of 16 to all lanes of that row (the one DPP control 64-bit operations
// take). Through the builtin, so that hipcc sees the DPP hazards and schedules around them; every lane is written, `old`
// only names the register: pass a dead value (a zero or an undefined one would cost a move or an s_nop).
template <int N>
__device__ inline double row_bcast(double old, double v) { return __builtin_amdgcn_update_dpp(old, v, 0x150 + N, 0xF, 0xF, false); }
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline double readlane_d(double v, int l) {
    return mkd(__builtin_amdgcn_readlane(dhi(v), l), __builtin_amdgcn_readlane(dlo(v), l));
}
// wave-uniform floating compare -> scalar branch (the operands are identical in every lane)
__device__ inline bool uni_lt(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 4 /*FCMP_OLT*/) != 0ull; }
__device__ inline bool uni_eq(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 1 /*FCMP_OEQ*/) != 0ull; }

__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// X = exp(j x), x = kfs * fo, |x| <= 0.284 (fo within the AFC clamp of +/-2000 Hz, |kf| <= 49):
//   sin x = x + x u q(u),  cos x = 1 + u r(u),  u = x^2,
// q a near-minimax cubic, r of degree 4 on u <= 0.0823 (mpmath chebyfit; in fp64 arithmetic sin to 2.2e-16 = one ulp over
// the whole range, cos to 1.3e-18 before rounding). Round 2 carried a degree-4 q as well (1e-19): one FMA per symbol for
// digits below the rounding of the products that follow. A cubic r (8e-15 at |x| = 0.287, lane 63's sample with fo at its
// clamp) was measured too - 9 cycles per symbol faster still - and is NOT used: on the 6 dB / -2 kHz fixture it moved the
// integer printed by one `raw=%.0f` tracker line (tests/test_gpu_parity.py::test_noisy_configs_vs_reference_fixtures).
// One asm block: the constants stay in registers as written and hipcc's hazard recogniser does
// not pad between the dependent FMAs.
// instantiations of the per-symbol body (see `symbol` in the kernel)
struct TagFirst { static constexpr bool first = true, wide = true; };     // first symbol of a demodulate() call
struct TagSecond { static constexpr bool first = false, wide = true; };   // second symbol under an out-of-range -o
struct TagSteady { static constexpr bool first = false, wide = false; };  // everything else

struct PrevSums {
    double a, b, c, d;  // on-time P1..P4
    double x40c, x40s;  // X[40] = exp(j 40 d) of that symbol
};
struct SinCosK {
    double s0, s1, s2, s3;      // q(u) low -> high (cubic)
    double c0, c1, c2, c3, c4;  // r(u) low -> high (degree 4)
};
__device__ inline void expj_small(double kfs, double fo, const SinCosK& k, double& xs, double& xc) {
    double x, u, p, r, t;
    asm("v_mul_f64 %[x], %[kfs], %[fo]\n\t"
        "v_mul_f64 %[u], %[x], %[x]\n\t"
        "v_fma_f64 %[r], %[c4], %[u], %[c3]\n\t"
        "v_fma_f64 %[p], %[s3], %[u], %[s2]\n\t"
        "v_fma_f64 %[r], %[r], %[u], %[c2]\n\t"
        "v_fma_f64 %[p], %[p], %[u], %[s1]\n\t"
        "v_fma_f64 %[r], %[r], %[u], %[c1]\n\t"
        "v_fma_f64 %[p], %[p], %[u], %[s0]\n\t"
        "v_fma_f64 %[r], %[r], %[u], %[c0]\n\t"
        "v_mul_f64 %[t], %[x], %[u]\n\t"
        "v_fma_f64 %[xc], %[r], %[u], 1.0\n\t"
        "v_fma_f64 %[xs], %[t], %[p], %[x]"
        : [x] "=&v"(x), [u] "=&v"(u), [p] "=&v"(p), [r] "=&v"(r), [t] "=&v"(t), [xs] "=&v"(xs), [xc] "=&v"(xc)
        : [kfs] "v"(kfs), [fo] "v"(fo), [s0] "v"(k.s0), [s1] "v"(k.s1), [s2] "v"(k.s2), [s3] "v"(k.s3),
          [c0] "v"(k.c0), [c1] "v"(k.c1), [c2] "v"(k.c2), [c3] "v"(k.c3), [c4] "v"(k.c4));
}

// Digital silence on either side of the phase detector (rare, wave-uniform, kept out of line).
// The reference's product dom * conj(prev) (ref :299) is then an exact zero whose SIGNS decide
// std::arg: atan2(+0,-0) = pi, everything else +/-0 (IEEE). Working the signs through its
// complex multiply:
//   dom == (+0,+0), prev != 0 : pi iff Re(prev) < 0 and Im(prev) < 0
//   prev == (+0,+0), dom != 0 : pi iff Re(dom)  < 0 and Im(dom)  < 0
//   both zero                  : 0
// where dom/prev are the reference's correlations, i.e. ours times the absolute LO phasor it
// carries: c_t(k) = S_t(k) conj(E_t(k)), prev_t = P_t conj(E_t(k)), P_t = S_t(k-1) (-/+ j) X40(k-1),
// E_t(k) = exp(j(-/+ k pi/2 + (80 pi/Fs) sum_{j<k} fo_j)), rebuilt here from the running sum of fo
// (fo_sum: over the symbols BEFORE this one; ksym: their number). Checked on 598 gap edges by
// tests/test_gpu_parity.py::test_many_silence_gaps_signed_zero_rule.
//
// The one input class that is NOT reproducible is counted here (`ties`, reported as
// opv_stream_state.edge_ties): a window with exactly ONE non-zero tap, i.e. the first symbol a burst
// touches or the last one it leaves. Both tone energies are then |s|^2 in exact arithmetic (P1 P2 ==
// P3 P4, soft = 4 (P3 P4 - P1 P2) = 0) and the reference's e1 > e2 (ref :272/:291) is decided by the
// rounding of its own cos^2 + sin^2 at the accumulated LO phase, which this kernel does not carry. Such a
// symbol always has digital silence on one side, so it passes through this routine either as `cur`
// (leading edge: prev is zero) or as `prv` (trailing edge: dom is zero) - no cost on the symbol path.
__device__ inline bool tone_tie(double p1, double p2, double p3, double p4) {
    const double x = p1 * p2, y = p3 * p4;
    return (p1 != 0.0 || p2 != 0.0 || p3 != 0.0 || p4 != 0.0) && fabs(y - x) <= 1e-12 * (fabs(x) + fabs(y));
}
// Returns {pd, 1.0 if such a tie was seen else 0.0} (by value: no stack slot on the caller's side).
__device__ __noinline__ double2 silence_pd(double dr, double di, PrevSums prv, bool dom1, double fo_sum, uint64_t ksym,
                                           double c1, double c2, double c3, double c4) {
    const double pr = dom1 ? prv.a + prv.b : prv.a - prv.b, pi = dom1 ? prv.c - prv.d : prv.c + prv.d;
    const bool dom_zero = (dr == 0.0 && di == 0.0), prev_zero = (pr == 0.0 && pi == 0.0);
    if (dom_zero == prev_zero) return make_double2(0.0, 0.0);
    const double tie = (prev_zero ? tone_tie(c1, c2, c3, c4) : tone_tie(prv.a, prv.b, prv.c, prv.d)) ? 1.0 : 0.0;
    double th = (80.0 * kPi / kFs) * fo_sum;
    th -= kTwoPi * rint(th / kTwoPi);
    double sn, cs;
    sincos(th, &sn, &cs);
    // multiply by (-/+ j)^k : tone 1 rotates by -pi/2 per symbol, tone 2 by +pi/2
    const unsigned q = (unsigned)((dom1 ? (4u - (unsigned)(ksym & 3u)) : (unsigned)(ksym & 3u)) & 3u);
    double er2 = cs, ei2 = sn;
    if (q == 1u) { er2 = -sn; ei2 = cs; }
    else if (q == 2u) { er2 = -cs; ei2 = -sn; }
    else if (q == 3u) { er2 = sn; ei2 = -cs; }
    double vr = dr, vi = di;
    if (dom_zero) {                                 // P = S_prev * (-/+ j) * X40_prev
        const double jr = dom1 ? pi : -pi, ji = dom1 ? -pr : pr;
        vr = jr * prv.x40c - ji * prv.x40s;
        vi = jr * prv.x40s + ji * prv.x40c;
    }
    const double qr = vr * er2 + vi * ei2;          // v * conj(E)
    const double qi = vi * er2 - vr * ei2;
    return make_double2((qr < 0.0 && qi < 0.0) ? kPi : 0.0, tie);
}

}  // namespace

#include "opv_atan2.h"  // kOpvAtanTabQ3R (constant-memory image of the angle table) + the host statement of the routine

// WPB = wavefronts (= streams) per workgroup. One wave per workgroup is the natural shape, but the dispatcher
// places single-wave workgroups without regard to SIMDs: with 1024 of them on the chip's 1024 SIMDs, 88 SIMDs
// received two waves and 88 none (census from this kernel's own HW_ID tap, profiles/r02_placement_1024_streams.txt),
// and the doubled-up waves - this kernel keeps a SIMD's fp64 pipe 80 % busy on its own - took 1.74x as long and set
// the kernel's duration. A 256-thread workgroup's four waves always land on the four SIMDs of one CU, so from 513
// streams on the shim launches four streams per workgroup (k_msk_frontend_wg4). Waves of a workgroup share nothing
// but the atan table.
// (A two-waves-per-stream mapping and the round-1 symbol body were measured against this one - 6 % and 25 % slower -
// and removed in round 6: NOTEBOOK.md has the measurements, git the code.)
template <int WPB>
__device__ __forceinline__ void msk_frontend_body(OpvStream* __restrict__ streams, OpvGlobalCfg cfg, int n_streams,
                                                  unsigned char* lds_all) {
    const int lane = threadIdx.x & 63;
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (a constant 0 lets the ring base fold into the tap address)
    const int sidx = (int)blockIdx.x * WPB + wave;
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();

    unsigned char* const lds = lds_all + wave * kTabOff;      // this wave's ring + guard
    const unsigned char* ringb = lds;
    double* atab = reinterpret_cast<double*>(lds_all + WPB * kTabOff);   // filled by the kernel wrapper, barrier included
    if (sidx >= n_streams) return;       // a last, partly filled workgroup
    OpvStream& st = streams[sidx];

    // ---- per-lane constants -------------------------------------------------------------
    // rows of 15 samples (lane 16 r + n <-> sample 15 r + n, n < 15; a row's last lane repeats its neighbour's
    // sample and carries no weight), so that the 60 samples take 15 broadcast steps per row instead of 16
    const int jsamp = lane - (lane >> 4) - ((lane & 15) == 15 ? 1 : 0);
    const double kf = (double)(jsamp - 10);
    const double kfs = kf * kDeltaPerHz;
    // T_1[i] = exp(-j 2 pi i / 160) = (a, b) = (cos(pi i/80), -sin(pi i/80)); zero outside a gate's window.
    // Output t = lane & 15 of every row is one of the symbol's window sums, and wr[n] / wi[n] are the weights of
    // Re / Im Z of the row's n-th sample in it (see `symbol_r`). Correlations C_1 = sum Z conj(T_1[i]) = (sum Zr a + Zi b,
    // sum Zi a - Zr b), C_2 = sum Z T_1[i] = (sum Zr a - Zi b, sum Zi a + Zr b); Re in t, Im in t + 8:
    //   t = 0, 1   on-time correlation of tone 1 / tone 2 (S_1, S_2)                          (window j in [10, 50), i = j - 10)
    //   t = 2..5   early / late correlation of tone 1, then of tone 2 (E_1, L_1, E_2, L_2)    (early: j in [0, 40), i = j;
    //                                                                                           late: j in [20, 60), i = j - 20)
    //   t = 6, 7, 14, 15   the on-time sums P1 = sum Zr a, P2 = sum Zi b, P3 = sum Zi a, P4 = sum Zr b
    //              (S_1 = (P1 + P2, P3 - P4), S_2 = (P1 - P2, P3 + P4): what the phase detector and the carry use)
    double wr[15], wi[15];
    {
        const int row = lane >> 4, t = lane & 15, u = t & 7;
        const bool is_p = u >= 6, imag = t >= 8;
        const int gate = (u < 2 || is_p) ? 1 : ((u & 1) ? 2 : 0);  // 0 early, 1 on-time, 2 late (u = 2: E_1, 3: L_1, 4: E_2, 5: L_2)
        const bool tone2 = is_p ? false : (u < 2 ? u == 1 : u >= 4);
#pragma unroll
        for (int n = 0; n < 15; ++n) {
            const int i = 15 * row + n - 10 * gate;
            double sn, cs;
            sincospi((double)i / 80.0, &sn, &cs);
            const bool in = i >= 0 && i < 40;
            const double a = in ? cs : 0.0, b = in ? -sn : 0.0;
            double r_, i_;
            if (is_p) { r_ = (t == 6) ? a : (t == 15 ? b : 0.0); i_ = (t == 7) ? b : (t == 14 ? a : 0.0); }   // P1, P4 | P2, P3
            else if (!imag) { r_ = a; i_ = tone2 ? -b : b; }      // Re C_1 / C_2
            else { r_ = tone2 ? b : -b; i_ = a; }                 // Im C_1 / C_2
            wr[n] = r_; wi[n] = i_;
            asm volatile("" : "+v"(wr[n]), "+v"(wi[n]));
        }
    }
    SinCosK sck;
    sck.s0 = -0x1.5555555555412p-3; sck.s1 = 0x1.1111110f26395p-7; sck.s2 = -0x1.a019e48059d90p-13; sck.s3 = 0x1.7150a543fadc5p-19;
    sck.c0 = -0x1.0000000000000p-1; sck.c1 = 0x1.5555555555014p-5; sck.c2 = -0x1.6c16c16818f3fp-10;
    sck.c3 = 0x1.a019dfaa26924p-16; sck.c4 = -0x1.276f06eab6283p-22;
    // Loop constants parked in VGPRs: one wave per SIMD has registers to spare, while hipcc
    // otherwise keeps re-forming 64-bit literals in SGPR pairs inside the loop.
    double kc_tfmax = 0.1, kc_beta = 0.00001, kc_alpha = 0.005, kc_fomax = 2000.0, kc_eps = 1e-10;
    double kc_tiny = 1e-100;
    asm volatile("" : "+v"(kc_tiny));
    double kc_halfpi = 1.57079632679489661923, kc_32 = 32.0, kc_m1_32 = -1.0 / 32.0, kc_gain = st.afc_alpha * (kSymRate / kTwoPi);
    // rows per unit of the angle table, its inverse, 1.5 * 2^52 + the row offset (the 1025 rows of opv_atan2_q3r)
    [[maybe_unused]] double kc_64 = 512.0, kc_m1_64 = -1.0 / 512.0, kc_magic = 6755399441055744.0 + 512.0;
    asm volatile("" : "+v"(kc_64), "+v"(kc_m1_64), "+v"(kc_magic));
    asm volatile("" : "+v"(kc_tfmax), "+v"(kc_beta), "+v"(kc_alpha), "+v"(kc_fomax), "+v"(kc_eps));
    asm volatile("" : "+v"(kc_halfpi), "+v"(kc_32), "+v"(kc_m1_32), "+v"(kc_gain));
    const double kc_nfomax = __builtin_canonicalize(-kc_fomax), kc_ntfmax = __builtin_canonicalize(-kc_tfmax);
    kc_fomax = __builtin_canonicalize(kc_fomax);
    kc_tfmax = __builtin_canonicalize(kc_tfmax);
    double sx = 1.0, nsg = 1.0;          // +/-1.0 rebuilt per symbol by rewriting the high word only
    asm volatile("" : "+v"(sx), "+v"(nsg));

    // ---- carry ---------------------------------------------------------------------------
    double fo = st.freq_offset, tf = st.timing_freq, mu = st.mu;
    PrevSums qp{st.p1r, st.p1i, st.p2r, st.p2i, st.x40c, st.x40s}, qq{0, 0, 0, 0, 1, 0};    // previous on-time P1..P4 (S_1 = (a+b, c-d), S_2 = (a-b, c+d))
    double fo_sum = st.fo_sum;
    uint32_t origin = uni((uint32_t)st.origin);
    const uint32_t n_avail = uni((uint32_t)st.n_avail);
    uint64_t n_soft = st.n_soft, total_samples = st.total_samples;
    uint32_t n_chunks = uni(st.n_chunks);
    int tail_done = (int)uni((uint32_t)st.tail_done);
    const int eof = (int)uni((uint32_t)st.eof);
    int overflow = (int)uni((uint32_t)st.overflow), stalled = 0;
    uint32_t edge_ties = uni(st.edge_ties);
    const uint64_t cap_soft = st.cap_soft;
    if (cap_soft > (1ull << 28)) overflow = 1;  // byte offsets into the soft ring are kept in 32 bits
    // oldest soft symbol the tracker may still read: its 24-symbol window, or the payload / next
    // sync check hanging off the current anchor
    uint64_t soft_keep = st.trk_next >= 24 ? st.trk_next - 24 : 0;
    if (st.trk_state != 0 && st.trk_anchor < soft_keep) soft_keep = st.trk_anchor;
    const uint32_t soft_bmask = (uint32_t)(cap_soft * 8u - 1u) & ~7u;
    gbyte* const soft_base = (gbyte*)st.soft;
    const gbyte* iq_bytes = (const gbyte*)st.iq;
    const uint64_t n_bytes = (uint64_t)n_avail * 4u;

    // ---- tile staging (wave-uniform state) ----------------------------------------------
    // Direct-to-LDS 16-byte load (global_load_lds_dwordx4): lane l moves 16 B from its own global
    // address to LDS byte (m0 + 16 l). Issued through inline asm on purpose: hipcc's waitcnt pass
    // would otherwise drain vmcnt(0) before EVERY later LDS read it cannot disambiguate from the
    // DMA destination. Completion is awaited explicitly with s_waitcnt vmcnt(0) one tile later
    // (see the tile events below).
    auto glds16 = [&](const gbyte* gsrc, uint32_t lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(uni(lds_byte))
                     : "memory");
    };
    const uint32_t lds_base = uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds);
    auto issue_tile = [&](uint32_t t) {
        // tile t -> slot t&1: 8 wave instructions; an even tile's first 16 B are mirrored into the
        // guard behind the ring. The capture's last, incomplete 16 bytes (n_avail not a multiple
        // of 4 samples) are copied sample by sample: nothing past n_avail is ever read.
        const uint64_t base = (uint64_t)t * OPV_TILE_BYTES;
        const uint32_t slot = (t & 1u) * OPV_TILE_BYTES;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t in_tile = (uint32_t)r * 1024u + (uint32_t)lane * 16u;
            const uint64_t off = base + in_tile;
            if (off + 16u <= n_bytes) glds16(iq_bytes + off, lds_base + slot + (uint32_t)r * 1024u);
            else if (off < n_bytes) {
                for (uint32_t j = 0; off + 4u * j < n_bytes; ++j)
                    *reinterpret_cast<int*>(lds + slot + in_tile + 4u * j) = *reinterpret_cast<const __attribute__((address_space(1))) int*>(iq_bytes + off + 4u * j);
            }
        }
        if ((t & 1u) == 0u && lane == 0) {
            if (base + 16u <= n_bytes) glds16(iq_bytes + base, lds_base + kRingBytes);
            else
                for (uint32_t j = 0; base + 4u * j < n_bytes && j < 4u; ++j)
                    *reinterpret_cast<int*>(lds + kRingBytes + 4u * j) = *reinterpret_cast<const __attribute__((address_space(1))) int*>(iq_bytes + base + 4u * j);
        }
    };
    // lowest sample any lane can touch at the first symbol of this launch
    uint32_t t_lo = (origin >= kBack ? origin - kBack : 0u) / kTile;
    issue_tile(t_lo);
    issue_tile(t_lo + 1u);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): both tiles (and the guard) landed
    bool evt_issue = true;               // next tile event: request tile t_lo+2 (else: wait for the newest)
    uint32_t next_evt = (t_lo + 1u) * kTile + kBack;

    for (;;) {
        // ---- which demodulate() call comes next (ref :1026 / :1088 / :1173) ----------------
        const uint32_t remaining = n_avail - origin;
        uint32_t N;
        bool last = false;
        if (cfg.streaming) {
            if (remaining >= OPV_CHUNK) N = OPV_CHUNK;
            else if (eof && !tail_done && remaining > 0) { N = remaining; last = true; }
            else { if (eof) tail_done = 1; break; }
        } else {
            if (!eof || tail_done) break;
            N = n_avail;
            last = true;
        }
        // worst case one symbol per 38 samples: postpone the call rather than overrun soft symbols the tracker
        // still needs (back-pressure: it is retried next round, once frames have been popped)
        if (overflow) break;
        if ((n_soft - soft_keep) + (uint64_t)(N / 38u + 2u) > cap_soft) { stalled = 1; break; }

        const double Nd = (double)N;
        double pos = mu;                                   // ref :217
        const uint32_t soft_off0 = ((uint32_t)n_soft * 8u) & soft_bmask;  // ring byte offset of this call's first symbol
        uint32_t soft_off = soft_off0;
        asm volatile("" : "+v"(soft_off));            // lives in a VGPR: it is the store's address operand

        // Tile bookkeeping for the symbol at `at` (wave-uniform, rare). Returns how many FOLLOWING
        // symbols need neither it nor the end-of-call test: pos advances by at most 42 samples per
        // symbol (|adj| <= 2, ref :285-286).
        auto housekeeping = [&](double at) {
            const uint32_t b = uni((uint32_t)at);
            const uint32_t gb = origin + b;                // global index of floor(pos)
            while (gb >= next_evt) {
                if (evt_issue) {
                    // the lowest tap has left tile t_lo for good: refill its slot with tile
                    // t_lo+2 (asynchronous; first needed a whole tile = ~51 symbols from now)
                    issue_tile(t_lo + 2u);
                    ++t_lo;
                    evt_issue = false;
                    next_evt = (t_lo + 1u) * kTile - kAhead;
                } else {
                    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): requested a tile ago, no stall
                    evt_issue = true;
                    next_evt = (t_lo + 1u) * kTile + kBack;
                }
            }
            // j more symbols are safe iff gb + 42 j + 1 < next_evt and b + 1 + 42 j + 50 <= N
            const int lim1 = (int)(next_evt - gb) - 2, lim2 = (int)N - 52 - (int)b;
            int lim = lim1 < lim2 ? lim1 : lim2;
            if (lim < 0) lim = 0;
            return ((uint32_t)lim * 1560u) >> 16;          // <= floor(lim / 42), lim < 2^17
        };

        // One interpolated sample per lane (ref :122-128, :232-238): the two int16 IQ words around
        // pos + kf. Issued for symbol k+1 as soon as pos(k+1) exists.
        int w0 = 0, w1 = 0;
        double f = 0.0;
        uint32_t tap_byte = 0;
        auto fetch_addr = [&](double at, bool clamp0) {
            double p = at + kf;
            if (clamp0) p = fmax(p, 0.0);                  // early gate before the chunk: s[0] (ref :237)
            const int idx = (int)p;
            f = __builtin_amdgcn_fract(p);                 // p - idx, p >= 0
            tap_byte = (((uint32_t)idx + origin) << 2) & (kRingBytes - 4u);
        };
        auto fetch_read = [&]() {
            const int* tap = reinterpret_cast<const int*>(ringb + tap_byte);
            w0 = tap[0];
            w1 = tap[1];
        };
        auto fetch = [&](double at, bool clamp0) {
            fetch_addr(at, clamp0);
            fetch_read();
        };

        // One symbol, with the ROW-BROADCAST reduction. gfx950's DP-ALU DPP form
        //   v_fmac_f64_dpp acc, src0 row_newbcast:n, src1      acc[l] += src0[row(l), lane n] * src1[l]
        // lets the 16 lanes of a row form 16 differently weighted sums of the row's samples, one broadcast step per
        // sample and component: 2 x 15 full-rate FMACs produce ALL twelve window sums of the symbol (on-time P1..P4, early
        // and late correlations of BOTH tones) as row partials in lanes t = lane & 15, with no product instructions, no
        // dominant-tone dependence and no v_readlane. One all-reduce over the four rows (two permlane swaps of the value
        // with a copy of itself) completes them; v_mov_b64_dpp row_newbcast hands the numbers the loop filters need to
        // every lane (one instruction per double instead of two v_readlane + the SGPR-operand restrictions), and the four
        // early / late energies are formed lane-parallel (square, row_ror:8, add: Re at t, Im at t + 8), the dominant
        // tone's pair selected lane-parallel (row_shl:2), before two of them are handed out. The phase detector's angle
        // comes without the octant fix-up (opv_atan2.h: opv_atan2_q3r, 1025-row table of pi/4 + atan, cubics in the argument itself; round 2: 257 rows, degree 5).
        // Scheduling notes: hipcc counts an asm block as no wait state and pads the fp64 instruction behind one with an
        // s_nop; every block here is therefore followed by a 32-bit instruction that was needed anyway, and the wait
        // states DPP reads / permlane swaps need behind a VALU write are filled with useful instructions, not s_nop.
        // (scripts/microbench/dpp64.hip has the instruction costs.)
        auto symbol_r = [&](auto tag, PrevSums& cur, const PrevSums& prv) {
            constexpr bool kFirst = decltype(tag)::first;
            constexpr bool kWide = decltype(tag)::wide;
            // The LO factor FIRST: it needs fo only, and its twelve instructions are the cover the tap read of the previous
            // symbol's tail still lacked (round 3: without that read the kernel ran 34 cycles per symbol faster, i.e. ~26 of
            // its latency were exposed when the unpack below came first)
            double xs, xc;
            if constexpr (kWide) {
                if (__builtin_expect(uni_lt(2000.0, fabs(fo)), 0)) sincos(kfs * fo, &xs, &xc);
                else expj_small(kfs, fo, sck, xs, xc);
            } else {
                expj_small(kfs, fo, sck, xs, xc);
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t my_soft_off = soft_off;
            uint32_t soft_off_next = soft_off + 8u;                 // (32-bit filler behind the asm block above)
            asm volatile("" : "+v"(soft_off_next));
            __builtin_amdgcn_sched_barrier(0);
            const int s0r = (int)(short)(w0 & 0xFFFF), s0i = w0 >> 16;      // ref :1023
            const int d_r = (int)(short)(w1 & 0xFFFF) - s0r, d_i = (w1 >> 16) - s0i;
            const double lr = fma(f, (double)d_r, (double)s0r);              // ref :122-128
            const double li = fma(f, (double)d_i, (double)s0i);
            const double zr = fma(lr, xc, li * xs);                         // Z = Lam * conj(X)
            const double zi = fma(li, xc, -(lr * xs));
            __builtin_amdgcn_sched_barrier(0);

            // ---- 1. the twelve window sums: row partials by broadcast-FMAC into two accumulators (the two moves are
            // the wait states a DPP read needs behind the VALU write of Z)
            double acc0, acc1, v, t;
#define OPV_RB(N) "v_fmac_f64_dpp %[a0], %[zr], %[r" #N "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %[a1], %[zi], %[i" #N "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
            asm("v_mov_b64 %[a0], 0\n\tv_mov_b64 %[a1], 0\n\t" OPV_RB(0) OPV_RB(1) OPV_RB(2) OPV_RB(3) OPV_RB(4) OPV_RB(5) OPV_RB(6) OPV_RB(7)
                : [a0] "=&v"(acc0), [a1] "=&v"(acc1)
                : [zr] "v"(zr), [zi] "v"(zi), [r0] "v"(wr[0]), [i0] "v"(wi[0]), [r1] "v"(wr[1]), [i1] "v"(wi[1]), [r2] "v"(wr[2]), [i2] "v"(wi[2]),
                  [r3] "v"(wr[3]), [i3] "v"(wi[3]), [r4] "v"(wr[4]), [i4] "v"(wi[4]), [r5] "v"(wr[5]), [i5] "v"(wi[5]),
                  [r6] "v"(wr[6]), [i6] "v"(wi[6]), [r7] "v"(wr[7]), [i7] "v"(wi[7]));
            __builtin_amdgcn_sched_barrier(0);
            soft_off = soft_off_next & soft_bmask;                  // (32-bit filler between the two blocks)
            asm volatile("" : "+v"(soft_off));
            __builtin_amdgcn_sched_barrier(0);
            // second half, the sum of the two accumulators and the copy the all-reduce swaps with
            asm(OPV_RB(8) OPV_RB(9) OPV_RB(10) OPV_RB(11) OPV_RB(12) OPV_RB(13) OPV_RB(14)
                "v_add_f64 %[v], %[a0], %[a1]\n\t"
                "v_mov_b64 %[t], %[v]"
                : [a0] "+v"(acc0), [a1] "+v"(acc1), [v] "=&v"(v), [t] "=&v"(t)
                : [zr] "v"(zr), [zi] "v"(zi), [r8] "v"(wr[8]), [i8] "v"(wi[8]), [r9] "v"(wr[9]), [i9] "v"(wi[9]), [r10] "v"(wr[10]), [i10] "v"(wi[10]),
                  [r11] "v"(wr[11]), [i11] "v"(wi[11]), [r12] "v"(wr[12]), [i12] "v"(wi[12]), [r13] "v"(wr[13]), [i13] "v"(wi[13]),
                  [r14] "v"(wr[14]), [i14] "v"(wi[14]));
#undef OPV_RB
            __builtin_amdgcn_sched_barrier(0);
            // ---- all-reduce over the four rows; the X[40] hand-over (sample 50 sits in row 3, lane 5) fills the two wait
            // states a swap needs behind the copy
            cur.x40c = readlane_d(xc, 53);
            __builtin_amdgcn_sched_barrier(0);
            v = swap32_add(v, t);
            asm("v_mov_b64 %0, %1" : "=v"(t) : "v"(v));
            __builtin_amdgcn_sched_barrier(0);
            cur.x40s = readlane_d(xs, 53);
            __builtin_amdgcn_sched_barrier(0);
            v = swap16_add(v, t);
            __builtin_amdgcn_sched_barrier(0);
            double fo_sum_next = fo_sum + fo;                       // sum of the fo every symbol USED
            asm volatile("" : "+v"(fo_sum_next));
            const double sq = v * v;
            __builtin_amdgcn_sched_barrier(0);
            // energies, lane-parallel (Re at t, Im at t + 8): t = 0..5 -> |S_1|^2, |S_2|^2 (on-time), |E_1|^2, |L_1|^2, |E_2|^2, |L_2|^2
            const double shv = mkd(__builtin_amdgcn_mov_dpp(dhi(v), 0x128, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(dlo(v), 0x128, 0xF, 0xF, true));
            const double en = fma(shv, shv, sq);
            __builtin_amdgcn_sched_barrier(0);
            // on-time sums P1..P4 to every lane (`old` operands: dead temporaries of the stages above)
            const double P1o = row_bcast<6>(zr, v), P2o = row_bcast<7>(zi, v), P3o = row_bcast<14>(lr, v), P4o = row_bcast<15>(li, v);
            __builtin_amdgcn_sched_barrier(0);
            // soft value = |S_2|^2 - |S_1|^2 (ref :264-268): the difference of the two energies like the reference, each from its
            // own correlation. (Where the reference's tones tie exactly - a real or imaginary Z - so do these: the two
            // correlations' accumulation chains are mirror images.)
            const double en1 = row_bcast<0>(acc0, en), en2 = row_bcast<1>(acc1, en);   // (behind the four hand-outs above: en's wait states)
            const double soft = en2 - en1;
            __builtin_amdgcn_sched_barrier(0);
            // the dominant tone's early / late pair moves to t = 2, 3 (row_shl:2 brings tone 2's over), then one hand-out each;
            // tone 1 iff e1 > e2 (ref :272 / :291; a tie gives +0: tone 2). sg = +1 for tone 1, -1 for tone 2.
            const bool dom1 = soft < 0.0;
            nsg = mkd((dhi(soft) & (int)0x80000000) | (dhi(nsg) & 0x7fffffff), dlo(nsg));
            asm volatile("" : "+v"(nsg));                           // (updated in place: no copy of the low word)
            const double sg = -nsg;
            const double oth = mkd(__builtin_amdgcn_mov_dpp(dhi(en), 0x102, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(dlo(en), 0x102, 0xF, 0xF, true));
            __builtin_amdgcn_sched_barrier(0);
            double seln = dom1 ? en : oth;
            asm volatile("" : "+v"(seln));                          // (the select stays here: the phase detector's arithmetic below is its two wait states)
            __builtin_amdgcn_sched_barrier(0);

            [[maybe_unused]] double pd = 0.0, cx = 0, cy = 0, sum = 1.0, dif = 0, dm_ = 1.0, ee, el;
            [[maybe_unused]] uint64_t zmask = 0;
            if constexpr (!kFirst) {
                // phase detector operands: dom * conj(prev) (ref :299), prev advanced by one symbol's LO rotation
                // (header: P_t = S_t (-/+ j) X[40]); sg picks the dominant tone's combination of the shared sums
                const double dr = fma(sg, P2o, P1o), di = fma(-sg, P4o, P3o);
                const double prs = fma(sg, prv.a, prv.b), pis = fma(sg, prv.c, -prv.d);
                const double ar = fma(dr, prs, di * pis), ai = fma(di, prs, -(dr * pis));
                cy = fma(ar, prv.x40c, ai * prv.x40s);              // Im z
                cx = fma(ar, prv.x40s, -(ai * prv.x40c));           // Re z
                __builtin_amdgcn_sched_barrier(0);
                ee = row_bcast<2>(sq, seln); el = row_bcast<3>(shv, seln);   // hand-outs of the two energies
                __builtin_amdgcn_sched_barrier(0);
                // cx < 0: pi - pd, as a +/-1 multiplier and a 0/pi offset built from the sign bit
                sx = mkd((dhi(cx) & (int)0x80000000) | (dhi(sx) & 0x7fffffff), dlo(sx));
                asm volatile("" : "+v"(sx));
                sum = fabs(cx) + fabs(cy); dif = fabs(cy) - fabs(cx);
                // the divisor's guard against digital silence: sum + 1e-100 IS sum unless sum is 0 (a non-zero sum of
                // products of window sums is far above 1e-84), and an add needs no canonicalised operand where fmax does
                dm_ = sum + kc_tiny;
                // One wave per workgroup (WPB == 1): the silence test's compare HERE, its branch 35 instructions later - a v_cmp
                // whose mask a scalar branch reads in the next instruction costs a lone wave a VALU -> SALU round trip (798 ->
                // 774 cycles per symbol on clean 60-frame streams, 779 -> 774 on the bench workload, although this form issues
                // one instruction more, an s_cmp on the mask). Four waves per workgroup keep the adjacent pair: with the CU's
                // other three SIMDs busy the early form measured 850 cycles per symbol against 804 (1024 streams).
                if constexpr (WPB == 1) {
                    zmask = __builtin_amdgcn_fcmp(sum, 0.0, 1 /*FCMP_OEQ*/);
                    asm volatile("" : "+s"(zmask));
                }
            } else {
                ee = row_bcast<2>(sq, seln); el = row_bcast<3>(shv, seln);
            }
            const double den = el + ee + kc_eps;                    // ted = num/den (ref :275/:279)

            // ---- 3. divides (one reciprocal for both), timing loop, AFC --------------------------------
            double ted, h = 0;
            [[maybe_unused]] double2 c01{0, 0}, c23{0, 0}, c45{0, 0};
            if constexpr (kFirst) {
                double y = __builtin_amdgcn_rcp(den);
                const double num = el - ee;
                y = fma(fma(-den, y, 1.0), y, y);
                y = fma(fma(-den, y, 1.0), y, y);
                ted = num * y;
                ted = fma(fma(-den, ted, num), y, ted);
            } else {
                const double dm = dm_;                              // max(|cx| + |cy|, 1e-100); digital silence: 0/1e-100 = 0, fixed up below
                const double tt = den * dm;
                double y = __builtin_amdgcn_rcp(tt);
                const double num = el - ee;                         // (fills the wait state behind the reciprocal)
                // ONE Newton step: v_rcp_f64 is good to 2^-24.4, one step to 2^-48.7, and both quotients below get their
                // own residual step - the same error profile as with two (scripts/microbench/rcp_accuracy.hip)
                y = fma(fma(-tt, y, 1.0), y, y);
                const double iden = y * dm, idm = y * den;
                // q = (|cy| - |cx|) / (|cy| + |cx|) in [-1, 1], good to 2^-48 without a residual step: 3.5e-15 rad on the
                // angle, i.e. 3e-14 Hz x afc_alpha / 0.001 on fo - nothing rounds on it the way pos does on ted
                const double ratio = dif * idm;
                // nearest expansion point k/128, k = -128..128, by the 1.5 * 2^52 trick: the sum's low word IS the row index
                // k + 128, and subtracting the constant gives k as a double - no v_rndne, no v_cvt
                const double kt = fma(ratio, kc_64, kc_magic);
                h = ratio;                                          // (the rows' cubics are written in the argument itself)
                const unsigned char* rowb = reinterpret_cast<const unsigned char*>(atab) + ((unsigned)dlo(kt) << 5);
                const double2* trow = reinterpret_cast<const double2*>(rowb);
                c23 = trow[1]; c01 = trow[0];
                ted = num * iden;
                ted = fma(fma(-den, ted, num), iden, ted);
            }
            __builtin_amdgcn_sched_barrier(0);
            tf = clampd(fma(kc_beta, ted, tf), kc_ntfmax, kc_tfmax);  // beta (ref :118,:283-284)
            const double adj = fma(kc_alpha, ted, tf);                // alpha (ref :117,:285); the +/-2 clamp (:286) cannot act
            pos += 40.0 + adj;                                        // ref :313
            fetch_addr(pos, false);
            __builtin_amdgcn_sched_barrier(0);
            fetch_read();                                             // next symbol's taps requested as soon as their address exists
            __builtin_amdgcn_sched_barrier(0);
            *(gdouble*)(soft_base + my_soft_off) = soft;
            if constexpr (!kFirst) {
                const double pd_off = fma(-sx, kc_halfpi, kc_halfpi);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC17F);                 // lgkmcnt(1): the table row landed (only the tap read may be in flight)
                __builtin_amdgcn_sched_barrier(0);
                pd = fma(c23.y, h, c23.x);                          // the row's cubic in q: pi/4 + atan(q) to 4e-14 rad
                pd = fma(pd, h, c01.y);
                pd = fma(pd, h, c01.x);
                pd = fma(sx, pd, pd_off);
                pd = mkd((dhi(pd) & 0x7fffffff) | (dhi(cy) & (int)0x80000000), dlo(pd));  // sign of cy
                if (__builtin_expect(WPB == 1 ? zmask != 0ull : uni_eq(sum, 0.0), 0)) {   // digital silence on either side
                    const double dr = fma(sg, P2o, P1o), di = fma(-sg, P4o, P3o);
                    const double2 sp = silence_pd(dr, di, prv, soft < 0.0, fo_sum, n_soft + (((my_soft_off - soft_off0) & soft_bmask) >> 3),
                                                  P1o, P2o, P3o, P4o);
                    pd = sp.x;
                    edge_ties += uni((uint32_t)sp.y);
                }
                const double fo_new = fma(kc_gain, pd, fo);         // ref :300-303
                asm("v_max_f64 %0, %1, %2\n\tv_min_f64 %0, %0, %3" : "=&v"(fo) : "v"(fo_new), "v"(kc_nfomax), "v"(kc_fomax));
            }
            fo_sum = fo_sum_next;
            cur.a = P1o; cur.b = P2o; cur.c = P3o; cur.d = P4o;     // prev <- this symbol's on-time correlations (ref :309-310)
        };

        auto sym = [&](auto tag, PrevSums& cur, const PrevSums& prv) { symbol_r(tag, cur, prv); };

        if (uni_lt(pos + 40.0 + 10.0, Nd)) {               // ref :221
            (void)housekeeping(pos);
            fetch(pos, true);
            sym(TagFirst{}, qp, qp);
            if (__builtin_expect(uni_lt(2000.0, fabs(fo)), 0) && uni_lt(pos + 40.0 + 10.0, Nd)) {
                (void)housekeeping(pos);                   // an out-of-range -o is still in force for one more symbol
                fetch(pos, false);
                sym(TagSecond{}, qq, qp);
                qp = qq;
            }
            // Batches: the end-of-call test and the tile events once, then as many symbols as are
            // provably clear of both. Every symbol fetches its successor's taps; across a batch
            // boundary that fetch is speculative (LDS only, harmless) and is repeated after the
            // bookkeeping. Symbols go in pairs so that the "previous correlation" registers
            // alternate instead of being copied.
            while (uni_lt(pos + 40.0 + 10.0, Nd)) {        // ref :221
                uint32_t pairs = uni(housekeeping(pos)) >> 1;
                fetch(pos, false);
                // four symbols per trip: a taken branch costs a lone wave 36 cycles (scripts/microbench/dpp64.hip, empty loop)
                for (uint32_t quads = pairs >> 1; quads != 0u; --quads) {
                    sym(TagSteady{}, qq, qp);
                    sym(TagSteady{}, qp, qq);
                    sym(TagSteady{}, qq, qp);
                    sym(TagSteady{}, qp, qq);
                }
                pairs &= 1u;
                for (; pairs != 0u; --pairs) {
                    sym(TagSteady{}, qq, qp);
                    sym(TagSteady{}, qp, qq);
                }
                sym(TagSteady{}, qq, qp);
                qp = qq;
            }
        }

        // ---- end of this demodulate() call (ref :318-328, :1067-1076) ------------------------
        const uint32_t nsym_call = ((soft_off - soft_off0) & soft_bmask) >> 3;
        const uint32_t used = uni((uint32_t)pos);
        mu = pos - (double)used;
        const uint32_t leftover = N - used;
        if (lane == 0) {                                   // chunk log is a ring
            double* c = st.chunk_log + 5 * (size_t)(n_chunks % st.cap_chunks);
            c[0] = fo;
            c[1] = tf; c[2] = mu; c[3] = (double)leftover; c[4] = (double)nsym_call;
        }
        ++n_chunks;
        n_soft += nsym_call;
        total_samples += N;
        origin += (leftover > 0u && leftover < N) ? used : N;
        if (last) { tail_done = 1; break; }
    }

    if (lane == 0) {
        st.freq_offset = fo;
        st.p1r = qp.a; st.p1i = qp.b; st.p2r = qp.c; st.p2i = qp.d; st.x40c = qp.x40c; st.x40s = qp.x40s;
        st.fo_sum = fo_sum;
        st.edge_ties = edge_ties;
        st.timing_freq = tf; st.mu = mu;
        st.origin = origin; st.n_soft = n_soft; st.total_samples = total_samples;
        st.n_chunks = n_chunks; st.tail_done = tail_done; st.overflow = overflow;
        st.stalled = stalled;
        // where and at which clock this stream's wave ran (two scalar reads per launch; opv_tap_wave_info)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st.dbg_hw_id = hw; st.dbg_xcc_id = xcc;
        st.dbg_cycles = __builtin_amdgcn_s_memtime() - dbg_t0;
        st.dbg_ticks = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
}

// the front-end kernels: one wave per stream, one or four waves (streams) per workgroup
constexpr uint32_t kAtanQBytes = 1025 * 32;
template <int NT>
__device__ __forceinline__ void load_atan_table_q(unsigned char* lds_tab) {
    double* atab = reinterpret_cast<double*>(lds_tab);
    for (int i = threadIdx.x; i < 1025 * 4; i += NT) atab[i] = (&kOpvAtanTabQ3R[0][0])[i];
}
extern "C" __global__ __launch_bounds__(64) void k_msk_frontend_rb(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                    int n_streams) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[kTabOff + kAtanQBytes];
    load_atan_table_q<64>(lds_all + kTabOff);
    __syncthreads();
    msk_frontend_body<1>(streams, cfg, n_streams, lds_all);
}
extern "C" __global__ __launch_bounds__(256) void k_msk_frontend_rb_wg4(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                         int n_streams) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4 * kTabOff + kAtanQBytes];
    load_atan_table_q<256>(lds_all + 4 * kTabOff);
    __syncthreads();
    msk_frontend_body<4>(streams, cfg, n_streams, lds_all);
}
