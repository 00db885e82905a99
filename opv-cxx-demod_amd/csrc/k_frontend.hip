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
//     20..59, so the 120 interpolations of the reference become 60, one per lane.
//   * The LO is factored  lo_t[i] = E_t * T_t[i] * X[i],  E_t = exp(j ph_t) (a rotation that
//     every use below is invariant to, so it is never formed and no phase is tracked),
//     T_t[i] = exp(-/+ j 2pi i/160) a per-lane CONSTANT (13550*40 = Fs/4), and
//     X[m] = exp(j m d), d = 2pi fo/Fs, |m d| <= 0.29 rad: a 7-term Taylor pair per lane instead
//     of two libm sincos per sample. T_2 = conj(T_1), so both tones share four real products
//     per gate:  P1=sum Zr a, P2=sum Zi b, P3=sum Zi a, P4=sum Zr b with Z = Lam conj(X):
//     C_1 = (P1+P2, P3-P4), C_2 = (P1-P2, P3+P4).
//   * |.|^2 of early/late is invariant to the gate's constant rotation; for the AFC the
//     previous on-time correlation is stored already advanced by the LO rotation of one
//     symbol, P_t = S_t * (-/+ j) * X[40]  (T_t[40] = -/+ j exactly), so that
//     arg(c(k) conj(c(k-1))) = arg(S(k) conj(P(k-1))).
//   * 12 real sums are reduced over the wave with v_permlane32_swap / v_permlane16_swap
//     (reduce-scatter, 2 steps) + DPP row rotations, then broadcast through LDS; the scalar
//     loop filters run redundantly on all lanes (wave-uniform, no divergence). The phase
//     detector uses a uniform-argument atan2 (opv_atan2.h) and well-scaled divisions.
//   * int16 IQ is staged HBM -> LDS in 2168-sample frame tiles (8672 B) with direct-to-LDS
//     16-byte loads (global_load_lds_dwordx4, 1 KiB per wave instruction), two tile slots
//     forming a 4336-sample ring plus a 68-sample guard that mirrors the head of the even
//     tile so that a lane's two interpolation taps never need a wrap test; the next tile is
//     requested one whole tile (~54 symbols) before its first use. Lanes read their taps
//     straight from the int16 ring (ds_read2_b32) and widen in registers.
//   * fp64 everywhere: the 1e-5 soft contract does not need it, bit-exact quantiser/sync
//     decisions on noisy input do. No MFMA: the per-symbol contraction is
//     3x4x60 with a serial dependence between symbols.
//
// Roofline: HBM-bound on paper (4 B/sample in, 8 B/symbol out => 4.2 B/sample) but actually
// issue/latency-bound by the per-symbol feedback recurrence (a few hundred wave instructions
// per symbol at one wave per stream); see DESIGN.md and profiles/.
#include <hip/hip_runtime.h>
#include <math.h>

#include "opv_device.h"

namespace {

constexpr double kPi = 3.14159265358979323846;  // ref :43
constexpr double kTwoPi = 2.0 * kPi;            // ref :44
constexpr double kFs = 2168000.0;               // ref :40
constexpr double kSymRate = 2168000.0 / 40.0;   // ref :41
constexpr double kDeltaPerHz = kTwoPi / kFs;    // d = 2 pi fo / Fs (ref :210-211, :305-306)

constexpr uint32_t kTile = OPV_TILE_SAMPLES;    // 2168 samples
constexpr uint32_t kRing = 2 * kTile;           // 4336 samples, two tile slots
constexpr uint32_t kGuard = 68;                 // mirror of the even slot's head (272 B)
constexpr uint32_t kBack = 11;                  // lowest tap is floor(pos) - 10, one spare
constexpr uint32_t kAhead = 56;                 // highest tap is floor(pos) + 54, one spare

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
template <int CTRL>
__device__ inline double dpp_add(double v) {
    // bound_ctrl:1 => no "old" operand to materialise (every lane of a row_ror is valid anyway)
    const int lo = __builtin_amdgcn_mov_dpp(dlo(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(dhi(v), CTRL, 0xF, 0xF, true);
    return v + mkd(hi, lo);
}
// sum over the 16 lanes of a row, result in every lane of the row (row_ror 8,4,2,1)
__device__ inline double row_allsum(double v) {
    v = dpp_add<0x128>(v);
    v = dpp_add<0x124>(v);
    v = dpp_add<0x122>(v);
    v = dpp_add<0x121>(v);
    return v;
}
// d = a*b + c as a 3-operand VOP3 (hipcc otherwise copies the constant addend and uses v_fmac)
__device__ inline double fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double readlane_d(double v, int l) {
    return mkd(__builtin_amdgcn_readlane(dhi(v), l), __builtin_amdgcn_readlane(dlo(v), l));
}
// wave-uniform floating compare -> scalar branch (the operands are identical in every lane)
__device__ inline bool uni_lt(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 4 /*FCMP_OLT*/) != 0ull; }

// n/d for well-scaled operands (correlator energies: 1e0..1e25, never denormal/inf): v_rcp_f64,
// two Newton steps and one residual correction — <= 1 ulp, 6 instructions instead of the 13
// of the IEEE expansion (div_scale/div_fmas/div_fixup only matter at the exponent extremes).
__device__ inline double div_fast(double n, double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    const double q = n * y;
    return fma(fma(-d, q, n), y, q);
}

__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

}  // namespace

#include "opv_atan2.h"  // kOpvAtanTab (constant-memory image of the table) + host reference routine

namespace {
// Device form of opv_atan2 (same table, same steps), BRANCH-FREE so that hipcc can interleave it
// with the timing-loop arithmetic of the same symbol: the coefficient row is read from the LDS
// copy of the table at a per-lane (identical) address, selects replace the quadrant branches.
// r = min/max has already been formed by the caller (its divide runs beside the TED's divide).
__device__ inline double atan_from_ratio(double r, double y, double x, const double* tab_lds) {
    const double kf = fmin(floor(r * 32.0), 31.0);                  // interval index, as a double
    const int k = (int)kf;
    // interval 0 is expanded at 0 (odd series), the others at their centre (k + 0.5)/32
    const double h = r - (k ? fma(kf, 1.0 / 32.0, 1.0 / 64.0) : 0.0);
    const double2* t = reinterpret_cast<const double2*>(tab_lds + k * 10);
    const double2 c01 = t[0], c23 = t[1], c45 = t[2], c67 = t[3], c89 = t[4];
    double p = fma3(c89.y, h, c89.x);
    p = fma3(p, h, c67.y);
    p = fma3(p, h, c67.x);
    p = fma3(p, h, c45.y);
    p = fma3(p, h, c45.x);
    p = fma3(p, h, c23.y);
    p = fma3(p, h, c23.x);
    p = fma3(p, h, c01.y);
    p = fma3(p, h, c01.x);
    p = (fabs(y) > fabs(x)) ? 1.57079632679489661923 - p : p;
    p = (x < 0.0) ? 3.14159265358979323846 - p : p;
    return (y < 0.0) ? -p : p;
}
}  // namespace

extern "C" __global__ __launch_bounds__(64) void k_msk_frontend(OpvStream* __restrict__ streams,
                                                                 OpvGlobalCfg cfg) {
    OpvStream& st = streams[blockIdx.x];
    const int lane = threadIdx.x;

    // One LDS object: int16 ring (2 tiles + guard) | reduction scratch | atan table.
    constexpr int kRingBytes = (kRing + kGuard) * 4;  // 17616
    __shared__ __attribute__((aligned(16))) unsigned char lds[kRingBytes + 128 + 32 * 10 * 8];
    int* ring = reinterpret_cast<int*>(lds);
    double* red = reinterpret_cast<double*>(lds + kRingBytes);
    double* atab = reinterpret_cast<double*>(lds + kRingBytes + 128);
    for (int i = lane; i < 320; i += 64) atab[i] = (&kOpvAtanTab[0][0])[i];

    // ---- per-lane constants -------------------------------------------------------------
    const double kf = (double)(lane - 10);
    // T_1[i] = exp(-j 2 pi i / 160) = (cos(pi i/80), -sin(pi i/80)); zero outside a gate's window
    double aE = 0, bE = 0, aO = 0, bO = 0, aL = 0, bL = 0;
    {
        double sn, cs;
        if (lane < 40) { sincospi((double)lane / 80.0, &sn, &cs); aE = cs; bE = -sn; }
        if (lane >= 10 && lane < 50) { sincospi((double)(lane - 10) / 80.0, &sn, &cs); aO = cs; bO = -sn; }
        if (lane >= 20 && lane < 60) { sincospi((double)(lane - 20) / 80.0, &sn, &cs); aL = cs; bL = -sn; }
    }

    // ---- carry ---------------------------------------------------------------------------
    double fo = st.freq_offset, tf = st.timing_freq, mu = st.mu;
    const double afc_gain = st.afc_alpha * (kSymRate / kTwoPi);  // ref :300-302
    double qA = st.p1r, qB = st.p1i, qC = st.p2r, qD = st.p2i;       // previous on-time P1..P4 (S_1 = (A+B, C-D), S_2 = (A-B, C+D))
    double x40c_prev = st.x40c, x40s_prev = st.x40s;                 // X[40] of that symbol
    double fo_sum = st.fo_sum;
    uint32_t origin = uni((uint32_t)st.origin);
    const uint32_t n_avail = uni((uint32_t)st.n_avail);
    uint64_t n_soft = st.n_soft, total_samples = st.total_samples;
    uint32_t n_chunks = uni(st.n_chunks);
    int tail_done = (int)uni((uint32_t)st.tail_done);
    const int eof = (int)uni((uint32_t)st.eof);
    int overflow = (int)uni((uint32_t)st.overflow);
    const uint64_t cap_soft = st.cap_soft;
    // oldest soft symbol the tracker may still read: its 24-symbol window, or the payload / next
    // sync check hanging off the current anchor
    uint64_t soft_keep = st.trk_next >= 24 ? st.trk_next - 24 : 0;
    if (st.trk_state != 0 && st.trk_anchor < soft_keep) soft_keep = st.trk_anchor;
    const uint32_t soft_mask = (uint32_t)(cap_soft - 1);
    double* __restrict__ soft_ring = st.soft;
    const unsigned char* iq_bytes = reinterpret_cast<const unsigned char*>(st.iq);
    const uint64_t n_bytes = (uint64_t)n_avail * 4u;

    // ---- tile staging (wave-uniform state) ----------------------------------------------
    // Direct-to-LDS 16-byte load (global_load_lds_dwordx4): lane l moves 16 B from its own global
    // address to LDS byte (m0 + 16 l). Issued through inline asm on purpose: hipcc's waitcnt pass
    // would otherwise drain vmcnt(0) before EVERY later LDS read it cannot disambiguate from the
    // DMA destination (here: once per symbol, behind the soft-symbol store). Completion is
    // awaited explicitly with s_waitcnt vmcnt(0) one tile later (see the tile events below).
    auto glds16 = [&](const unsigned char* gsrc, uint32_t lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_byte)
                     : "memory");
    };
    const uint32_t lds_base = uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds);
    auto issue_tile = [&](uint32_t t) {
        // tile t -> slot t&1: 8 full + 1 partial wave instruction; an even tile's first 272 B are
        // mirrored into the guard behind the ring.
        const uint64_t base = (uint64_t)t * OPV_TILE_BYTES;
        const uint32_t slot = lds_base + (t & 1u) * OPV_TILE_BYTES;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const uint32_t in_tile = (uint32_t)r * 1024u + (uint32_t)lane * 16u;
            const uint64_t off = base + in_tile;
            if (in_tile < OPV_TILE_BYTES && off + 16u <= n_bytes) glds16(iq_bytes + off, slot + (uint32_t)r * 1024u);
        }
        if ((t & 1u) == 0u) {
            const uint64_t off = base + (uint64_t)lane * 16u;
            if (lane < 17 && off + 16u <= n_bytes) glds16(iq_bytes + off, lds_base + kRing * 4u);
        }
    };
    // lowest sample any lane can touch at the first symbol of this launch
    uint32_t gb_prev = origin;
    uint32_t t_lo = (origin >= kBack ? origin - kBack : 0u) / kTile;
    uint32_t ring_b = (origin + kRing * 4u - kBack) % kRing;  // ring slot of sample floor(pos)-11
    issue_tile(t_lo);
    issue_tile(t_lo + 1u);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): both tiles (and the guard) landed
    bool evt_issue = true;               // next tile event: request tile t_lo+2 (else: wait for the newest)
    uint32_t next_evt = (t_lo + 1u) * kTile + kBack;
    __syncthreads();                     // atan table visible (single wave: LDS ordering only)

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
        // worst case one symbol per 38 samples: refuse the call rather than overrun the soft log
        if (overflow || (n_soft - soft_keep) + (uint64_t)(N / 38u + 2u) > cap_soft) { overflow = 1; break; }

        const double Nd = (double)N;
        double pos = mu;                                   // ref :217
        double delta = fo * kDeltaPerHz;                   // fo part of phase_inc (ref :210-211)
        uint32_t nsym_call = 0;
        const uint32_t soft_pos0 = (uint32_t)n_soft & soft_mask;  // ring slot of this call's first symbol

        // Tap fetch for one symbol: tile bookkeeping (wave-uniform, scalar), then each lane's two
        // ring words. It is issued for symbol k+1 as soon as the timing loop has produced pos(k+1),
        // so the LDS latency hides under the AFC arithmetic of symbol k (software pipelining).
        int w0 = 0, w1 = 0;
        double f = 0.0;
        auto fetch = [&](double at) {
            const uint32_t b = uni((uint32_t)at);
            const uint32_t gb = origin + b;                // global index of floor(pos)
            ring_b += gb - gb_prev;
            gb_prev = gb;
            if (ring_b >= kRing) ring_b -= kRing;
            while (gb >= next_evt) {                       // rare: tile bookkeeping
                if (evt_issue) {
                    // the lowest tap has left tile t_lo for good: refill its slot with tile
                    // t_lo+2 (asynchronous; first needed a whole tile = ~54 symbols from now)
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
            // one interpolated sample per lane (ref :122-128, :232-238)
            const double p = fmax(at + kf, 0.0);           // early gate before the chunk: s[0] (ref :237)
            const int idx = (int)p;
            f = p - (double)idx;
            const uint32_t slot = ring_b + (uint32_t)(idx - (int)b + (int)kBack);  // < kRing + 66: guard covers it
            w0 = ring[slot];
            w1 = ring[slot + 1u];
        };
        bool go = uni_lt(pos + 40.0 + 10.0, Nd);           // ref :221
        if (go) fetch(pos);

        while (go) {
            const double g1 = 1.0 - f;
            const double s0r = (double)(int)(short)(w0 & 0xFFFF), s0i = (double)(w0 >> 16);  // ref :1023
            const double s1r = (double)(int)(short)(w1 & 0xFFFF), s1i = (double)(w1 >> 16);
            const double lr = fma(s1r, f, s0r * g1);
            const double li = fma(s1i, f, s0i * g1);

            // ---- X = exp(j kf delta) by Taylor (|x| <= 0.29) --------------------------------
            const double x = kf * delta;
            const double x2 = x * x;
            double sp = -1.0 / 39916800.0;                 // x^11
            sp = fma3(sp, x2, 1.0 / 362880.0);
            sp = fma3(sp, x2, -1.0 / 5040.0);
            sp = fma3(sp, x2, 1.0 / 120.0);
            sp = fma3(sp, x2, -1.0 / 6.0);
            sp = fma3(sp, x2, 1.0);
            const double xs = x * sp;                      // sin
            double cp = 1.0 / 479001600.0;                 // x^12
            cp = fma3(cp, x2, -1.0 / 3628800.0);
            cp = fma3(cp, x2, 1.0 / 40320.0);
            cp = fma3(cp, x2, -1.0 / 720.0);
            cp = fma3(cp, x2, 1.0 / 24.0);
            cp = fma3(cp, x2, -0.5);
            const double xc = fma3(cp, x2, 1.0);           // cos

            // Z = Lam * conj(X)
            const double zr = fma(lr, xc, li * xs);
            const double zi = fma(li, xc, -(lr * xs));

            // ---- 12 partial products, reduce-scatter over the wave ---------------------------
            // value order v[3r+k]: row r = P-term (P1..P4), k = gate (E,O,L)
            const double v0 = zr * aE, v1 = zr * aO, v2 = zr * aL;    // P1 = sum Zr a
            const double v3 = zi * bE, v4 = zi * bO, v5 = zi * bL;    // P2 = sum Zi b
            const double v6 = zi * aE, v7 = zi * aO, v8 = zi * aL;    // P3 = sum Zi a
            const double v9 = zr * bE, v10 = zr * bO, v11 = zr * bL;  // P4 = sum Zr b
            const double r0 = swap32_add(v0, v6), r1 = swap32_add(v1, v7), r2 = swap32_add(v2, v8);
            const double r3 = swap32_add(v3, v9), r4 = swap32_add(v4, v10), r5 = swap32_add(v5, v11);
            double q0 = swap16_add(r0, r3), q1 = swap16_add(r1, r4), q2 = swap16_add(r2, r5);
            q0 = row_allsum(q0);
            q1 = row_allsum(q1);
            q2 = row_allsum(q2);
            // row 0: P1{E,O,L}  row 1: P2  row 2: P3  row 3: P4
            if ((lane & 15) == 0) {
                double* d = red + (lane >> 4) * 3;
                d[0] = q0; d[1] = q1; d[2] = q2;
            }
            const double x40c = readlane_d(xc, 50), x40s = readlane_d(xs, 50);  // X[40] lives in lane 50
            __builtin_amdgcn_wave_barrier();
            const double P1e = red[0], P1o = red[1], P1l = red[2];
            const double P2e = red[3], P2o = red[4], P2l = red[5];
            const double P3e = red[6], P3o = red[7], P3l = red[8];
            const double P4e = red[9], P4o = red[10], P4l = red[11];
            __builtin_amdgcn_wave_barrier();

            // ---- uniform tail (all lanes, identical). Deliberately free of branches up to the
            // fetch of the next symbol: the timing chain (TED divide -> loop filter -> pos) and the
            // AFC chain (phase detector divide -> atan -> fo) are independent and each is a long
            // string of dependent fp64 operations; in one basic block hipcc interleaves them.
            const double s1r_ = P1o + P2o, s1i_ = P3o - P4o;        // S_1 (tone -13550)
            const double s2r_ = P1o - P2o, s2i_ = P3o + P4o;        // S_2 (tone +13550)
            const double en1 = s1r_ * s1r_ + s1i_ * s1i_;           // ref :264-265
            const double en2 = s2r_ * s2r_ + s2i_ * s2i_;
            const double soft = en2 - en1;                          // ref :268
            const bool dom1 = en2 < en1;                            // e1 > e2 (ref :272 / :291)
            // dominant tone of the early/late gates: C = (P1 +/- P2, P3 -/+ P4); the sign is a
            // bit flipped into the high word instead of eight selects
            const int sgn = dom1 ? 0 : (int)0x80000000;
            auto flip = [&](double v) { return mkd(dhi(v) ^ sgn, dlo(v)); };
            const double er = P1e + flip(P2e), ei = P3e - flip(P4e);
            const double lr2 = P1l + flip(P2l), li2 = P3l - flip(P4l);
            const double ee = er * er + ei * ei, el = lr2 * lr2 + li2 * li2;

            // phase detector operands: dom * conj(prev) (ref :299). prev of the reference = S_prev
            // advanced by one symbol of LO rotation, (-/+ j) X40_prev; applied to the product:
            //   z = (S conj(S_prev)) * conj(X40_prev) * (+/- j)
            // dominant tone's S = (P1 +/- P2, P3 -/+ P4), now and one symbol ago (same sign flip)
            const double dr = P1o + flip(P2o), di = P3o - flip(P4o);
            const double pr = qA + flip(qB), pi = qC - flip(qD);         // previous S of that tone
            const double ar = dr * pr + di * pi, ai = di * pr - dr * pi;
            const double ur = fma(ar, x40c_prev, ai * x40s_prev);
            const double ui = fma(ai, x40c_prev, -(ar * x40s_prev));
            const double cr = mkd(dhi(ui) ^ (sgn ^ (int)0x80000000), dlo(ui));  // tone 1: -ui, tone 2: +ui
            const double ci = mkd(dhi(ur) ^ sgn, dlo(ur));                      // tone 1: +ur, tone 2: -ur
            const double ax = fabs(cr), ay = fabs(ci);
            const double mx = fmax(ax, ay), mn = fmin(ax, ay);
            const bool degenerate = (mx == 0.0);                    // digital silence, fixed up below

            // the two divides of the symbol, side by side
            const double ted = div_fast(el - ee, el + ee + 1e-10);  // ref :275/:279
            const double ratio = div_fast(mn, degenerate ? 1.0 : mx);

            tf = clampd(fma(0.00001, ted, tf), -0.1, 0.1);          // beta (ref :118,:283-284)
            const double adj = clampd(fma(0.005, ted, tf), -2.0, 2.0);  // alpha (ref :117,:285-286)
            pos += 40.0 + adj;                                      // ref :313

            const double pd = atan_from_ratio(ratio, ci, cr, atab);
            const double fo_used = fo;
            const bool first = (nsym_call == 0);                    // no AFC on the first symbol of a call (ref :289)
            const double fo_afc = clampd(fma(afc_gain, pd, fo), -2000.0, 2000.0);  // ref :300-303
            fo = first ? fo : fo_afc;

            if (__builtin_expect(uni_i(degenerate && !first), 0)) {
                // Digital silence on either side. The reference's product (ref :299) is then
                // an exact zero whose SIGNS decide std::arg: atan2(+0,-0) = pi, everything
                // else +/-0 (IEEE). Working the signs through its complex multiply:
                //   dom == (+0,+0), prev != 0 : pi iff Re(prev) < 0 and Im(prev) < 0
                //   prev == (+0,+0), dom != 0 : pi iff Re(dom)  < 0 and Im(dom)  < 0
                //   both zero                  : 0
                // where dom/prev are the reference's correlations, i.e. ours times the
                // absolute LO phasor it carries: c_t(k) = S_t(k) conj(E_t(k)),
                // prev_t = P_t conj(E_t(k)), P_t = S_t(k-1) (-/+ j) X40(k-1),
                // E_t(k) = exp(j(-/+ k pi/2 + (80 pi/Fs) sum_{j<k} fo_j)).
                // Rare and wave-uniform; rebuilt here from the running sum of fo.
                const bool dom_zero = (dr == 0.0 && di == 0.0), prev_zero = (pr == 0.0 && pi == 0.0);
                double pdz = 0.0;
                if (dom_zero != prev_zero) {
                    const uint64_t ksym = n_soft + nsym_call;       // symbols before this one
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
                        vr = jr * x40c_prev - ji * x40s_prev;
                        vi = jr * x40s_prev + ji * x40c_prev;
                    }
                    const double qr = vr * er2 + vi * ei2;          // v * conj(E)
                    const double qi = vi * er2 - vr * ei2;
                    if (qr < 0.0 && qi < 0.0) pdz = kPi;
                }
                fo = clampd(fma(afc_gain, pdz, fo_used), -2000.0, 2000.0);
            }
            // prev <- this symbol's on-time correlations and LO rotation (ref :309-310)
            qA = P1o; qB = P2o; qC = P3o; qD = P4o;
            x40c_prev = x40c; x40s_prev = x40s;
            fo_sum += fo_used;
            delta = fo * kDeltaPerHz;

            soft_ring[(soft_pos0 + nsym_call) & soft_mask] = soft;  // all lanes, same value and address
            ++nsym_call;

            // ---- next symbol's taps --------------------------------------------------------------
            go = uni_lt(pos + 40.0 + 10.0, Nd);                     // ref :221
            if (go) fetch(pos);
        }

        // ---- end of this demodulate() call (ref :318-328, :1067-1076) ------------------------
        const uint32_t used = uni((uint32_t)pos);
        mu = pos - (double)used;
        const uint32_t leftover = N - used;
        if (lane == 0) {                                   // chunk log is a ring
            double* c = st.chunk_log + 5 * (size_t)(n_chunks % st.cap_chunks);
            c[0] = fo; c[1] = tf; c[2] = mu; c[3] = (double)leftover; c[4] = (double)nsym_call;
        }
        ++n_chunks;
        n_soft += nsym_call;
        total_samples += N;
        origin += (leftover > 0u && leftover < N) ? used : N;
        if (last) { tail_done = 1; break; }
    }

    if (lane == 0) {
        st.freq_offset = fo; st.timing_freq = tf; st.mu = mu;
        st.p1r = qA; st.p1i = qB; st.p2r = qC; st.p2i = qD; st.x40c = x40c_prev; st.x40s = x40s_prev;
        st.fo_sum = fo_sum;
        st.origin = origin; st.n_soft = n_soft; st.total_samples = total_samples;
        st.n_chunks = n_chunks; st.tail_done = tail_done; st.overflow = overflow;
    }
}
