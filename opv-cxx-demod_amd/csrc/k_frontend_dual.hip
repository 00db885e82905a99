// k_frontend_dual.hip — MSK front-end for FEW streams: TWO wavefronts per IQ stream, one per feedback loop.
//
// Same arithmetic contract as k_frontend.hip (reference src/opv-demod.cpp:206-329 + the chunker :1012-1113 /
// :1132-1173). A symbol of the one-wave kernel is a chain of ~210 issued instructions, and a wave that is alone on
// its SIMD pays ~4.6 cycles for every one of them, so with 64 streams on 1024 SIMDs the only way to go faster is a
// shorter chain per wave. The demodulator has two loops that meet only in the samples they look at:
//     timing:    taps -> on-time sums -> dominant tone -> early/late sums -> TED -> pos          (ref :271-286, :313)
//     frequency: taps -> on-time sums -> dominant tone -> dom conj(prev) -> atan2 -> fo         (ref :289-306)
// Wave T runs the first and wave F the second, each on its own SIMD of the CU; both form the interpolated
// samples and the on-time sums (the shared prefix is repeated, not handed over: a mid-symbol hand-over would put two
// LDS round trips on the cycle instead of one). At the end of a symbol T publishes pos(k+1), F publishes fo(k+1)
// (16-byte LDS slots, value then tag, polled by the other wave - no s_barrier); each needs the other's number to
// start symbol k+1. Both waves see bit-identical (pos, fo), execute the same IEEE operations on them and therefore
// take every decision (dominant tone, end of call, chunk grid) identically; the soft value is F's.
// The cycle per symbol is about (T + F) / 2 + one hand-over instead of T + F - prefix.
//
// STATUS (round 2, MI355X): exact - the parity tests pass on this mapping - but NOT faster: 1117 cycles per symbol
// against 1050 for the hand-scheduled one-wave kernel (64 streams). PMC: 156 VALU + 10 SALU + 8 LDS instructions per
// wave and symbol (the two loops do not split evenly and each repeats the 85-instruction prefix), 27 % of each wave's
// cycles in s_waitcnt (its own LDS round trips - poll, taps, atan row - are no longer covered by the other loop's
// arithmetic, plus the partner). It is therefore not selected automatically (opv_set_frontend(ctx, -2) only);
// DESIGN.md §3.1 has the arithmetic of what a hand-scheduled version could reach (~800 cycles).
//
// Tile staging is T's (it leads in pos); its events keep one more symbol of margin than the one-wave kernel's so that
// F, which is at most one symbol behind, never reads a tile that has not landed or has been recycled.
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
constexpr uint32_t kRing = 2 * kTile;
constexpr uint32_t kRingBytes = kRing * 4;      // 16384
constexpr uint32_t kGuardBytes = 16;
constexpr uint32_t kBack = 11 + 44;             // lowest tap is floor(pos) - 10; + one symbol for the lagging wave
constexpr uint32_t kAhead = 56 + 44;            // highest tap is floor(pos) + 54; + one symbol (see the file header)
constexpr uint32_t kTabOff = kRingBytes + kGuardBytes;      // 16400
constexpr uint32_t kTabRow = 10;
constexpr uint32_t kXchgOff = kTabOff + 33 * kTabRow * 8;   // 19040: 4 slots x 16 B
constexpr uint32_t kLdsBytes = kXchgOff + 64 + 16;
static_assert(kXchgOff % 16 == 0, "16-byte LDS alignment");
constexpr uint32_t kPollLimit = 1u << 22;       // a wave that waits this long gives up (sets st.overflow = 2): no hang

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) unsigned char gbyte;

__device__ inline int dlo(double v) { return __double2loint(v); }
__device__ inline int dhi(double v) { return __double2hiint(v); }
__device__ inline double mkd(int hi, int lo) { return __hiloint2double(hi, lo); }
__device__ inline double swap32_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)dlo(a), (unsigned)dlo(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)dhi(a), (unsigned)dhi(b), false, false);
    return mkd((int)hi[0], (int)lo[0]) + mkd((int)hi[1], (int)lo[1]);
}
__device__ inline double swap16_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)dlo(a), (unsigned)dlo(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)dhi(a), (unsigned)dhi(b), false, false);
    return mkd((int)hi[0], (int)lo[0]) + mkd((int)hi[1], (int)lo[1]);
}
template <int CTRL>
__device__ inline double dpp_add(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(dlo(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(dhi(v), CTRL, 0xF, 0xF, true);
    return v + mkd(hi, lo);
}
// four values -> their wave sums in lanes 0 / 16 / 32 / 48 (k_frontend.hip: reduce-scatter + row rotations)
__device__ inline double reduce4(double v0, double v1, double v2, double v3) {
    double q = swap16_add(swap32_add(v0, v2), swap32_add(v1, v3));
    q = dpp_add<0x128>(q);
    q = dpp_add<0x124>(q);
    q = dpp_add<0x122>(q);
    q = dpp_add<0x121>(q);
    return q;
}
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline double readlane_d(double v, int l) {
    return mkd(__builtin_amdgcn_readlane(dhi(v), l), __builtin_amdgcn_readlane(dlo(v), l));
}
__device__ inline bool uni_lt(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 4 /*FCMP_OLT*/) != 0ull; }
__device__ inline bool uni_eq(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 1 /*FCMP_OEQ*/) != 0ull; }
__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// exp(j x), x = kfs * fo, |x| <= 0.284: the near-minimax pair of k_frontend.hip (abs error 1e-19 / 1.3e-18), as one
// asm block with the coefficients parked in registers (hipcc would copy each into the destructive v_fmac's accumulator)
struct SinCosK {
    double s0, s1, s2, s3, s4;  // q(u) low -> high
    double c0, c1, c2, c3, c4;  // r(u) low -> high
};
__device__ inline void expj_small(double kfs, double fo, const SinCosK& k, double& xs, double& xc) {
    double x, u, p, r, t;
    asm("v_mul_f64 %[x], %[kfs], %[fo]\n\t"
        "v_mul_f64 %[u], %[x], %[x]\n\t"
        "v_fma_f64 %[p], %[s4], %[u], %[s3]\n\t"
        "v_fma_f64 %[r], %[c4], %[u], %[c3]\n\t"
        "v_fma_f64 %[p], %[p], %[u], %[s2]\n\t"
        "v_fma_f64 %[r], %[r], %[u], %[c2]\n\t"
        "v_fma_f64 %[p], %[p], %[u], %[s1]\n\t"
        "v_fma_f64 %[r], %[r], %[u], %[c1]\n\t"
        "v_fma_f64 %[p], %[p], %[u], %[s0]\n\t"
        "v_fma_f64 %[r], %[r], %[u], %[c0]\n\t"
        "v_mul_f64 %[t], %[x], %[u]\n\t"
        "v_fma_f64 %[xc], %[r], %[u], 1.0\n\t"
        "v_fma_f64 %[xs], %[t], %[p], %[x]"
        : [x] "=&v"(x), [u] "=&v"(u), [p] "=&v"(p), [r] "=&v"(r), [t] "=&v"(t), [xs] "=&v"(xs), [xc] "=&v"(xc)
        : [kfs] "v"(kfs), [fo] "v"(fo), [s0] "v"(k.s0), [s1] "v"(k.s1), [s2] "v"(k.s2), [s3] "v"(k.s3), [s4] "v"(k.s4),
          [c0] "v"(k.c0), [c1] "v"(k.c1), [c2] "v"(k.c2), [c3] "v"(k.c3), [c4] "v"(k.c4));
}

struct TagFirst { static constexpr bool first = true, wide = true; };     // first symbol of a demodulate() call
struct TagSecond { static constexpr bool first = false, wide = true; };   // second symbol under an out-of-range -o
struct TagSteady { static constexpr bool first = false, wide = false; };  // everything else

struct PrevSums {
    double a, b, c, d;  // on-time P1..P4
    double x40c, x40s;  // X[40] = exp(j 40 d) of that symbol
};

// std::arg on digital silence (ref :299) and the one-tap tie census: k_frontend.hip::silence_pd, verbatim logic.
__device__ inline bool tone_tie(double p1, double p2, double p3, double p4) {
    const double x = p1 * p2, y = p3 * p4;
    return (p1 != 0.0 || p2 != 0.0 || p3 != 0.0 || p4 != 0.0) && fabs(y - x) <= 1e-12 * (fabs(x) + fabs(y));
}
__device__ __noinline__ double2 silence_pd_dual(double dr, double di, PrevSums prv, bool dom1, double fo_sum, uint64_t ksym,
                                                double c1, double c2, double c3, double c4) {
    const double pr = dom1 ? prv.a + prv.b : prv.a - prv.b, pi = dom1 ? prv.c - prv.d : prv.c + prv.d;
    const bool dom_zero = (dr == 0.0 && di == 0.0), prev_zero = (pr == 0.0 && pi == 0.0);
    if (dom_zero == prev_zero) return make_double2(0.0, 0.0);
    const double tie = (prev_zero ? tone_tie(c1, c2, c3, c4) : tone_tie(prv.a, prv.b, prv.c, prv.d)) ? 1.0 : 0.0;
    double th = (80.0 * kPi / kFs) * fo_sum;
    th -= kTwoPi * rint(th / kTwoPi);
    double sn, cs;
    sincos(th, &sn, &cs);
    const unsigned q = (unsigned)((dom1 ? (4u - (unsigned)(ksym & 3u)) : (unsigned)(ksym & 3u)) & 3u);
    double er2 = cs, ei2 = sn;
    if (q == 1u) { er2 = -sn; ei2 = cs; }
    else if (q == 2u) { er2 = -cs; ei2 = -sn; }
    else if (q == 3u) { er2 = sn; ei2 = -cs; }
    double vr = dr, vi = di;
    if (dom_zero) {
        const double jr = dom1 ? pi : -pi, ji = dom1 ? -pr : pr;
        vr = jr * prv.x40c - ji * prv.x40s;
        vi = jr * prv.x40s + ji * prv.x40c;
    }
    const double qr = vr * er2 + vi * ei2;
    const double qi = vi * er2 - vr * ei2;
    return make_double2((qr < 0.0 && qi < 0.0) ? kPi : 0.0, tie);
}

}  // namespace

extern __constant__ double kOpvAtanTab[33][10];  // defined with k_frontend.hip (opv_atan2.h)

// ROLE 0 = T (timing loop, tile staging, chunk bookkeeping), ROLE 1 = F (AFC, soft log, tracker-side state)
template <int ROLE>
__device__ __forceinline__ void dual_body(OpvStream& st, OpvGlobalCfg cfg, unsigned char* lds, int lane) {
    constexpr bool kT = ROLE == 0, kF = ROLE == 1;
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned char* ringb = lds;
    const double* atab = reinterpret_cast<const double*>(lds + kTabOff);
    volatile uint32_t* abort_flag = reinterpret_cast<volatile uint32_t*>(lds + kXchgOff + 64);

    // ---- per-lane constants (k_frontend.hip) ------------------------------------------------------
    const double kf = (double)(lane - 10);
    const double kfs = kf * kDeltaPerHz;
    double aE = 0, bE = 0, aO = 0, bO = 0, aL = 0, bL = 0;
    {
        double sn, cs;
        if (lane < 40) { sincospi((double)lane / 80.0, &sn, &cs); aE = cs; bE = -sn; }
        if (lane >= 10 && lane < 50) { sincospi((double)(lane - 10) / 80.0, &sn, &cs); aO = cs; bO = -sn; }
        if (lane >= 20 && lane < 60) { sincospi((double)(lane - 20) / 80.0, &sn, &cs); aL = cs; bL = -sn; }
    }
    SinCosK sck;
    sck.s0 = -0x1.5555555555555p-3; sck.s1 = 0x1.1111111110f73p-7; sck.s2 = -0x1.a01a019da51d6p-13;
    sck.s3 = 0x1.71de256e9bdffp-19; sck.s4 = -0x1.add325df5e3b5p-26;
    sck.c0 = -0x1.0000000000000p-1; sck.c1 = 0x1.5555555555014p-5; sck.c2 = -0x1.6c16c16818f3fp-10;
    sck.c3 = 0x1.a019dfaa26924p-16; sck.c4 = -0x1.276f06eab6283p-22;
    // loop constants parked in VGPRs (k_frontend.hip)
    double kc_tfmax = 0.1, kc_beta = 0.00001, kc_alpha = 0.005, kc_fomax = 2000.0, kc_eps = 1e-10, kc_tiny = 1e-100;
    double kc_halfpi = 1.57079632679489661923, kc_32 = 32.0, kc_m1_32 = -1.0 / 32.0, kc_gain = st.afc_alpha * (kSymRate / kTwoPi);
    asm volatile("" : "+v"(kc_tfmax), "+v"(kc_beta), "+v"(kc_alpha), "+v"(kc_fomax), "+v"(kc_eps), "+v"(kc_tiny));
    asm volatile("" : "+v"(kc_halfpi), "+v"(kc_32), "+v"(kc_m1_32), "+v"(kc_gain));
    const double kc_nfomax = __builtin_canonicalize(-kc_fomax), kc_ntfmax = __builtin_canonicalize(-kc_tfmax);
    kc_fomax = __builtin_canonicalize(kc_fomax);
    kc_tfmax = __builtin_canonicalize(kc_tfmax);
    double sx = 1.0, nsg = 1.0;
    asm volatile("" : "+v"(sx), "+v"(nsg));

    // ---- carry (both waves load the same state) ------------------------------------------------------
    double fo = st.freq_offset, tf = st.timing_freq, mu = st.mu;
    PrevSums qp{st.p1r, st.p1i, st.p2r, st.p2i, st.x40c, st.x40s}, qq{0, 0, 0, 0, 1, 0};
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
    if (cap_soft > (1ull << 28)) overflow = 1;
    uint64_t soft_keep = st.trk_next >= 24 ? st.trk_next - 24 : 0;
    if (st.trk_state != 0 && st.trk_anchor < soft_keep) soft_keep = st.trk_anchor;
    const uint32_t soft_bmask = (uint32_t)(cap_soft * 8u - 1u) & ~7u;
    gbyte* const soft_base = (gbyte*)st.soft;
    const gbyte* iq_bytes = (const gbyte*)st.iq;
    const uint64_t n_bytes = (uint64_t)n_avail * 4u;
    uint32_t seq = 0;                         // symbols of this launch so far: fo(seq) / pos(seq) carry that tag
    bool aborted = false;

    // ---- tile staging: T issues, both keep the event schedule (k_frontend.hip) ---------------------------
    auto glds16 = [&](const gbyte* gsrc, uint32_t lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(uni(lds_byte))
                     : "memory");
    };
    const uint32_t lds_base = uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds);
    auto issue_tile = [&](uint32_t t) {
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
    uint32_t t_lo = (origin >= kBack ? origin - kBack : 0u) / kTile;
    bool evt_issue = true;
    uint32_t next_evt = (t_lo + 1u) * kTile + kBack;

    // ---- hand-over ----------------------------------------------------------------------------------------
    // Writer: value, then tag (two LDS stores of one wave stay in order; every lane stores the same bytes, so no exec
    // masking). Reader: ONE 16-byte read of the slot - if it carries the awaited tag, the value in front of it is the
    // one that was written before that tag.
    auto publish = [&](uint32_t slot_byte, double v, uint32_t tag) {
        const uint32_t a = lds_base + slot_byte + ((tag & 1u) << 4);
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %0, %2 offset:8" : : "v"(a), "v"(v), "v"(tag) : "memory");
    };
    // The poll is one scalar loop in assembly (tag first, then the value: LDS reads of a wave return in order, so a
    // matching tag guarantees the value behind it). It gives up after kPollLimit rounds; `timeouts` is looked at once
    // per batch of symbols, not here.
    uint32_t timeouts = 0;
    auto await = [&](uint32_t slot_byte, uint32_t tag, double& v) {
        const uint32_t a = lds_base + slot_byte + ((tag & 1u) << 4);
        const uint32_t want = uni(tag);
        uint32_t got_v, got_s, cnt;
        double val;
        asm volatile(
            "s_mov_b32 %[cnt], 0\n"
            "1:\n\t"
            "ds_read_b32 %[gv], %[a] offset:8\n\t"
            "ds_read_b64 %[val], %[a]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_readfirstlane_b32 %[gs], %[gv]\n\t"
            "s_cmp_eq_u32 %[gs], %[want]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_add_u32 %[cnt], %[cnt], 1\n\t"
            "s_cmp_lt_u32 %[cnt], %[lim]\n\t"
            "s_cbranch_scc1 1b\n"
            "2:"
            : [gv] "=&v"(got_v), [val] "=&v"(val), [gs] "=&s"(got_s), [cnt] "=&s"(cnt)
            : [a] "v"(a), [want] "s"(want), [lim] "s"(kPollLimit)
            : "memory", "scc");
        timeouts |= (cnt >= kPollLimit) ? 1u : 0u;
        v = val;
    };
    constexpr uint32_t kSlotPos = kXchgOff, kSlotFo = kXchgOff + 32;

    if constexpr (kT) {
        issue_tile(t_lo);
        issue_tile(t_lo + 1u);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): both tiles (and the guard) landed
    }
    if constexpr (kF) publish(kSlotFo, fo, 0u);   // T awaits fo(seq) at every symbol, the launch's first included
    __syncthreads();                          // tiles before F's first tap; atan table, slot tags

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
        if (overflow || aborted) break;
        if ((n_soft - soft_keep) + (uint64_t)(N / 38u + 2u) > cap_soft) { stalled = 1; break; }

        const double Nd = (double)N;
        double pos = mu;                                   // ref :217
        const uint32_t soft_off0 = ((uint32_t)n_soft * 8u) & soft_bmask;
        uint32_t soft_off = soft_off0;
        asm volatile("" : "+v"(soft_off));

        // Tile events for the symbol at `at` and the number of FOLLOWING symbols that need neither an event nor the
        // end-of-call test (k_frontend.hip::housekeeping; both waves keep the schedule, only T moves data)
        auto housekeeping = [&](double at) {
            const uint32_t b = uni((uint32_t)at);
            const uint32_t gb = origin + b;
            while (gb >= next_evt) {
                if (evt_issue) {
                    if constexpr (kT) issue_tile(t_lo + 2u);
                    ++t_lo;
                    evt_issue = false;
                    next_evt = (t_lo + 1u) * kTile - kAhead;
                } else {
                    if constexpr (kT) __builtin_amdgcn_s_waitcnt(0x0F70);
                    evt_issue = true;
                    next_evt = (t_lo + 1u) * kTile + kBack;
                }
            }
            const int lim1 = (int)(next_evt - gb) - 2, lim2 = (int)N - 52 - (int)b;
            int lim = lim1 < lim2 ? lim1 : lim2;
            if (lim < 0) lim = 0;
            return ((uint32_t)lim * 1560u) >> 16;          // <= floor(lim / 42)
        };

        int w0 = 0, w1 = 0;
        double f = 0.0;
        [[maybe_unused]] double nxs = 0.0, nxc = 1.0;      // F: exp(j kf d) of the NEXT symbol, formed while T's position is on its way
        auto fetch = [&](double at, bool clamp0) {
            double p = at + kf;
            if (clamp0) p = fmax(p, 0.0);                  // early gate before the chunk: s[0] (ref :237)
            const int idx = (int)p;
            f = __builtin_amdgcn_fract(p);
            const uint32_t tap_byte = (((uint32_t)idx + origin) << 2) & (kRingBytes - 4u);
            const int* tap = reinterpret_cast<const int*>(ringb + tap_byte);
            w0 = tap[0];
            w1 = tap[1];
        };

        // One symbol for this wave's role (k_frontend.hip::symbol with the other loop's statements removed).
        auto symbol = [&](auto tag, PrevSums& cur, const PrevSums& prv) {
            constexpr bool kFirst = decltype(tag)::first;   // first symbol of a call: no AFC update (ref :289)
            constexpr bool kWide = decltype(tag)::wide;     // fo may still be an unclamped -o value
            // ---- the lane's sample (needs pos only) ------------------------------------------------------
            const int s0r = (int)(short)(w0 & 0xFFFF), s0i = w0 >> 16;      // ref :1023
            const int d_r = (int)(short)(w1 & 0xFFFF) - s0r, d_i = (w1 >> 16) - s0i;
            const double lr = fma(f, (double)d_r, (double)s0r);              // ref :122-128
            const double li = fma(f, (double)d_i, (double)s0i);
            if constexpr (kT) await(kSlotFo, seq, fo);                       // the frequency this symbol runs at
            double xs, xc;
            if constexpr (kWide) {
                if (__builtin_expect(uni_lt(2000.0, fabs(fo)), 0)) sincos(kfs * fo, &xs, &xc);   // an unclamped -o (ref :1004-1005)
                else expj_small(kfs, fo, sck, xs, xc);
            } else if constexpr (kF) {
                xs = nxs; xc = nxc;                                          // formed at the end of the previous symbol
            } else {
                expj_small(kfs, fo, sck, xs, xc);
            }
            const double zr = fma(lr, xc, li * xs);                          // Z = Lam conj(X)
            const double zi = fma(li, xc, -(lr * xs));
            // ---- on-time gate (shared prefix: identical operations in both waves) ------------------------
            double q1 = swap16_add(swap32_add(zr * aO, zi * aO), swap32_add(zi * bO, zr * bO));   // rows: P1, P2, P3, P4 partials
            q1 = dpp_add<0x128>(q1);
            q1 = dpp_add<0x124>(q1);
            q1 = dpp_add<0x122>(q1);
            q1 = dpp_add<0x121>(q1);
            double P2o = readlane_d(q1, 16), P4o = readlane_d(q1, 48);
            const double P1o = readlane_d(q1, 0), P3o = readlane_d(q1, 32);
            asm volatile("" : "+v"(P2o), "+v"(P4o));                        // one scalar source per instruction
            const double s1r_ = P1o + P2o, s1i_ = P3o - P4o;                 // S_1 (tone -13550)
            const double s2r_ = P1o - P2o, s2i_ = P3o + P4o;                 // S_2 (tone +13550)
            const double en1 = fma(s1r_, s1r_, s1i_ * s1i_);                 // ref :264-265
            const double en2 = fma(s2r_, s2r_, s2i_ * s2i_);
            const double soft = en2 - en1;                                   // ref :268
            nsg = mkd((dhi(soft) & (int)0x80000000) | 0x3ff00000, dlo(nsg)); // -1 iff tone 1 dominates (soft < 0)
            const double sg = -nsg;

            if constexpr (kT) {
                // ---- early / late gates of the dominant tone, TED, timing loop (ref :271-286, :313) ----
                const double szi = sg * zi, szr = sg * zr;
                const double wEr = fma(szi, bE, zr * aE), wEi = fma(-szr, bE, zi * aE);
                const double wLr = fma(szi, bL, zr * aL), wLi = fma(-szr, bL, zi * aL);
                double q2 = swap16_add(swap32_add(wEr, wLr), swap32_add(wEi, wLi));   // rows: E.re, E.im, L.re, L.im
                q2 = dpp_add<0x128>(q2);
                q2 = dpp_add<0x124>(q2);
                q2 = dpp_add<0x122>(q2);
                q2 = dpp_add<0x121>(q2);
                const double Eim = readlane_d(q2, 16), Lim = readlane_d(q2, 48);
                const double Ere = readlane_d(q2, 0), Lre = readlane_d(q2, 32);
                const double ee = fma(Ere, Ere, Eim * Eim), el = fma(Lre, Lre, Lim * Lim);
                const double num = el - ee, den = el + ee + kc_eps;          // ted = num / den (ref :275 / :279)
                double y = __builtin_amdgcn_rcp(den);
                y = fma(fma(-den, y, 1.0), y, y);
                y = fma(fma(-den, y, 1.0), y, y);
                double ted = num * y;
                ted = fma(fma(-den, ted, num), y, ted);
                tf = clampd(fma(kc_beta, ted, tf), kc_ntfmax, kc_tfmax);     // ref :283-284
                pos += 40.0 + fma(kc_alpha, ted, tf);                        // ref :285, :313 (the +/-2 clamp of :286 cannot act)
                publish(kSlotPos, pos, seq + 1u);
                fetch(pos, false);                                           // next symbol's taps (speculative at the end of a call)
            }
            if constexpr (kF) {
                *(gdouble*)(soft_base + soft_off) = soft;                    // all lanes, same value and address
                if constexpr (kWide) {
                    if (__builtin_expect(uni_lt(2000.0, fabs(fo)), 0)) sincos((40.0 * kDeltaPerHz) * fo, &cur.x40s, &cur.x40c);
                    else { cur.x40c = readlane_d(xc, 50); cur.x40s = readlane_d(xs, 50); }
                } else {
                    cur.x40c = readlane_d(xc, 50);
                    cur.x40s = readlane_d(xs, 50);
                }
                const double fo_used = fo;
                if constexpr (!kFirst) {
                    // ---- phase detector: arg(dom conj(prev)) (ref :289-299; k_frontend.hip for the algebra) ----
                    const double dr = fma(sg, P2o, P1o), di = fma(-sg, P4o, P3o);
                    const double prs = fma(sg, prv.a, prv.b), pis = fma(sg, prv.c, -prv.d);
                    const double ar = fma(dr, prs, di * pis), ai = fma(di, prs, -(dr * pis));
                    const double cy = fma(ar, prv.x40c, ai * prv.x40s);      // Im z
                    const double cx = fma(ar, prv.x40s, -(ai * prv.x40c));   // Re z
                    const double ax = fabs(cx), ay = fabs(cy);
                    double mx, mn, dm;
                    asm("v_max_f64 %0, |%3|, |%4|\n\tv_min_f64 %1, |%3|, |%4|\n\tv_max_f64 %2, %0, %5"
                        : "=&v"(mx), "=&v"(mn), "=&v"(dm) : "v"(cx), "v"(cy), "v"(kc_tiny));
                    double y = __builtin_amdgcn_rcp(dm);
                    y = fma(fma(-dm, y, 1.0), y, y);
                    y = fma(fma(-dm, y, 1.0), y, y);
                    double ratio = mn * y;
                    ratio = fma(fma(-dm, ratio, mn), y, ratio);
                    const double kd = rint(ratio * kc_32);
                    const int k = (int)kd;
                    const double h = fma(kd, kc_m1_32, ratio);
                    const unsigned char* rowb = reinterpret_cast<const unsigned char*>(atab) + __umul24((unsigned)k, kTabRow * 8u);
                    const double2* trow = reinterpret_cast<const double2*>(rowb);
                    const double c8 = reinterpret_cast<const double*>(rowb)[8];
                    const double2 c67 = trow[3], c45 = trow[2], c23 = trow[1], c01 = trow[0];
                    sx = mkd((dhi(cx) & (int)0x80000000) | 0x3ff00000, dlo(sx));
                    const double pd_off = fma(-sx, kc_halfpi, kc_halfpi);
                    double pd = fma(c8, h, c67.y);
                    pd = fma(pd, h, c67.x);
                    pd = fma(pd, h, c45.y);
                    pd = fma(pd, h, c45.x);
                    pd = fma(pd, h, c23.y);
                    pd = fma(pd, h, c23.x);
                    pd = fma(pd, h, c01.y);
                    pd = fma(pd, h, c01.x);
                    pd = (ay > ax) ? kc_halfpi - pd : pd;
                    pd = fma(sx, pd, pd_off);
                    pd = mkd((dhi(pd) & 0x7fffffff) | (dhi(cy) & (int)0x80000000), dlo(pd));
                    if (__builtin_expect(uni_eq(mx, 0.0), 0)) {              // digital silence on either side
                        const double2 sp = silence_pd_dual(dr, di, prv, soft < 0.0, fo_sum,
                                                           n_soft + (((soft_off - soft_off0) & soft_bmask) >> 3), P1o, P2o, P3o, P4o);
                        pd = sp.x;
                        edge_ties += uni((uint32_t)sp.y);
                    }
                    const double fo_new = fma(kc_gain, pd, fo);              // ref :300-303
                    asm("v_max_f64 %0, %1, %2\n\tv_min_f64 %0, %0, %3" : "=&v"(fo) : "v"(fo_new), "v"(kc_nfomax), "v"(kc_fomax));
                }
                publish(kSlotFo, fo, seq + 1u);
                fo_sum += fo_used;
                cur.a = P1o; cur.b = P2o; cur.c = P3o; cur.d = P4o;          // ref :309-310
            }
            soft_off = (soft_off + 8u) & soft_bmask;
            ++seq;
            if constexpr (kF) {
                expj_small(kfs, fo, sck, nxs, nxc);                          // next symbol's LO factor (fo is final) while T finishes
                await(kSlotPos, seq, pos);                                   // the position this symbol led to
                fetch(pos, false);                                           // (speculative at the end of a call)
            }
        };

        if (uni_lt(pos + 40.0 + 10.0, Nd)) {               // ref :221
            (void)housekeeping(pos);
            fetch(pos, true);
            symbol(TagFirst{}, qp, qp);
            if (__builtin_expect(uni_lt(2000.0, fabs(fo)), 0) && uni_lt(pos + 40.0 + 10.0, Nd)) {
                (void)housekeeping(pos);                   // an out-of-range -o is still in force for one more symbol
                symbol(TagSecond{}, qq, qp);
                qp = qq;
            }
            while (!aborted && uni_lt(pos + 40.0 + 10.0, Nd)) {        // ref :221
                if (__builtin_expect(timeouts != 0u || uni(*abort_flag) != 0u, 0)) {   // the other wave never came: both give up
                    aborted = true;
                    if (lane == 0) *abort_flag = 1u;
                    break;
                }
                uint32_t pairs = uni(housekeeping(pos)) >> 1;
                for (; pairs != 0u; --pairs) {
                    symbol(TagSteady{}, qq, qp);
                    symbol(TagSteady{}, qp, qq);
                }
                symbol(TagSteady{}, qq, qp);
                qp = qq;
            }
        }
        if (aborted) break;

        // ---- end of this demodulate() call (ref :318-328, :1067-1076) ------------------------
        const uint32_t nsym_call = ((soft_off - soft_off0) & soft_bmask) >> 3;
        const uint32_t used = uni((uint32_t)pos);
        mu = pos - (double)used;
        const uint32_t leftover = N - used;
        if (lane == 0) {                                                     // chunk log is a ring; each wave writes what it owns
            double* c = st.chunk_log + 5 * (size_t)(n_chunks % st.cap_chunks);
            if constexpr (kF) c[0] = fo;
            if constexpr (kT) { c[1] = tf; c[2] = mu; c[3] = (double)leftover; c[4] = (double)nsym_call; }
        }
        ++n_chunks;
        n_soft += nsym_call;
        total_samples += N;
        origin += (leftover > 0u && leftover < N) ? used : N;
        if (last) { tail_done = 1; break; }
    }

    if (lane == 0) {
        if constexpr (kT) {
            st.timing_freq = tf; st.mu = mu;
            st.origin = origin; st.n_soft = n_soft; st.total_samples = total_samples;
            st.n_chunks = n_chunks; st.tail_done = tail_done; st.overflow = aborted ? 2 : overflow;
            st.stalled = stalled;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            st.dbg_hw_id = hw; st.dbg_xcc_id = xcc;
            st.dbg_cycles = __builtin_amdgcn_s_memtime() - dbg_t0;
            st.dbg_ticks = __builtin_amdgcn_s_memrealtime() - dbg_r0;
        }
        if constexpr (kF) {
            st.freq_offset = fo;
            st.p1r = qp.a; st.p1i = qp.b; st.p2r = qp.c; st.p2i = qp.d; st.x40c = qp.x40c; st.x40s = qp.x40s;
            st.fo_sum = fo_sum;
            st.edge_ties = edge_ties;
        }
    }
}

// one 128-thread workgroup per stream: wave 0 = T, wave 1 = F (two SIMDs of one CU)
extern "C" __global__ __launch_bounds__(128) void k_msk_frontend_dual(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                       int n_streams) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* atab = reinterpret_cast<double*>(lds + kTabOff);
    for (int i = threadIdx.x; i < 33 * (int)kTabRow; i += 128) atab[i] = (&kOpvAtanTab[0][0])[i];
    if (threadIdx.x < 16) reinterpret_cast<uint32_t*>(lds + kXchgOff)[threadIdx.x] = 0xFFFFFFFFu;   // slot tags: "nothing yet"
    if (threadIdx.x == 16) *reinterpret_cast<uint32_t*>(lds + kXchgOff + 64) = 0u;                    // abort flag
    OpvStream& st = streams[blockIdx.x];
    (void)n_streams;
    if (wave == 0) dual_body<0>(st, cfg, lds, lane);
    else dual_body<1>(st, cfg, lds, lane);
}
