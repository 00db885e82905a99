// k_frontend_x4.hip — MSK front-end for MANY streams: FOUR IQ streams per wavefront, one per DPP row
// (16 lanes), four interpolated samples per lane. Same arithmetic contract as k_frontend.hip
// (reference src/opv-demod.cpp:206-329 + the chunker :1012-1113 / :1132-1173); selected by the shim
// when a context carries enough streams to fill the chip without the one-wave-per-stream mapping
// (opv_capi.hip: opv_set_frontend / automatic from 2049 streams).
//
// Why: a symbol's loop filters, divides and atan2 are scalar work per STREAM. With one stream per
// wave they are executed on 64 lanes for one result (about 65 of that kernel's 161 instructions per
// symbol). Here a wave instruction advances four streams: the scalar tail is shared by four, the
// reductions stay inside a DPP row (4 rotations, no cross-row swaps), and the per-sample work grows only
// from one to four taps per lane. Rows reach their chunk ends, first symbols and refill points at different
// symbols, so the loop carries per-row call state under exec masks; symbols run in BATCHES that provably need none
// of it for any row (round 2: the same statements compiled without the tests), the rings are refilled in 256-sample
// blocks by the whole wave, the row sums run four in lockstep and finish with broadcast FMACs; together they took the
// per-wave-symbol count (rocprofv3 PMC, MI355X, 4096 streams) from 448 VALU + 87 SALU + 12 LDS/VMEM to 319 + 25 + 7 =
// 88 issued instructions per symbol and stream (one wave per stream: 161). Because a wave carries four streams the chip fills four times later, and a launch lasts as long as one wave
// needs for its four streams: 47 ms for 30 frames whether the context has 1025 or 4096 streams (four waves per
// workgroup = one per SIMD of a CU by construction, see msk_frontend_x4_body), 81 ms for 8192 (two waves per SIMD):
// front-end alone 228 GS/s at 4096 streams, 264 at 8192 (round 1: 87 / 130). The one-wave kernel runs 1024 streams at a
// time in 23 ms per 30 frames: faster up to 2048 streams, slower from 2049 on, which is where the shim switches
// (NOTEBOOK.md §3.1).
//
// Mapping (row r = lane / 16 serves stream 4*blockIdx.x + r, t = lane % 16):
//   * lane t owns the interpolated samples Lam_j = L(pos + j - 10), j = t + 16 q, q = 0..3 (j < 60); the
//     three gates are j in [0,40) / [10,50) / [20,60) as in k_frontend.hip, their LO constants T[i] zero
//     outside the window, so a lane accumulates its (up to) four taps per gate in registers;
//   * X[m] = exp(j m d) for m = t - 10 by the same polynomial, X[m+16 q] by three complex multiplies
//     with X[16];
//   * every stream-level quantity (pos, fo, tf, previous sums, chunk bookkeeping) lives in VGPRs,
//     replicated over the 16 lanes of its row; rows run their own chunk schedule under exec masks;
//   * int16 IQ: a 1024-sample ring per row in LDS (+ 60-sample guard mirroring its head), refilled in
//     256-sample blocks (one direct-to-LDS 16 B/lane load of the WHOLE wave per row and block), requested
//     one refill point (four symbols) before their first use and awaited with one s_waitcnt vmcnt(0) there.
//
// Differences from the reference are of the same kind and size as k_frontend.hip's (shared
// interpolation fraction, factored LO, FMA, table atan2): soft symbols agree to ~1e-14 of their mean,
// every decision downstream is identical (tests/test_gpu_parity.py runs both mappings).
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

constexpr uint32_t kRingSamples = 1024;
constexpr uint32_t kRingBytes = kRingSamples * 4;   // 4096
constexpr uint32_t kGuardBytes = 240;               // mirror of the ring's first 60 samples (a tap reaches 200 B past the ring; 240 keeps
                                                    // two 16-stream workgroups with their 12 KB angle table inside a CU's 160 KB)
constexpr uint32_t kRowBytes = kRingBytes + kGuardBytes;
constexpr uint32_t kBlock = 256;                    // samples per refill block (64 lanes x 16 B: the WHOLE wave loads for one row)
// Refill rule, applied every fourth symbol after the previous blocks have landed (g = floor(pos) of the row, hi = end of
// what its ring holds): a symbol reads samples g - 11 .. g + 55 and g grows by at most 42 per symbol, so the four symbols
// up to the next refill point need hi >= g + 182 NOW (invariant) and the ones after it hi >= g + 350 THEN. A block is
// requested while hi < g + 648: it lands by the next refill point, where hi + 256 >= (g + 168) + 182 again, and it
// overwrites samples below hi - 768 <= g - 120, which nothing reads any more. At most one block per row and refill point.
constexpr uint32_t kAheadMin = 648;
constexpr uint32_t kTabOff = 4 * kRowBytes;         // 17408
// LDS per workgroup: WPB x kTabOff + the atan table (257 x 48 B) = 29 680 B for one wave (five workgroups per CU), 81 712 B for four (two)
static_assert(kTabOff % 16 == 0, "16-byte LDS alignment");

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) unsigned char gbyte;

__device__ inline int dlo(double v) { return __double2loint(v); }
__device__ inline int dhi(double v) { return __double2hiint(v); }
__device__ inline double mkd(int hi, int lo) { return __hiloint2double(hi, lo); }

// four row sums in lockstep: a DPP read needs two wait states behind the VALU write of its source, and one sum's rotation
// steps are a dependent chain (hipcc pads every step with s_nop 1) - the other three sums' instructions fill the slots
template <int CTRL>
__device__ inline void dpp_add4(double& a, double& b, double& c, double& d) {
    const int al = __builtin_amdgcn_mov_dpp(dlo(a), CTRL, 0xF, 0xF, true), ah = __builtin_amdgcn_mov_dpp(dhi(a), CTRL, 0xF, 0xF, true);
    const int bl = __builtin_amdgcn_mov_dpp(dlo(b), CTRL, 0xF, 0xF, true), bh = __builtin_amdgcn_mov_dpp(dhi(b), CTRL, 0xF, 0xF, true);
    const int cl = __builtin_amdgcn_mov_dpp(dlo(c), CTRL, 0xF, 0xF, true), ch = __builtin_amdgcn_mov_dpp(dhi(c), CTRL, 0xF, 0xF, true);
    const int dl = __builtin_amdgcn_mov_dpp(dlo(d), CTRL, 0xF, 0xF, true), dh = __builtin_amdgcn_mov_dpp(dhi(d), CTRL, 0xF, 0xF, true);
    __builtin_amdgcn_sched_barrier(0);
    a += mkd(ah, al); b += mkd(bh, bl); c += mkd(ch, cl); d += mkd(dh, dl);
    __builtin_amdgcn_sched_barrier(0);
}
// Two rotation steps leave the row's four partial sums in its lanes 0..3 (every lane l holds the one of l mod 4); the
// DP-ALU DPP forms finish the job in 16 instructions instead of the 24 of two more rotation steps: v_mov_b64_dpp
// row_newbcast:0 + three v_fmac_f64_dpp row_newbcast:n with a factor of 1.0 per value (k_frontend.hip: symbol_r,
// scripts/microbench/dpp64.hip). The four adds of the step before are the wait states the first DPP reads need.
__device__ inline void row_sum4(double& a, double& b, double& c, double& d, double one) {
    dpp_add4<0x128>(a, b, c, d);
    dpp_add4<0x124>(a, b, c, d);
    double ra, rb, rc, rd;
#define OPV_B4(N) "v_fmac_f64_dpp %0, %4, %8 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %1, %5, %8 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %2, %6, %8 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t" \
                  "v_fmac_f64_dpp %3, %7, %8 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm("v_mov_b64_dpp %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %6 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %3, %7 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        OPV_B4(1) OPV_B4(2) OPV_B4(3)
        : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(one));
#undef OPV_B4
    a = ra; b = rb; c = rc; d = rd;
}
__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// exp(j x), |x| <= 0.284: same near-minimax pair as k_frontend.hip (abs error 1e-19 / 1.3e-18)
__device__ inline void expj_small(double x, double& xs, double& xc) {
    const double u = x * x;
    double p = fma(-0x1.add325df5e3b5p-26, u, 0x1.71de256e9bdffp-19);
    double r = fma(-0x1.276f06eab6283p-22, u, 0x1.a019dfaa26924p-16);
    p = fma(p, u, -0x1.a01a019da51d6p-13);
    r = fma(r, u, -0x1.6c16c16818f3fp-10);
    p = fma(p, u, 0x1.1111111110f73p-7);
    r = fma(r, u, 0x1.5555555555014p-5);
    p = fma(p, u, -0x1.5555555555555p-3);
    r = fma(r, u, -0x1.0000000000000p-1);
    xc = fma(r, u, 1.0);
    xs = fma(x * u, p, x);
}

struct PrevSums {
    double a, b, c, d;  // on-time P1..P4
    double x40c, x40s;  // X[40] = exp(j 40 d) of that symbol
};

// std::arg on digital silence (ref :299): see k_frontend.hip::silence_pd for the derivation.
// `ties` counts the windows with exactly one non-zero tap (opv_stream_state.edge_ties), as there.
__device__ inline bool tone_tie(double p1, double p2, double p3, double p4) {
    const double x = p1 * p2, y = p3 * p4;
    return (p1 != 0.0 || p2 != 0.0 || p3 != 0.0 || p4 != 0.0) && fabs(y - x) <= 1e-12 * (fabs(x) + fabs(y));
}
__device__ __noinline__ double2 silence_pd_x4(double dr, double di, double pa, double pb, double pc, double pd_, double x40c,
                                               double x40s, bool dom1, double fo_sum, uint32_t ksym,
                                               double c1, double c2, double c3, double c4) {
    const double pr = dom1 ? pa + pb : pa - pb, pi = dom1 ? pc - pd_ : pc + pd_;
    const bool dom_zero = (dr == 0.0 && di == 0.0), prev_zero = (pr == 0.0 && pi == 0.0);
    if (dom_zero == prev_zero) return make_double2(0.0, 0.0);
    const double tie = (prev_zero ? tone_tie(c1, c2, c3, c4) : tone_tie(pa, pb, pc, pd_)) ? 1.0 : 0.0;
    double th = (80.0 * kPi / kFs) * fo_sum;
    th -= kTwoPi * rint(th / kTwoPi);
    double sn, cs;
    sincos(th, &sn, &cs);
    const unsigned q = (unsigned)((dom1 ? (4u - (ksym & 3u)) : (ksym & 3u)) & 3u);
    double er2 = cs, ei2 = sn;
    if (q == 1u) { er2 = -sn; ei2 = cs; }
    else if (q == 2u) { er2 = -cs; ei2 = -sn; }
    else if (q == 3u) { er2 = sn; ei2 = -cs; }
    double vr = dr, vi = di;
    if (dom_zero) {
        const double jr = dom1 ? pi : -pi, ji = dom1 ? -pr : pr;
        vr = jr * x40c - ji * x40s;
        vi = jr * x40s + ji * x40c;
    }
    const double qr = vr * er2 + vi * ei2;
    const double qi = vi * er2 - vr * ei2;
    return make_double2((qr < 0.0 && qi < 0.0) ? kPi : 0.0, tie);
}

}  // namespace

// this translation unit's own image of the angle table (opv_atan2.h: kOpvAtanTabQ): every .hip file is compiled to a
// code object of its own (no relocatable device code), so that the two front-end files can go through tools/align_vop3.py
__constant__ double kOpvAtanTabQx4[257][6] = {
#include "opv_atan_table_q.inc"
};

// WPB = wavefronts per workgroup (k_frontend.hip, msk_frontend_body: single-wave workgroups are placed without regard
// to SIMDs, four waves of one workgroup always land on the four SIMDs of a CU). Waves share only the atan table.
template <int WPB>
__device__ __forceinline__ void msk_frontend_x4_body(OpvStream* __restrict__ streams, OpvGlobalCfg cfg, int n_streams) {
    const int lane = threadIdx.x & 63, row = lane >> 4, t = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sidx = ((int)blockIdx.x * WPB + wave) * 4 + row;
    const bool have = sidx < n_streams;
    OpvStream& st = streams[have ? sidx : n_streams - 1];   // idle rows read a valid record and never write
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();

    __shared__ __attribute__((aligned(16))) unsigned char lds_all[WPB * kTabOff + 257 * 48];
    unsigned char* const lds = lds_all + wave * kTabOff;    // this wave's four rings
    double* atab = reinterpret_cast<double*>(lds_all + WPB * kTabOff);
    for (int i = threadIdx.x; i < 257 * 6; i += 64 * WPB) atab[i] = (&kOpvAtanTabQx4[0][0])[i];
    const unsigned char* ring = lds + (uint32_t)row * kRowBytes;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t ring_lds = lds_base + (uint32_t)row * kRowBytes;

    // ---- per-lane constants: T_1[i] = (cos(pi i/80), -sin(pi i/80)) inside each gate's window ----
    double aE[4], bE[4], aO[4], bO[4], aL[4], bL[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = t + 16 * q;
        double sn, cs;
        aE[q] = bE[q] = aO[q] = bO[q] = aL[q] = bL[q] = 0.0;
        if (j < 40) { sincospi((double)j / 80.0, &sn, &cs); aE[q] = cs; bE[q] = -sn; }
        if (j >= 10 && j < 50) { sincospi((double)(j - 10) / 80.0, &sn, &cs); aO[q] = cs; bO[q] = -sn; }
        if (j >= 20 && j < 60) { sincospi((double)(j - 20) / 80.0, &sn, &cs); aL[q] = cs; bL[q] = -sn; }
    }
    const double kf0 = (double)(t - 10);
    const double kfs0 = kf0 * kDeltaPerHz;
    const double kgain = st.afc_alpha * (kSymRate / kTwoPi);
    double kc_one = 1.0;                                // the broadcast FMACs' second factor has to be a VGPR
    asm volatile("" : "+v"(kc_one));

    // ---- carry (row-uniform, in VGPRs) ----------------------------------------------------------
    double fo = st.freq_offset, tf = st.timing_freq, mu = st.mu, fo_sum = st.fo_sum;
    PrevSums pv{st.p1r, st.p1i, st.p2r, st.p2i, st.x40c, st.x40s};
    uint32_t origin = (uint32_t)st.origin;
    const uint32_t n_avail = (uint32_t)st.n_avail;
    uint64_t n_soft = st.n_soft, total_samples = st.total_samples;
    uint32_t n_chunks = st.n_chunks;
    int tail_done = st.tail_done, overflow = st.overflow, stalled = 0;
    uint32_t edge_ties = st.edge_ties;
    const int eof = st.eof;
    const uint64_t cap_soft = st.cap_soft;
    if (cap_soft > (1ull << 28)) overflow = 1;
    uint64_t soft_keep = st.trk_next >= 24 ? st.trk_next - 24 : 0;
    if (st.trk_state != 0 && st.trk_anchor < soft_keep) soft_keep = st.trk_anchor;
    const uint32_t soft_bmask = (uint32_t)(cap_soft * 8u - 1u) & ~7u;
    gbyte* const soft_base = (gbyte*)st.soft;
    const gbyte* const iq_bytes = (const gbyte*)st.iq;
    const uint64_t n_bytes = (uint64_t)n_avail * 4u;
    double* const chunk_log = st.chunk_log;
    const uint32_t cap_chunks = st.cap_chunks;

    // ---- call state ---------------------------------------------------------------------------------
    bool done = !have, in_call = false, first = false, last = false;
    uint32_t N = 0, soft_off = 0, soft_off0 = 0;
    double Nd = 0.0, pos = 0.0;

    // ---- ring refill ------------------------------------------------------------------------------
    // hi: the row's ring holds absolute samples [hi - 1024, hi) (as far as the capture reaches); blocks of 256 samples,
    // 1 KB aligned in the capture, moved by ONE direct-to-LDS load of the whole wave (lane l writes LDS byte m0 + 16 l):
    // all 64 lanes load for one row at a time, from that row's capture (its pointer and cursor broadcast by v_readlane).
    // The ring head is mirrored into the guard by the row's own 16 lanes (64 samples).
    auto glds16 = [&](const gbyte* gsrc, uint32_t m0v) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(m0v)
                     : "memory");
    };
    uint32_t hi;
    {
        const uint32_t g0 = origin + (uint32_t)(int)mu;
        hi = (g0 >= 11u ? g0 - 11u : 0u) & ~(kBlock - 1u);
    }
    auto issue_block = [&](uint32_t dst_off) {   // the calling lanes are the 16 lanes of ONE row
        const uint64_t off = (uint64_t)hi * 4u + (uint32_t)t * 16u;
        const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ring_lds + dst_off - 256u * (uint32_t)row));
        if (off + 16u <= n_bytes) glds16(iq_bytes + off, m0v);
        else if (off < n_bytes) {   // the capture's last, incomplete 16 bytes: nothing past n_avail is read
            for (uint32_t j = 0; off + 4u * j < n_bytes; ++j)
                *reinterpret_cast<int*>(lds + (uint32_t)row * kRowBytes + dst_off + (uint32_t)t * 16u + 4u * j) =
                    *reinterpret_cast<const __attribute__((address_space(1))) int*>(iq_bytes + off + 4u * j);
        }
    };
    auto issue_wide = [&](int r, uint32_t hi_r) {   // all 64 lanes; r is a constant after unrolling
        const uint32_t nb_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)n_bytes, 16 * r);
        const uint32_t nb_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(n_bytes >> 32), 16 * r);
        const uint64_t nb = ((uint64_t)nb_hi << 32) | nb_lo;
        const uint64_t pb = (uint64_t)(uintptr_t)iq_bytes;
        const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pb, 16 * r);
        const uint32_t p_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pb >> 32), 16 * r);
        const gbyte* src = (const gbyte*)(uintptr_t)(((uint64_t)p_hi << 32) | p_lo);
        const uint32_t dst = (hi_r * 4u) & (kRingBytes - 1u);
        const uint64_t off = (uint64_t)hi_r * 4u + (uint32_t)lane * 16u;
        if (off + 16u <= nb) glds16(src + off, lds_base + (uint32_t)r * kRowBytes + dst);
        else if (off < nb) {   // the capture's last, incomplete 16 bytes: nothing past n_avail is read
            for (uint32_t j = 0; off + 4u * j < nb; ++j)
                *reinterpret_cast<int*>(lds + (uint32_t)r * kRowBytes + dst + (uint32_t)lane * 16u + 4u * j) =
                    *reinterpret_cast<const __attribute__((address_space(1))) int*>(src + off + 4u * j);
        }
    };
    auto refill = [&](bool wants, uint32_t g, int max_rounds) {
        for (int rep = 0; rep < max_rounds; ++rep) {
            const bool need = wants && hi < g + kAheadMin && (uint64_t)hi * 4u < n_bytes;
            const uint64_t m = __ballot(need);
            if (m == 0ull) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((m >> (16 * r)) & 1ull) {      // wave-uniform
                    const uint32_t hi_r = (uint32_t)__builtin_amdgcn_readlane((int)hi, 16 * r);
                    issue_wide(r, hi_r);
                    if (((hi_r * 4u) & (kRingBytes - 1u)) == 0u) {   // ring head: mirror its first 64 samples into the guard
                        if (row == r && t < 15) issue_block(kRingBytes);   // 15 lanes x 16 B = the guard's 240 B
                    }
                }
            }
            if (need) hi += kBlock;
        }
    };
    // Soft symbols are written four at a time: lane t < 4 of a row keeps the value of the symbol with
    // iter % 4 == t and stores it at the next refill point, right AFTER that point's s_waitcnt - a store
    // per symbol would put a fresh store in front of every vmcnt(0) and make the wave wait out its latency.
    double held = 0.0;
    uint32_t held_off = 0;
    bool held_valid = false;
    auto flush_soft = [&]() {
        if (held_valid) *(gdouble*)(soft_base + held_off) = held;
        held_valid = false;
    };
    // One symbol of every row that executes this (exec = the rows inside a demodulate() call whose next symbol exists).
    // Generic: with the tests the first symbols of a call need (early gate before the chunk, no AFC on the first symbol,
    // an out-of-range -o still in force). Fast: the same statements without them - bit-identical where both apply
    // (no contraction, no re-association) - for the batches below. `slot`: which of a row's four soft-log lanes keeps
    // this symbol's value until the next flush.
    auto symbol_body = [&](auto generic_tag, uint32_t slot) {
        constexpr bool kGeneric = decltype(generic_tag)::value;
        // ---- taps (ref :122-128, :232-238) --------------------------------------------------
        const double pf = pos + kf0;
        const double fl = floor(pf);
        const double f = pf - fl;
        const int i0 = (int)fl;
        const uint32_t byte0 = (((uint32_t)(i0 + (int)origin)) << 2) & (kRingBytes - 1u);
        int w0[4], w1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int* tap = reinterpret_cast<const int*>(ring + byte0 + 64u * (uint32_t)q);
            w0[q] = tap[0];
            w1[q] = tap[1];
        }
        __builtin_amdgcn_sched_barrier(0);                              // taps requested FIRST, the LO under their latency
        if (kGeneric && first && pf < 0.0) {                           // early gate before the chunk: s[0] (ref :237)
            const int s0 = *reinterpret_cast<const int*>(ring + ((origin << 2) & (kRingBytes - 1u)));
            w0[0] = s0;
            w1[0] = s0;
        }
        // ---- LO: X[m] for m = t - 10 + 16 q ---------------------------------------------------
        double xs[4], xc[4], s16, c16;
        expj_small(kfs0 * fo, xs[0], xc[0]);
        expj_small((16.0 * kDeltaPerHz) * fo, s16, c16);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            xc[q] = fma(xc[q - 1], c16, -(xs[q - 1] * s16));
            xs[q] = fma(xc[q - 1], s16, xs[q - 1] * c16);
        }
        if (kGeneric && fabs(fo) > 2000.0) {
            // -o takes any value (ref :1004-1005) and the AFC clamp (:303) first acts at the END of the
            // call's second symbol: outside the polynomial's range those symbols take the full-range routine
#pragma unroll
            for (int q = 0; q < 4; ++q) sincos((kfs0 + (16.0 * q) * kDeltaPerHz) * fo, &xs[q], &xc[q]);
        }
        // X[40] = exp(j 40 d), needed by the NEXT symbol's phase detector: it is lane 2's fourth tap (m = 2 - 10 + 48),
        // handed to the row by v_mov_b64_dpp row_newbcast:2 (`old` operands: the two dead X[16] registers)
        const double x40c = __builtin_amdgcn_update_dpp(c16, xc[3], 0x152, 0xF, 0xF, false);
        const double x40s = __builtin_amdgcn_update_dpp(s16, xs[3], 0x152, 0xF, 0xF, false);
        // (the LO above does not depend on the taps: it stays between their LDS reads and their first use - left to
        // itself hipcc unpacks the taps first and waits for them)
        asm volatile("" : "+v"(xs[3]), "+v"(xc[3]));
        __builtin_amdgcn_sched_barrier(0);

        double o1 = 0, o2 = 0, o3 = 0, o4 = 0;             // on-time P1..P4 partials
        double eA = 0, eB = 0, eC = 0, eD = 0, lA = 0, lB = 0, lC = 0, lD = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s0r = (int)(short)(w0[q] & 0xFFFF), s0i = w0[q] >> 16;    // ref :1023
            const int d_r = (int)(short)(w1[q] & 0xFFFF) - s0r, d_i = (w1[q] >> 16) - s0i;
            const double lr = fma(f, (double)d_r, (double)s0r);
            const double li = fma(f, (double)d_i, (double)s0i);
            const double zr = fma(lr, xc[q], li * xs[q]);   // Z = Lam conj(X)
            const double zi = fma(li, xc[q], -(lr * xs[q]));
            o1 = fma(zr, aO[q], o1); o2 = fma(zi, bO[q], o2); o3 = fma(zi, aO[q], o3); o4 = fma(zr, bO[q], o4);
            if (q < 3) { eA = fma(zr, aE[q], eA); eB = fma(zi, aE[q], eB); eC = fma(zi, bE[q], eC); eD = fma(zr, bE[q], eD); }
            if (q > 0) { lA = fma(zr, aL[q], lA); lB = fma(zi, aL[q], lB); lC = fma(zi, bL[q], lC); lD = fma(zr, bL[q], lD); }
        }
        // ---- on-time gate: soft value, dominant tone (ref :264-272) --------------------------
        row_sum4(o1, o2, o3, o4, kc_one);
        const double P1o = o1, P2o = o2, P3o = o3, P4o = o4;
        const double s1r_ = P1o + P2o, s1i_ = P3o - P4o;
        const double s2r_ = P1o - P2o, s2i_ = P3o + P4o;
        const double en1 = fma(s1r_, s1r_, s1i_ * s1i_);
        const double en2 = fma(s2r_, s2r_, s2i_ * s2i_);
        const double soft = en2 - en1;                      // ref :268
        const double nsg = mkd((dhi(soft) & (int)0x80000000) | 0x3ff00000, 0);  // -1 iff tone 1 dominates
        const double sg = -nsg;
        // ---- early / late gates of the dominant tone (ref :271-280) ---------------------------
        double Ere = fma(sg, eC, eA), Eim = fma(-sg, eD, eB), Lre = fma(sg, lC, lA), Lim = fma(-sg, lD, lB);
        row_sum4(Ere, Eim, Lre, Lim, kc_one);
        const double ee = fma(Ere, Ere, Eim * Eim), el = fma(Lre, Lre, Lim * Lim);
        const double num = el - ee, den = el + ee + 1e-10;
        // ---- phase detector operands: dom * conj(prev) (ref :289-299, see k_frontend.hip) -----
        const double dr = fma(sg, P2o, P1o), di = fma(-sg, P4o, P3o);
        const double prs = fma(sg, pv.a, pv.b), pis = fma(sg, pv.c, -pv.d);
        const double ar = fma(dr, prs, di * pis), ai = fma(di, prs, -(dr * pis));
        const double cy = fma(ar, pv.x40c, ai * pv.x40s);   // Im z
        const double cx = fma(ar, pv.x40s, -(ai * pv.x40c)); // Re z
        // the angle without an octant fix-up (opv_atan2.h: opv_atan2_q): atan(|cy| / |cx|) = pi/4 + atan(q),
        // q = (|cy| - |cx|) / (|cy| + |cx|) in [-1, 1]
        const double sum = fabs(cx) + fabs(cy), dif = fabs(cy) - fabs(cx);
        // ---- the two divides on one reciprocal ------------------------------------------------
        const double dm = sum + 1e-100;                     // the guard against digital silence: IS sum unless sum is 0 (k_frontend.hip)
        const double tt = den * dm;
        double y = __builtin_amdgcn_rcp(tt);
        y = fma(fma(-tt, y, 1.0), y, y);                    // one Newton step (2^-24.4 -> 2^-48.7, scripts/microbench/rcp_accuracy.hip)
        const double iden = y * dm, idm = y * den;
        const double ratio = dif * idm;                     // good to 2^-48: 3.5e-15 rad on the angle
        // the angle's table row is requested here and used after the timing loop: with one wave per SIMD nothing else
        // covers the LDS round trip (the row index is in range on every path: |ratio| <= 1)
        // nearest expansion point k/128 by the 1.5 * 2^52 trick: the sum's low word is the row index k + 128
        const double kt = fma(ratio, 128.0, 6755399441055744.0 + 128.0);
        const double h = fma(kt - (6755399441055744.0 + 128.0), -1.0 / 128.0, ratio);   // |h| <= 1/256
        const double2* trow = reinterpret_cast<const double2*>(atab + (unsigned)dlo(kt) * 6u);
        const double2 c45 = trow[2], c23 = trow[1], c01 = trow[0];
        __builtin_amdgcn_sched_barrier(0);
        double ted = num * iden;
        ted = fma(fma(-den, ted, num), iden, ted);
        // ---- timing loop (ref :283-286, :313) ------------------------------------------------
        tf = clampd(fma(0.00001, ted, tf), -0.1, 0.1);
        const double adj = fma(0.005, ted, tf);   // |adj| <= 0.105: the reference's clamp to +/-2 (:286) cannot act, see k_frontend.hip
        double pos_next = pos + (40.0 + adj);
        if ((uint32_t)t == slot) { held = soft; held_off = soft_off; held_valid = true; }
        asm volatile("" : "+v"(pos_next), "+v"(tf), "+v"(held));   // (keeps these statements HERE: hipcc otherwise sinks them below the AFC block)
        __builtin_amdgcn_sched_barrier(0);
        // ---- AFC (ref :289-306): not on the first symbol of a call -------------------------------
        if (!kGeneric || !first) {
            double pd = fma(c45.y, h, c45.x);                   // degree 5: pi/4 + atan(q)
            pd = fma(pd, h, c23.y);
            pd = fma(pd, h, c23.x);
            pd = fma(pd, h, c01.y);
            pd = fma(pd, h, c01.x);
            const double sx = mkd((dhi(cx) & (int)0x80000000) | 0x3ff00000, 0);
            pd = fma(sx, pd, fma(-sx, 1.57079632679489661923, 1.57079632679489661923));
            pd = mkd((dhi(pd) & 0x7fffffff) | (dhi(cy) & (int)0x80000000), dlo(pd));
            if (sum == 0.0) {                                // digital silence on either side
                const double2 sp = silence_pd_x4(dr, di, pv.a, pv.b, pv.c, pv.d, pv.x40c, pv.x40s, soft < 0.0, fo_sum,
                                                 (uint32_t)n_soft + (((soft_off - soft_off0) & soft_bmask) >> 3),
                                                 P1o, P2o, P3o, P4o);
                pd = sp.x;
                edge_ties += (uint32_t)sp.y;
            }
            const double fo_used = fo;
            fo = clampd(fma(kgain, pd, fo), -2000.0, 2000.0);
            fo_sum += fo_used;
        } else {
            fo_sum += fo;
        }
        soft_off = (soft_off + 8u) & soft_bmask;
        pv.a = P1o; pv.b = P2o; pv.c = P3o; pv.d = P4o; pv.x40c = x40c; pv.x40s = x40s;
        pos = pos_next;
        first = false;
    };
    refill(!done, origin + (uint32_t)(int)mu, 4);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();                     // atan table visible (single wave: LDS ordering only)

    for (uint32_t iter = 0;; ++iter) {
        // ---- which demodulate() call comes next (ref :1026 / :1088 / :1173) ---------------------
        if (!in_call && !done) {
            const uint32_t remaining = n_avail - origin;
            bool go = true;
            last = false;
            if (cfg.streaming) {
                if (remaining >= OPV_CHUNK) N = OPV_CHUNK;
                else if (eof && !tail_done && remaining > 0) { N = remaining; last = true; }
                else { if (eof) tail_done = 1; go = false; }
            } else {
                if (!eof || tail_done) go = false;
                else { N = n_avail; last = true; }
            }
            if (go && overflow) go = false;
            if (go && (n_soft - soft_keep) + (uint64_t)(N / 38u + 2u) > cap_soft) { stalled = 1; go = false; }  // back-pressure, see k_frontend.hip
            if (go) {
                in_call = true;
                first = true;
                Nd = (double)N;
                pos = mu;                                          // ref :217
                soft_off0 = ((uint32_t)n_soft * 8u) & soft_bmask;
                soft_off = soft_off0;
            } else {
                done = true;
            }
        }
        if (__ballot(in_call) == 0ull) break;

        // ---- batches: as many symbols as EVERY row inside a call can take without its end-of-call test, its first-symbol
        // rules or an out-of-range -o (pos advances by at most 42 samples per symbol), in groups of four (the soft-log
        // lanes and the refill points keep their rhythm); the per-symbol bookkeeping of the loop below - what makes up a
        // third of its instructions - is then paid once per batch. Rows outside a call are finished streams here (a row
        // that could start a call has just done so and asks for 0): they sit the batch out under the exec mask.
        if ((iter & 3u) == 0u) {
            int krow = 0x7fffffff;
            if (in_call) {
                krow = 0;
                const double room = Nd - 51.0 - pos;
                if (!first && !(fabs(fo) > 2000.0) && room > 0.0) krow = (int)(room * (1.0 / 42.0));
            }
            int kmin = __builtin_amdgcn_readlane(krow, 0);
            { const int k1 = __builtin_amdgcn_readlane(krow, 16); kmin = k1 < kmin ? k1 : kmin; }
            { const int k2 = __builtin_amdgcn_readlane(krow, 32); kmin = k2 < kmin ? k2 : kmin; }
            { const int k3 = __builtin_amdgcn_readlane(krow, 48); kmin = k3 < kmin ? k3 : kmin; }
            for (uint32_t quads = (uint32_t)kmin >> 2; quads != 0u; --quads) {
                __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): blocks and stores issued 4 symbols ago
                flush_soft();
                refill(in_call, origin + (uint32_t)(int)pos, 4);
                if (in_call) {
                    symbol_body(std::false_type{}, 0u);
                    symbol_body(std::false_type{}, 1u);
                    symbol_body(std::false_type{}, 2u);
                    symbol_body(std::false_type{}, 3u);
                }
                iter += 4u;
            }
        }

        if ((iter & 3u) == 0u) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): blocks and stores issued 4 symbols ago
            flush_soft();
            refill(in_call, origin + (uint32_t)(int)pos, 4);
        }

        if (in_call) {
            if (pos + 40.0 + 10.0 < Nd) {                          // ref :221
                symbol_body(std::true_type{}, iter & 3u);
            } else {
                // ---- end of this demodulate() call (ref :318-328, :1067-1076) ---------------------
                const uint32_t nsym_call = ((soft_off - soft_off0) & soft_bmask) >> 3;
                const uint32_t used = (uint32_t)pos;
                mu = pos - (double)used;
                const uint32_t leftover = N - used;
                if (t == 0) {
                    double* c = chunk_log + 5 * (size_t)(n_chunks % cap_chunks);
                    c[0] = fo; c[1] = tf; c[2] = mu; c[3] = (double)leftover; c[4] = (double)nsym_call;
                }
                ++n_chunks;
                n_soft += nsym_call;
                total_samples += N;
                origin += (leftover > 0u && leftover < N) ? used : N;
                in_call = false;
                if (last) { tail_done = 1; done = true; }
            }
        }
    }

    flush_soft();
    if (have && t == 0) {
        st.freq_offset = fo; st.timing_freq = tf; st.mu = mu;
        st.p1r = pv.a; st.p1i = pv.b; st.p2r = pv.c; st.p2i = pv.d; st.x40c = pv.x40c; st.x40s = pv.x40s;
        st.fo_sum = fo_sum;
        st.origin = origin; st.n_soft = n_soft; st.total_samples = total_samples;
        st.n_chunks = n_chunks; st.tail_done = tail_done; st.overflow = overflow;
        st.stalled = stalled; st.edge_ties = edge_ties;
        // where and at which clock the wave that carried this stream (and three others) ran (opv_tap_wave_info)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st.dbg_hw_id = hw; st.dbg_xcc_id = xcc;
        st.dbg_cycles = __builtin_amdgcn_s_memtime() - dbg_t0;
        st.dbg_ticks = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
}

// sixteen streams per workgroup: one wave per SIMD of a CU by construction. (A one-wave workgroup shape of this body existed
// until round 6; it was reachable only by forcing this mapping beyond 8192 streams, where the automatic choice is sixteen per wave.)
extern "C" __global__ __launch_bounds__(256) void k_msk_frontend_x4_wg4(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                        int n_streams) {
    msk_frontend_x4_body<4>(streams, cfg, n_streams);
}
