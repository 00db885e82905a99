// k_frontend_x16.hip — MSK front-end for VERY MANY streams: SIXTEEN IQ streams per wavefront, one per DPP quad
// (4 lanes), fifteen CONSECUTIVE interpolated samples per lane. Same arithmetic contract as k_frontend.hip /
// k_frontend_x4.hip (reference src/opv-demod.cpp:206-329 + the chunker :1012-1113 / :1132-1173); selected with
// opv_set_frontend(ctx, 16), and by opv_process itself above 8192 streams per context (opv_capi.hip: kFrontendX16MinStreams;
// bench.py's stream_sweep, front-end alone: 4096 x 15 frames 207 GS/s four-per-wave against 129 sixteen-per-wave - 256 waves
// leave three SIMDs in four idle -, 8192 x 7: 221 against 255, 16 384 x 3: 175 against 407; see the constant's comment).
//
// Why: the per-stream arithmetic of a symbol is ~1080 lane-FMAs, everything else - loop filters,
// divides, atan2, chunk bookkeeping - is scalar work per STREAM that a wave executes on all of its lanes. With four
// streams per wave (k_frontend_x4.hip) that tail is 247 of the 319 vector instructions of a wave-symbol, i.e. 88 issued
// instructions per symbol and stream; here it is shared by sixteen streams.
//
// Mapping (quad r = lane / 4 serves stream 16 * (wave index) + r, t = lane % 4):
//   * lane t owns the interpolated samples Lam_j = L(pos + j - 10), j = 15 t + q, q = 0..14: sixteen CONSECUTIVE int16 IQ
//     samples from the quad's ring give its fifteen linear interpolations (the reference's interp(), :122-128, with the shared
//     fraction of k_frontend.hip), so every sample is unpacked and widened once per lane instead of twice per tap;
//   * the LO X[m] = exp(j m d), m = j - 10, from a seed per lane (same polynomial as the other kernels) and fourteen complex
//     multiplies by X[1];
//   * ONE window coefficient set T'[j] = exp(-j pi (j - 10) / 80) for all three gates: the on-time gate (j in [10, 50)) is
//     then exactly the other kernels' convention (its four sums P1..P4 are the carry in OpvStream, shared with every other
//     mapping), the early gate (j in [0, 40)) and the late gate (j in [20, 60)) come out rotated by exp(j pi / 8) and
//     exp(-j pi / 8) - and only their NORMS are used (:271-280). A tap is therefore accumulated ONCE (four FMAs), into one of
//     three per-lane groups (q < 5, q < 10, q < 15); a gate is a 0/1-weighted sum of groups (the five segments [0,10) [10,20)
//     [20,40) [40,50) [50,60) of the sample axis meet lane boundaries at 15, 30, 45), reduced over the quad by two
//     quad_perm steps;
//   * every stream-level quantity (pos, fo, tf, previous sums, chunk bookkeeping) lives in VGPRs, replicated over the 4
//     lanes of its quad; quads run their own chunk schedule under exec masks, batches as in k_frontend_x4.hip;
//   * int16 IQ: a 256-sample ring per quad in LDS (+ a 16-sample guard mirroring its head). Every second symbol each
//     quad requests the 16-sample blocks (64 B: one 16-byte load per lane, all sixteen quads in ONE instruction) that the two
//     symbols AFTER the next refill point will need, holds them in registers, and writes them into its ring at that next
//     point: two symbols of latency budget, no scalar bookkeeping per block, the HBM stream of a quad stays sequential.
//
// Differences from the reference are of the same kind and size as the other mappings' (shared interpolation fraction,
// factored LO, FMA, table atan2, and here one rotation-free re-association of the early / late sums): soft symbols agree to
// ~1e-14 of their mean, every decision downstream is identical (tests/test_gpu_parity.py runs every mapping).
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

constexpr uint32_t kRingSamples = 256;
constexpr uint32_t kRingBytes = kRingSamples * 4;   // 1024
constexpr uint32_t kGuardBytes = 64;                // mirror of the ring's first 16 samples (a lane reads 16 consecutive samples)
constexpr uint32_t kRowBytes = kRingBytes + kGuardBytes;   // 1088
constexpr uint32_t kBlk = 16;                       // samples per block: 4 lanes x 16 B
constexpr int kBlocksPerPoint = 6;                  // blocks a quad may request per refill point (two symbols consume <= 84 samples)
// Refill rule (g = floor(pos) of the quad at a refill point, hi = end of what its ring holds or has requested): a symbol reads
// samples g - 11 .. g + 55 and g grows by 38..42 per symbol. Blocks requested at point k are written at point k + 1 (two symbols
// later, g' <= g + 84) and have to carry the two symbols after THAT: up to g' + 42 + 55 <= g + 181. So a quad requests blocks
// while hi < g + 182 (at most 6: it held hi >= g_prev + 182 >= g + 98 already). A written block [h, h + 16) with h < g + 198
// takes the ring slot of [h - 256, h - 240), below g - 42 <= g' - 11 - 31: nothing reads that any more.
constexpr uint32_t kAhead = 182;
constexpr uint32_t kTabOff = 16 * kRowBytes;        // 17408 B of rings per wave
static_assert(kTabOff % 16 == 0, "16-byte LDS alignment");

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) unsigned char gbyte;

__device__ inline int dlo(double v) { return __double2loint(v); }
__device__ inline int dhi(double v) { return __double2hiint(v); }
__device__ inline double mkd(int hi, int lo) { return __hiloint2double(hi, lo); }

// four quad sums in lockstep (quad_perm [1,0,3,2] then [2,3,0,1]): a DPP read needs two wait states behind the VALU write of
// its source - the other three sums' instructions fill the slots (k_frontend_x4.hip: dpp_add4)
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
__device__ inline void quad_sum4(double& a, double& b, double& c, double& d) {
    dpp_add4<0xB1>(a, b, c, d);
    dpp_add4<0x4E>(a, b, c, d);
}
__device__ inline double quad_bcast3(double v) {        // lane 3 of the quad -> every lane of it (quad_perm [3,3,3,3])
    return mkd(__builtin_amdgcn_mov_dpp(dhi(v), 0xFF, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(dlo(v), 0xFF, 0xF, 0xF, true));
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
__device__ __noinline__ double2 silence_pd_x16(double dr, double di, double pa, double pb, double pc, double pd_, double x40c,
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

// this translation unit's own image of the angle table (opv_atan2.h: kOpvAtanTabQ; see k_frontend_x4.hip)
__constant__ double kOpvAtanTabQx16[257][6] = {
#include "opv_atan_table_q.inc"
};

// WPB = wavefronts per workgroup (four waves of one workgroup always land on the four SIMDs of a CU). Waves share only the
// atan table.
template <int WPB>
__device__ __forceinline__ void msk_frontend_x16_body(OpvStream* __restrict__ streams, OpvGlobalCfg cfg, int n_streams) {
    const int lane = threadIdx.x & 63, row = lane >> 2, t = lane & 3;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sidx = ((int)blockIdx.x * WPB + wave) * 16 + row;
    const bool have = sidx < n_streams;
    OpvStream& st = streams[have ? sidx : n_streams - 1];   // idle quads read a valid record and never write
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();

    __shared__ __attribute__((aligned(16))) unsigned char lds_all[WPB * kTabOff + 257 * 48];
    unsigned char* const lds = lds_all + wave * kTabOff;    // this wave's sixteen rings
    double* atab = reinterpret_cast<double*>(lds_all + WPB * kTabOff);
    for (int i = threadIdx.x; i < 257 * 6; i += 64 * WPB) atab[i] = (&kOpvAtanTabQx16[0][0])[i];
    unsigned char* const ring = lds + (uint32_t)row * kRowBytes;

    // ---- per-lane constants: T'[j] = (cos, -sin)(pi (j - 10) / 80), j = 15 t + q --------------------------------
    double Tc[15], Ts[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) {
        double sn, cs;
        sincospi((double)(15 * t + q - 10) / 80.0, &sn, &cs);
        Tc[q] = cs; Ts[q] = -sn;
    }
    // which of the lane's three tap groups (q < 5, < 10, < 15) lie inside each gate: E = j in [0,40), O = [10,50), L = [20,60)
    //   lane 0: j 0..14    lane 1: j 15..29    lane 2: j 30..44    lane 3: j 45..59
    const double wE0 = t <= 2 ? 1.0 : 0.0, wE1 = t <= 2 ? 1.0 : 0.0, wE2 = t <= 1 ? 1.0 : 0.0;
    const double wO0 = t >= 1 ? 1.0 : 0.0, wO1 = (t == 1 || t == 2) ? 1.0 : 0.0, wO2 = t <= 2 ? 1.0 : 0.0;
    const double wL0 = t >= 2 ? 1.0 : 0.0, wL1 = t >= 1 ? 1.0 : 0.0, wL2 = t >= 1 ? 1.0 : 0.0;
    const double kf0 = (double)(15 * t - 10);
    const double kfs0 = kf0 * kDeltaPerHz;
    const double kgain = st.afc_alpha * (kSymRate / kTwoPi);
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

    // ---- ring refill (see the rule at kAhead) --------------------------------------------------------------------
    // hi: the quad's ring holds (or has in flight) absolute samples [.., hi); blocks of 16 samples, 64 B aligned in the capture.
    uint32_t hi;
    {
        const uint32_t g0 = origin + (uint32_t)(int)mu;
        hi = (g0 >= 11u ? g0 - 11u : 0u) & ~(kBlk - 1u);
    }
    v4i blkv[kBlocksPerPoint];
    uint32_t blk_hi0 = 0, blk_n = 0;                      // the blocks in registers: samples [blk_hi0, blk_hi0 + 16 blk_n)
    auto load_block = [&](uint32_t h) -> v4i {           // this lane's 16 bytes of block [h, h + 16)
        const uint64_t off = (uint64_t)h * 4u + (uint32_t)t * 16u;
        v4i v = {0, 0, 0, 0};
        if (off + 16u <= n_bytes) v = *reinterpret_cast<const __attribute__((address_space(1))) v4i*>(iq_bytes + off);
        else if (off < n_bytes) {                         // the capture's last, incomplete 16 bytes: nothing past n_avail is read
            int e[4] = {0, 0, 0, 0};
            for (uint32_t j = 0; j < 4u && off + 4u * j < n_bytes; ++j)
                e[j] = *reinterpret_cast<const __attribute__((address_space(1))) int*>(iq_bytes + off + 4u * j);
            v = v4i{e[0], e[1], e[2], e[3]};
        }
        return v;
    };
    auto store_block = [&](uint32_t h, v4i v) {
        const uint32_t o = ((h * 4u) & (kRingBytes - 1u)) + (uint32_t)t * 16u;
        *reinterpret_cast<v4i*>(ring + o) = v;
        if (o < kGuardBytes) *reinterpret_cast<v4i*>(ring + kRingBytes + o) = v;   // the ring's head is mirrored behind its end
    };
    auto write_held = [&]() {                             // the blocks requested at the previous refill point have landed
#pragma unroll
        for (int k = 0; k < kBlocksPerPoint; ++k)
            if ((uint32_t)k < blk_n) store_block(blk_hi0 + kBlk * (uint32_t)k, blkv[k]);
        blk_n = 0;
    };
    auto request = [&](bool wants, uint32_t g) {          // at most kBlocksPerPoint blocks towards hi >= g + kAhead
        blk_hi0 = hi;
        uint32_t n = 0;
        if (wants && hi < g + kAhead && (uint64_t)hi * 4u < n_bytes) {
            n = (g + kAhead - hi + kBlk - 1u) / kBlk;
            if (n > (uint32_t)kBlocksPerPoint) n = (uint32_t)kBlocksPerPoint;
        }
#pragma unroll
        for (int k = 0; k < kBlocksPerPoint; ++k)
            if ((uint32_t)k < n) blkv[k] = load_block(hi + kBlk * (uint32_t)k);
        blk_n = n;
        hi += kBlk * n;
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
    // One symbol of every quad that executes this (exec = the quads inside a demodulate() call whose next symbol exists).
    // Generic: with the tests the first symbols of a call need (early gate before the chunk, no AFC on the first symbol,
    // an out-of-range -o still in force). Fast: the same statements without them - bit-identical where both apply
    // (no contraction, no re-association) - for the batches below. `slot`: which of a quad's four lanes keeps
    // this symbol's soft value until the next flush.
    auto symbol_body = [&](auto generic_tag, uint32_t slot) {
        constexpr bool kGeneric = decltype(generic_tag)::value;
        // ---- taps (ref :122-128, :232-238): sixteen consecutive samples -> fifteen interpolations ---------------
        const double pf = pos + kf0;
        const double fl = floor(pf);
        const double f = pf - fl;
        const int i0 = (int)fl;
        const uint32_t byte0 = (((uint32_t)(i0 + (int)origin)) << 2) & (kRingBytes - 1u);
        int w[16];
        {
            const int* tap = reinterpret_cast<const int*>(ring + byte0);
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = tap[q];
        }
        __builtin_amdgcn_sched_barrier(0);                              // taps requested FIRST, the LO seed under their latency
        // ---- LO: X[m] for m = 15 t - 10 + q: seed and step ---------------------------------------------------------
        double xs, xc, s1, c1;
        expj_small(kfs0 * fo, xs, xc);
        expj_small(kDeltaPerHz * fo, s1, c1);
        if (kGeneric && fabs(fo) > 2000.0) {
            // -o takes any value (ref :1004-1005) and the AFC clamp (:303) first acts at the END of the
            // call's second symbol: outside the polynomial's range those symbols take the full-range routine
            sincos(kfs0 * fo, &xs, &xc);
            sincos(kDeltaPerHz * fo, &s1, &c1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int s0_first = (kGeneric && first) ? *reinterpret_cast<const int*>(ring + ((origin << 2) & (kRingBytes - 1u))) : 0;

        double g0A = 0, g0B = 0, g0C = 0, g0D = 0, g1A = 0, g1B = 0, g1C = 0, g1D = 0, g2A = 0, g2B = 0, g2C = 0, g2D = 0;
        double x40c_l = 0.0, x40s_l = 0.0;
        double pr = (double)(int)(short)(w[0] & 0xFFFF), pi_ = (double)(w[0] >> 16);   // ref :1023
        asm("" : "+v"(pr), "+v"(pi_));
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            double nr = (double)(int)(short)(w[q + 1] & 0xFFFF), ni = (double)(w[q + 1] >> 16);
            asm("" : "+v"(nr), "+v"(ni));                          // (opaque: hipcc otherwise subtracts the int16 values and widens the difference too - 60 conversions instead of 32)
            double lr = fma(f, nr - pr, pr);                       // (differences of int16 values are exact in fp64)
            double li = fma(f, ni - pi_, pi_);
            if (kGeneric && first && pf + (double)q < 0.0) {        // early gate before the chunk: s[0] (ref :237)
                lr = (double)(int)(short)(s0_first & 0xFFFF);
                li = (double)(s0_first >> 16);
            }
            pr = nr; pi_ = ni;
            if (q == 5) { x40c_l = xc; x40s_l = xs; }               // X[40] is lane 3's sixth tap (m = 45 - 10 + 5)
            const double zr = fma(lr, xc, li * xs);                 // Z = Lam conj(X)
            const double zi = fma(li, xc, -(lr * xs));
            if (q < 5) { g0A = fma(zr, Tc[q], g0A); g0B = fma(zi, Ts[q], g0B); g0C = fma(zi, Tc[q], g0C); g0D = fma(zr, Ts[q], g0D); }
            else if (q < 10) { g1A = fma(zr, Tc[q], g1A); g1B = fma(zi, Ts[q], g1B); g1C = fma(zi, Tc[q], g1C); g1D = fma(zr, Ts[q], g1D); }
            else { g2A = fma(zr, Tc[q], g2A); g2B = fma(zi, Ts[q], g2B); g2C = fma(zi, Tc[q], g2C); g2D = fma(zr, Ts[q], g2D); }
            if (q < 14) {                                           // X[m + 1] = X[m] X[1]
                const double nc = fma(xc, c1, -(xs * s1));
                xs = fma(xc, s1, xs * c1);
                xc = nc;
            }
        }
        // X[40] = exp(j 40 d), needed by the NEXT symbol's phase detector
        const double x40c = quad_bcast3(x40c_l), x40s = quad_bcast3(x40s_l);
        // ---- on-time gate: soft value, dominant tone (ref :264-272) --------------------------
        double o1 = fma(wO2, g2A, fma(wO1, g1A, wO0 * g0A)), o2 = fma(wO2, g2B, fma(wO1, g1B, wO0 * g0B));
        double o3 = fma(wO2, g2C, fma(wO1, g1C, wO0 * g0C)), o4 = fma(wO2, g2D, fma(wO1, g1D, wO0 * g0D));
        quad_sum4(o1, o2, o3, o4);
        const double P1o = o1, P2o = o2, P3o = o3, P4o = o4;
        const double s1r_ = P1o + P2o, s1i_ = P3o - P4o;
        const double s2r_ = P1o - P2o, s2i_ = P3o + P4o;
        const double en1 = fma(s1r_, s1r_, s1i_ * s1i_);
        const double en2 = fma(s2r_, s2r_, s2i_ * s2i_);
        const double soft = en2 - en1;                      // ref :268
        const double nsg = mkd((dhi(soft) & (int)0x80000000) | 0x3ff00000, 0);  // -1 iff tone 1 dominates
        const double sg = -nsg;
        // ---- early / late gates of the dominant tone (ref :271-280): rotated by a constant, only their norms are used ----
        const double r0 = fma(sg, g0B, g0A), i0_ = fma(-sg, g0D, g0C);
        const double r1 = fma(sg, g1B, g1A), i1_ = fma(-sg, g1D, g1C);
        const double r2 = fma(sg, g2B, g2A), i2_ = fma(-sg, g2D, g2C);
        double Ere = fma(wE2, r2, fma(wE1, r1, wE0 * r0)), Eim = fma(wE2, i2_, fma(wE1, i1_, wE0 * i0_));
        double Lre = fma(wL2, r2, fma(wL1, r1, wL0 * r0)), Lim = fma(wL2, i2_, fma(wL1, i1_, wL0 * i0_));
        quad_sum4(Ere, Eim, Lre, Lim);
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
                const double2 sp = silence_pd_x16(dr, di, pv.a, pv.b, pv.c, pv.d, pv.x40c, pv.x40s, soft < 0.0, fo_sum,
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
    // ---- initial fill: everything the first two symbols can read, synchronously ---------------------------------
    {
        const uint32_t g = origin + (uint32_t)(int)mu;
        for (int rep = 0; rep < 3; ++rep) {                // (198 samples = 13 blocks at most)
            request(!done, g);
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
            write_held();
        }
    }
    __syncthreads();                     // atan table visible (single wave: LDS ordering only)

    // a refill point: the blocks requested two symbols ago go into the ring, the next ones are requested, the held soft
    // symbols leave (every second point: four symbols)
    auto refill_point = [&](bool flush) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): blocks and stores issued two symbols ago
        write_held();
        if (flush) flush_soft();
        request(in_call, origin + (uint32_t)(int)pos);
    };

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

        // ---- batches: as many symbols as EVERY quad inside a call can take without its end-of-call test, its first-symbol
        // rules or an out-of-range -o (pos advances by at most 42 samples per symbol), in groups of four (the soft-log
        // lanes and the refill points keep their rhythm). Quads outside a call are finished streams here: they sit the
        // batch out under the exec mask.
        if ((iter & 3u) == 0u) {
            int krow = 0x7fffffff;
            if (in_call) {
                krow = 0;
                const double room = Nd - 51.0 - pos;
                if (!first && !(fabs(fo) > 2000.0) && room > 0.0) krow = (int)(room * (1.0 / 42.0));
            }
#pragma unroll
            for (int off = 32; off >= 4; off >>= 1) { const int o = __shfl_xor(krow, off, 64); krow = o < krow ? o : krow; }
            const int kmin = __builtin_amdgcn_readfirstlane(krow);
            for (uint32_t quads = (uint32_t)kmin >> 2; quads != 0u; --quads) {
                refill_point(true);
                if (in_call) { symbol_body(std::false_type{}, 0u); symbol_body(std::false_type{}, 1u); }
                refill_point(false);
                if (in_call) { symbol_body(std::false_type{}, 2u); symbol_body(std::false_type{}, 3u); }
                iter += 4u;
            }
        }

        if ((iter & 1u) == 0u) refill_point((iter & 3u) == 0u);

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
        // where and at which clock the wave that carried this stream (and fifteen others) ran (opv_tap_wave_info)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st.dbg_hw_id = hw; st.dbg_xcc_id = xcc;
        st.dbg_cycles = __builtin_amdgcn_s_memtime() - dbg_t0;
        st.dbg_ticks = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
}

extern "C" __global__ __launch_bounds__(64) void k_msk_frontend_x16(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                     int n_streams) {
    msk_frontend_x16_body<1>(streams, cfg, n_streams);
}
// sixty-four streams per workgroup: one wave per SIMD of a CU by construction
extern "C" __global__ __launch_bounds__(256) void k_msk_frontend_x16_wg4(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                         int n_streams) {
    msk_frontend_x16_body<4>(streams, cfg, n_streams);
}
// 128 streams per workgroup: two waves per SIMD of a CU (8 x 17 KB of rings + one 12 KB angle table = 148 KB: one workgroup
// per CU; two of the four-wave kind miss the CU's 160 KB by 96 bytes)
extern "C" __global__ __launch_bounds__(512) void k_msk_frontend_x16_wg8(OpvStream* __restrict__ streams, OpvGlobalCfg cfg,
                                                                         int n_streams) {
    msk_frontend_x16_body<8>(streams, cfg, n_streams);
}
