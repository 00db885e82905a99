// k_frame_decode.hip — scale -> 3-bit quantise -> 67x32 deinterleave -> soft-decision K=7 r=1/2
// Viterbi (64 states) -> bit-pack -> CCSDS derandomise. One wavefront per frame: lane s owns
// trellis state s.
//
// Replaces FrameDecoder::decode (reference src/opv-demod.cpp:854-898), deinterleave_addr
// (:792-795) and ViterbiDecoder::decode (:800-847).
//
// Bit-exactness. Everything after the quantiser is integer. The quantiser itself
// (:856-866) is reproduced operation for operation: the scale is the SEQUENTIAL fp64 sum of
// |soft| (every lane runs the same 2144 dependent adds out of LDS — redundant, uniform, and
// cheap next to the front-end), then one IEEE divide, one multiply, two adds and a
// truncation per symbol with FMA contraction disabled. Given the same 2144 doubles this
// kernel returns the same bytes, decisions and metric as the reference, always.
//
// Viterbi on a wave: the 64 path metrics live one per lane (int32) under a ROTATING state-to-lane
// map (state s at time t in lane rotr6(s, t)), which turns the trellis step into an XOR butterfly:
// the two predecessors of the state a lane will hold are the lane itself and lane ^ (1 << k),
// k = (5 - t) mod 6 - one DPP quad_perm / row shift / v_permlane*_swap per step, no LDS-crossbar
// permute on the step's dependency chain. G1=0x4F has no tap on state bit 5, G2=0x6D has, so the
// other predecessor flips e2 only; the own predecessor's (e1, e2) are per-lane constants for each of
// the six phases. Add-compare-select with the reference's tie rule (m0 <= m1 -> lower predecessor,
// :829); the 64 decision bits of a step are two ballots -> one 64-bit word in LDS (1072 x 8 B =
// 8.6 KB/frame instead of the reference's 68.6 KB byte matrix). Traceback is a serial walk over those
// words in lane space from the first-minimum end state (:835-843), emitting bytes
// MSB-of-byte-133-first exactly as the packer does (:878-884), XORed with the LFSR table
// (:887-895; the LFSR restarts at 0xFF every frame so it is a constant 134-byte table).
// (The index algebra was checked against the oracle's decoder in a numpy model before it was written.)
//
// Bytes: 17 152 B of soft symbols in, 134 B out per frame (L2-resident right after the
// front-end). Integer ACS rate: 68 608 ACS/frame. No MFMA.
#include <hip/hip_runtime.h>
#include <math.h>

#include <type_traits>

#include "opv_device.h"

namespace {

struct LfsrTable { uint8_t b[OPV_FB]; };
constexpr LfsrTable make_lfsr() {  // ref :887-893
    LfsrTable t{};
    uint8_t st = 0xFF;
    for (int i = 0; i < OPV_FB; ++i) {
        uint8_t o = 0;
        for (int b = 7; b >= 0; --b) {
            o = (uint8_t)(o | (((st >> 7) & 1u) << b));
            const uint8_t fb = (uint8_t)(((st >> 7) ^ (st >> 6) ^ (st >> 4) ^ (st >> 2)) & 1u);
            st = (uint8_t)((st << 1) | fb);
        }
        t.b[i] = o;
    }
    return t;
}
__constant__ LfsrTable kLfsr = make_lfsr();

__device__ inline uint32_t deint_addr(uint32_t i) {  // ref :792-795
    const uint32_t p = (i & 31u) * 67u + (i >> 5);
    return (p & ~7u) + (7u - (p & 7u));
}

struct DecodeTaps {
    int8_t* q;      // [n][2144] or null
    int8_t* deint;  // [n][2144] or null
    uint8_t* bits;  // [n][1072] or null
};

// The whole FrameDecoder::decode for one frame, executed by one wave.
__device__ __forceinline__ void decode_one(const double* __restrict__ soft, uint32_t first, uint32_t mask,
                                  uint8_t* __restrict__ out,
                                  int32_t* __restrict__ metric_out, int8_t* tq, int8_t* td, uint8_t* tb,
                                  unsigned char* lds) {
    const int lane = threadIdx.x;
    // LDS: the 2144 soft doubles are dead once quantised, so everything after them lives in their space.
    // 17.3 KB per frame instead of 30 KB: nine frames per CU, i.e. two waves per SIMD hiding each other's
    // latencies (measured: 9.3 -> 5.6 ms per 64 000 frames from this alone).
    // s_q overlays the soft values it is computed from: lane l writes byte i = l + 64 it after ALL lanes of
    // the wave have read soft[64 it .. 64 it + 63] (bytes >= 512 it), so nothing unread is overwritten.
    double* s_soft = reinterpret_cast<double*>(lds);                                 // 17 152 B
    uint8_t* s_q = lds;                                                              //  2 144 B (over s_soft)
    uint8_t* s_d = lds + OPV_CODED;                                                  //  2 144 B (after quantising)
    unsigned long long* s_dec = reinterpret_cast<unsigned long long*>(lds + 2 * OPV_CODED);  // 8 576 B (4288 is 8-aligned)
    uint8_t* s_out = lds + 17152;                                                    //    136 B

    for (int i = lane; i < OPV_CODED; i += 64) s_soft[i] = soft[(first + (uint32_t)i) & mask];  // ring or linear (mask = ~0)
    __syncthreads();

    // ---- scale = mean |soft|, summed in index order (ref :856-858) --------------------------
    double scale = 0.0;
    {
        const double2* s2 = reinterpret_cast<const double2*>(s_soft);   // 16-byte LDS reads (same address in every lane: a broadcast)
#pragma unroll 8
        for (int i = 0; i < OPV_CODED / 2; ++i) {
            const double2 v = s2[i];
            scale += fabs(v.x);                                          // strictly in index order, like the reference's loop
            scale += fabs(v.y);
        }
    }
    scale /= (double)OPV_CODED;
    if (scale < 1e-10) {  // ref :859 — frame silently dropped
        if (lane == 0) *metric_out = -1;
        return;
    }

    // ---- quantise (ref :862-866): q=0 confident bit 0 ... q=7 confident bit 1 ----------------
    for (int i = lane; i < OPV_CODED; i += 64) {
        const double nrm = (-s_soft[i] / scale) * 3.5 + 3.5;  // contraction is off for this TU
        int v = (int)(nrm + 0.5);                             // C truncation toward zero
        v = v < 0 ? 0 : (v > 7 ? 7 : v);
        s_q[i] = (uint8_t)v;
        if (tq) tq[i] = (int8_t)v;
    }
    __syncthreads();
    // ---- deinterleave gather (ref :869-871) -----------------------------------------------
    for (int i = lane; i < OPV_CODED; i += 64) {
        const uint8_t v = s_q[deint_addr((uint32_t)i)];
        s_d[i] = v;
        if (td) td[i] = (int8_t)v;
    }
    __syncthreads();

    // ---- add-compare-select, 1072 steps (ref :810-833), as XOR butterflies ----------------------
    // The metric of state s at time t lives in lane rotr6(s, t): the predecessors (s>>1) and (s>>1)+32
    // (:815-816) of the state that will sit in lane l at time t+1 are then lane l ITSELF and lane
    // l ^ (1 << k), k = (5 - t) mod 6 - one DPP / permlane move per step instead of two LDS-crossbar
    // permutes on the step's critical path. Bit k of l is the state's input bit (:817) and tells which
    // of the two is the lower predecessor p0 (ties keep p0, :829). The expected code bits of the own
    // predecessor are per-lane constants for each of the six phases; 7 - x == x ^ 7 for 3-bit x.
    // Unreachable states carry 0x3FFFFFF0 instead of the reference's saturating 0x7FFFFFFF (:826-827):
    // they vanish after six steps, never win against a reachable one (finite metrics stay below 15 008),
    // and their decisions are never visited by the traceback.
    int m1c[6], m2c[6];
#pragma unroll
    for (int ph = 0; ph < 6; ++ph) {
        const int r = (ph + 1) % 6;
        const int st = r ? (((lane << r) | (lane >> (6 - r))) & 63) : lane;   // state in this lane at time t+1
        const int b0 = st & 1, pown = (st >> 1) | (b0 << 5), f = (b0 << 6) | pown;
        m1c[ph] = __builtin_parity((unsigned)(f & 0x4F)) ? 7 : 0;
        m2c[ph] = __builtin_parity((unsigned)(f & 0x6D)) ? 7 : 0;
    }
    int metric = (lane == 0) ? 0 : 0x3FFFFFF0;                   // ref :805-806
    const uint16_t* s_d2 = reinterpret_cast<const uint16_t*>(s_d);
    // The step's inputs and outputs are wave-uniform, and fetching / storing them one step at a time cost more
    // issue slots than the add-compare-select itself (an LDS read + wait + v_readfirstlane to get the symbol pair,
    // an exec-masked 64-bit LDS store by lane 0 for the decision word: 17 of 27 instructions per step). So the
    // trellis runs in blocks of 48 steps (a multiple of the six phases): lane l fetches the symbol pair of step
    // tb + l ONCE (one ds_read_u16 for the block), each step takes its pair with v_readlane and leaves its decision
    // word in lane (t - tb) of a VGPR pair with two v_writelane, and the 48 words are stored by ONE ds_write_b64.
    // The six butterfly masks live in SGPR pairs so that the word is three 64-bit scalar operations.
    unsigned long long kmask[6] = {0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull,
                                   0xFF00FF00FF00FF00ull, 0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
#pragma unroll
    for (int k = 0; k < 6; ++k) asm volatile("" : "+s"(kmask[k]));   // opaque: keeps them in SGPR pairs, 64-bit ops
    int pairs = 0, dlo = 0, dhi = 0;
    auto acs = [&](auto slot_tag) {                               // SLOT = t - tb: the lane that holds this step's pair / word
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int PH = SLOT % 6;                              // blocks start on a multiple of six steps
        constexpr int K = (5 - PH + 6) % 6;
        const unsigned pair = (unsigned)__builtin_amdgcn_readlane(pairs, SLOT);   // sg1 | sg2<<8, wave-uniform
        const int sg1 = (int)(pair & 0xFF), sg2 = (int)(pair >> 8);
        const int b1 = m1c[PH] ^ sg1, c = m2c[PH] ^ sg2;         // ref :823-824
        int mp;                                                   // metric held by lane ^ (1 << K)
        if constexpr (K == 0) mp = __builtin_amdgcn_mov_dpp(metric, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
        else if constexpr (K == 1) mp = __builtin_amdgcn_mov_dpp(metric, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        else if constexpr (K == 2) {
            mp = __builtin_amdgcn_update_dpp(0, metric, 0x104, 0xF, 0x5, false);                    // row_shl:4 -> banks 0,2
            mp = __builtin_amdgcn_update_dpp(mp, metric, 0x114, 0xF, 0xA, false);                   // row_shr:4 -> banks 1,3
        } else if constexpr (K == 3) mp = __builtin_amdgcn_mov_dpp(metric, 0x128, 0xF, 0xF, true); // row_ror:8
        else if constexpr (K == 4) {
            auto r = __builtin_amdgcn_permlane16_swap((unsigned)metric, (unsigned)metric, false, false);
            mp = (lane & 16) ? (int)r[0] : (int)r[1];
        } else {
            auto r = __builtin_amdgcn_permlane32_swap((unsigned)metric, (unsigned)metric, false, false);
            mp = (lane & 32) ? (int)r[0] : (int)r[1];
        }
        const int own = metric + b1 + c;
        const int oth = mp + b1 + (7 - c);
        const unsigned long long gt = __ballot(own > oth), lt = __ballot(own < oth);
        metric = own < oth ? own : oth;
        const unsigned long long word = (gt & ~kmask[K]) | (lt & kmask[K]);  // 1 = upper predecessor taken (:829-831)
        // (no clang builtin for v_writelane on this toolchain; the lane select is an inline constant, so the one
        // scalar operand the instruction may take is the data)
        int wl = dlo, wh = dhi;                                   // (locals: asm operands cannot name captures of a generic lambda)
        asm("v_writelane_b32 %0, %1, %2" : "+v"(wl) : "s"((unsigned)word), "n"(SLOT));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(wh) : "s"((unsigned)(word >> 32)), "n"(SLOT));
        dlo = wl; dhi = wh;
    };
#define OPV_ACS6(G)                                                                                            \
    acs(std::integral_constant<int, (G)>{}); acs(std::integral_constant<int, (G) + 1>{});                      \
    acs(std::integral_constant<int, (G) + 2>{}); acs(std::integral_constant<int, (G) + 3>{});                  \
    acs(std::integral_constant<int, (G) + 4>{}); acs(std::integral_constant<int, (G) + 5>{})
    constexpr int kBlk = 48;
    static_assert(kBlk % 6 == 0 && OPV_FBITS % kBlk == 16, "22 full blocks and a tail of 16 steps (6 + 6 + 4)");
    for (int tb = 0; tb + kBlk <= OPV_FBITS; tb += kBlk) {
        pairs = (int)s_d2[tb + (lane < kBlk ? lane : 0)];
        OPV_ACS6(0); OPV_ACS6(6); OPV_ACS6(12); OPV_ACS6(18); OPV_ACS6(24); OPV_ACS6(30); OPV_ACS6(36); OPV_ACS6(42);
        if (lane < kBlk) s_dec[tb + lane] = ((unsigned long long)(unsigned)dhi << 32) | (unsigned)dlo;
    }
    {
        constexpr int tb = OPV_FBITS - 16;
        pairs = (int)s_d2[tb + (lane & 15)];
        OPV_ACS6(0); OPV_ACS6(6);
        acs(std::integral_constant<int, 12>{}); acs(std::integral_constant<int, 13>{});
        acs(std::integral_constant<int, 14>{}); acs(std::integral_constant<int, 15>{});
        if (lane < 16) s_dec[tb + lane] = ((unsigned long long)(unsigned)dhi << 32) | (unsigned)dlo;
    }
#undef OPV_ACS6
    __syncthreads();

    // ---- best end state: first minimum in STATE order (ref :835-837) ---------------------------
    constexpr int kEndRot = OPV_FBITS % 6;                       // lane l holds state rotl6(l, 4) at the end
    int bm = metric, bs = ((lane << kEndRot) | (lane >> (6 - kEndRot))) & 63, bl = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int om = __shfl_xor(bm, off, 64), os = __shfl_xor(bs, off, 64), ol = __shfl_xor(bl, off, 64);
        if (om < bm || (om == bm && os < bs)) { bm = om; bs = os; bl = ol; }
    }

    // ---- traceback + pack + derandomise (ref :839-843, :878-895), uniform on all lanes ---------
    // in lane space: the decoded bit is bit k of the current lane, the predecessor's lane has that bit
    // replaced by the decision
    int cur = bl;
    // 24 steps are four turns of the bit position k and three output bytes: inside such a group every step's k is a
    // compile-time constant (shifts by immediates, no scalar bookkeeping). The bytes go to LDS still randomised; the
    // LFSR table is applied by all lanes at once on the way out (a per-byte constant-memory load inside this serial
    // walk would put a global-memory round trip on every eighth step).
    constexpr int kK0 = (5 - (OPV_FBITS - 1) % 6 + 6) % 6;
    auto trace_byte = [&](auto first_step_tag, int i) {
        constexpr int S0 = decltype(first_step_tag)::value;
        unsigned byte = 0;
        const int t0 = OPV_FBITS - 1 - 8 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = (kK0 + S0 + j) % 6;                     // k(t-1) = k(t) + 1 mod 6
            const int t = t0 - j;
            const unsigned b = (unsigned)(cur >> k) & 1u;         // bits[t] = s % 2 -> bit j of byte i
            byte |= b << j;
            if (tb && lane == 0) tb[t] = (uint8_t)b;
            const unsigned d = (unsigned)((s_dec[t] >> cur) & 1ull);
            cur ^= (int)((b ^ d) << k);
        }
        if (lane == 0) s_out[i] = (uint8_t)byte;
    };
    static_assert(OPV_FB % 3 == 2, "44 groups of three bytes and a tail of two");
    int i = 0;
    for (; i + 3 <= OPV_FB; i += 3) {
        trace_byte(std::integral_constant<int, 0>{}, i);
        trace_byte(std::integral_constant<int, 8>{}, i + 1);
        trace_byte(std::integral_constant<int, 16>{}, i + 2);
    }
    trace_byte(std::integral_constant<int, 0>{}, i);
    trace_byte(std::integral_constant<int, 8>{}, i + 1);
    __syncthreads();
    for (int q = lane; q < OPV_FB; q += 64) out[q] = (uint8_t)(s_out[q] ^ kLfsr.b[q]);   // derandomise (ref :887-895)
    if (lane == 0) *metric_out = bm;
}

constexpr int kDecodeLds = 17152 + 144;
static_assert(2 * OPV_CODED % 8 == 0 && 2 * OPV_CODED + 8 * OPV_FBITS <= 17152, "decision words fit behind s_q / s_d");

}  // namespace

// grid = n_streams x (max new frames per stream), flattened (stream-major: a stream's frames are neighbours, so are
// their soft symbols in L2); frames dec_from .. n_frames-1 of each stream
extern "C" __global__ __launch_bounds__(64) void k_frame_decode(OpvStream* __restrict__ streams, uint32_t per_stream) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDecodeLds];
    OpvStream& st = streams[blockIdx.x / per_stream];
    // per_stream is the host's ESTIMATE of the frames a stream releases in a round (from the samples it pushed); a
    // stream that was held back by back-pressure releases its backlog in one round, so the workgroups stride on
    const uint32_t n_frames = st.n_frames;
    for (uint32_t f = st.dec_from + blockIdx.x % per_stream; f < n_frames; f += per_stream) {
        const uint32_t slot = f % st.cap_frames;  // frame records / frames / metrics are rings
        const OpvFrameRec rec = st.frec[slot];
        decode_one(st.soft, (uint32_t)rec.payload_sym, (uint32_t)(st.cap_soft - 1), st.frames + (size_t)slot * OPV_FB,
                   st.metrics + slot, nullptr, nullptr, nullptr, lds);
        __syncthreads();                          // the next frame reuses this workgroup's LDS
    }
}

// stand-alone decoder over caller-provided payloads (parity tap / opv_decode_payloads)
extern "C" __global__ __launch_bounds__(64) void k_decode_payloads(const double* __restrict__ soft, uint32_t n,
                                                                    uint8_t* __restrict__ out,
                                                                    int32_t* __restrict__ metrics, int8_t* q,
                                                                    int8_t* deint, uint8_t* bits) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDecodeLds];
    const uint32_t f = blockIdx.x;
    if (f >= n) return;
    decode_one(soft + (size_t)f * OPV_CODED, 0u, 0xFFFFFFFFu, out + (size_t)f * OPV_FB, metrics + f,
               q ? q + (size_t)f * OPV_CODED : nullptr, deint ? deint + (size_t)f * OPV_CODED : nullptr,
               bits ? bits + (size_t)f * OPV_FBITS : nullptr, lds);
}
