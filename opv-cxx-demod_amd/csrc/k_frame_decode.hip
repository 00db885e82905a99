// k_frame_decode.hip — scale -> 3-bit quantise -> 67x32 deinterleave -> soft-decision K=7 r=1/2
// Viterbi (64 states) -> bit-pack -> CCSDS derandomise. One wavefront per frame: lane s owns
// trellis state s.
//
// Replaces FrameDecoder::decode (reference src/opv-demod.cpp:854-898), deinterleave_addr
// (:792-795) and ViterbiDecoder::decode (:800-847).
//
// Bit-exactness. Everything after the quantiser is integer. The quantiser itself
// (:856-866) is reproduced operation for operation: the scale is the SEQUENTIAL fp64 sum of
// |soft| in index order - 2144 dependent additions, which a wave-per-frame kernel can only run
// redundantly on its 64 lanes (round 2: 4.8 k of the 35 k instructions of a frame). They now run in a
// pre-pass with one FRAME per lane (k_frame_scale / k_payload_scale: 64 frames' sums side by side, same
// additions in the same order), and the decoder reads the scale - then one IEEE divide, one multiply,
// two adds and a truncation per symbol with FMA contraction disabled. Given the same 2144 doubles this
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
// Bytes: 17 152 B of soft symbols in (read twice: pre-pass and quantiser, the second time from L2), 134 B out per
// frame. With the scale known up front the soft doubles are never staged in LDS: a lane quantises straight into the
// deinterleaved position (a gather of 8-byte words inside the frame's 17 KB; a trellis step's two 3-bit values share a
// byte), and a frame needs 9.8 KB of LDS instead of 17.3 KB - sixteen frames per CU (four waves per SIMD) instead of nine. Integer ACS rate: 68 608 ACS/frame. No MFMA.
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

__device__ inline uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ inline uint32_t deint_addr(uint32_t i) {  // ref :792-795
    const uint32_t p = (i & 31u) * 67u + (i >> 5);
    return (p & ~7u) + (7u - (p & 7u));
}

struct DecodeTaps {
    int8_t* q;      // [n][2144] or null
    int8_t* deint;  // [n][2144] or null
    uint8_t* bits;  // [n][1072] or null
};

// sum |soft| of one payload in index order, divided by its length (ref :856-858): ONE lane's work. 64 lanes walk 64
// payloads, i.e. every load instruction touches 64 different cache lines: 16 bytes per lane and load (the ring is only
// 8-byte aligned at a payload's first symbol; unaligned 16-byte global loads are fine) unless the payload wraps the ring.
__device__ __forceinline__ double payload_scale(const double* __restrict__ soft, uint32_t first, uint32_t mask) {
    typedef double __attribute__((ext_vector_type(2), aligned(8))) double2u;        // 16-byte load, 8-byte aligned
    double sum = 0.0;
    first &= mask;
    if ((uint64_t)first + OPV_CODED - 1u <= (uint64_t)mask) {
        const double2u* p = reinterpret_cast<const double2u*>(soft + first);
        for (uint32_t i = 0; i < OPV_CODED / 2; i += 8) {
            double2u v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[i + j];                                 // eight loads in flight
#pragma unroll
            for (int j = 0; j < 8; ++j) { sum += fabs(v[j].x); sum += fabs(v[j].y); }    // strictly in index order, like the reference's loop
        }
    } else {
        for (uint32_t i = 0; i < OPV_CODED; ++i) sum += fabs(soft[(first + i) & mask]);
    }
    return sum / (double)OPV_CODED;
}

// FrameDecoder::decode for one frame behind its scale (ref :859-898), executed by one wave.
__device__ __forceinline__ void decode_one(const double* __restrict__ soft, uint32_t first, uint32_t mask, double scale,
                                  uint8_t* __restrict__ out,
                                  int32_t* __restrict__ metric_out, int8_t* tq, int8_t* td, uint8_t* tb,
                                  unsigned char* lds) {
    const int lane = threadIdx.x;
    uint8_t* s_d = lds;                                                              //  1 072 B: the step's two 3-bit values, one per nibble
    unsigned long long* s_dec = reinterpret_cast<unsigned long long*>(lds + OPV_FBITS);      //  8 576 B decision words (1072 is 8-aligned)
    uint8_t* s_out = lds + OPV_FBITS + 8 * OPV_FBITS;                                //    136 B

    if (scale < 1e-10) {  // ref :859 — frame silently dropped
        if (lane == 0) *metric_out = -1;
        return;
    }
    // ---- quantise (ref :862-866: q=0 confident bit 0 ... q=7 confident bit 1) straight into the deinterleaved
    // position (ref :869-871): value i of the decoder's input is symbol deint_addr(i) of the payload; a lane does the
    // pair (2 t, 2 t + 1) of trellis step t
    for (int t = lane; t < OPV_FBITS; t += 64) {
        unsigned pair = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t i = 2u * (uint32_t)t + (uint32_t)h, a = deint_addr(i);
            const double nrm = (-soft[(first + a) & mask] / scale) * 3.5 + 3.5;  // contraction is off for this TU
            int v = (int)(nrm + 0.5);                                            // C truncation toward zero
            v = v < 0 ? 0 : (v > 7 ? 7 : v);
            pair |= (unsigned)v << (4 * h);
            if (tq) tq[a] = (int8_t)v;
            if (td) td[i] = (int8_t)v;
        }
        s_d[t] = (uint8_t)pair;
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
        const unsigned pair = (unsigned)__builtin_amdgcn_readlane(pairs, SLOT);   // sg1 | sg2 << 4, wave-uniform
        const int sg1 = (int)(pair & 0xF), sg2 = (int)(pair >> 4);
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
        pairs = (int)s_d[tb + (lane < kBlk ? lane : 0)];
        OPV_ACS6(0); OPV_ACS6(6); OPV_ACS6(12); OPV_ACS6(18); OPV_ACS6(24); OPV_ACS6(30); OPV_ACS6(36); OPV_ACS6(42);
        if (lane < kBlk) s_dec[tb + lane] = ((unsigned long long)(unsigned)dhi << 32) | (unsigned)dlo;
    }
    {
        constexpr int tb = OPV_FBITS - 16;
        pairs = (int)s_d[tb + (lane & 15)];
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

    // ---- traceback + pack + derandomise (ref :839-843, :878-895) on the SCALAR unit ----------------
    // in lane space: the decoded bit is bit k of the current lane, the predecessor's lane has that bit replaced by the
    // decision. The walk is one chain of 1072 dependent steps, the same for every lane: as vector code (round 2: 8.7
    // instructions per bit, a v_lshrrev_b64 by the current lane among them) it filled the SIMD's issue slots with 64
    // copies of one number. Now the decision words come a block at a time (lane l <- word of step t_hi - l: one
    // ds_read_b64 per 48 steps), each step takes its word by two v_readlane and does everything else in SGPRs - shift by
    // the current lane, bit extract, three xors - which the scalar unit runs beside the other waves' vector work.
    // Blocks are 48 steps (eight turns of the bit position k, six output bytes) so that every k is an immediate; the 16
    // steps of 1072 = 16 + 22 x 48 go first. The bytes go to LDS still randomised; the LFSR table is applied by all lanes
    // at once on the way out.
    uint32_t cur = uni32((uint32_t)bl);
    constexpr int kK0 = (5 - (OPV_FBITS - 1) % 6 + 6) % 6;       // bit position of the walk's first step (t = 1071)
    auto trace_block = [&](auto nsteps_tag, auto k0_tag, int t_hi, int byte0) {
        constexpr int NS = decltype(nsteps_tag)::value;          // steps of this block: t_hi, t_hi - 1, ...
        constexpr int KB = decltype(k0_tag)::value;              // k of its first step
        const unsigned long long mine = s_dec[t_hi - (lane < NS ? lane : 0)];
        const int wlo = (int)(unsigned)mine, whi = (int)(unsigned)(mine >> 32);
        unsigned long long acc = 0;                              // decoded bits in walk order: bit j of byte i is step 8 i + j
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int k = (KB + j) % 6;                          // k(t-1) = k(t) + 1 mod 6
            const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(whi, j) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane(wlo, j);
            const uint32_t b = (cur >> k) & 1u;                  // bits[t] = s % 2
            const uint32_t d = (uint32_t)(word >> cur) & 1u;
            acc |= (unsigned long long)b << j;
            if (tb) { if (lane == 0) tb[t_hi - j] = (uint8_t)b; }
            cur ^= (b ^ d) << k;
        }
        if (lane < NS / 8) s_out[byte0 + lane] = (uint8_t)(acc >> (8 * lane));
    };
    static_assert(OPV_FBITS == 16 + 22 * 48, "a head of 16 steps (two bytes), then 22 blocks of 48 (six bytes each)");
    trace_block(std::integral_constant<int, 16>{}, std::integral_constant<int, kK0>{}, OPV_FBITS - 1, 0);
    for (int blk = 0; blk < 22; ++blk)
        trace_block(std::integral_constant<int, 48>{}, std::integral_constant<int, (kK0 + 16) % 6>{}, OPV_FBITS - 17 - 48 * blk, 2 + 6 * blk);
    __syncthreads();
    for (int q = lane; q < OPV_FB; q += 64) out[q] = (uint8_t)(s_out[q] ^ kLfsr.b[q]);   // derandomise (ref :887-895)
    if (lane == 0) *metric_out = bm;
}

constexpr int kDecodeLds = OPV_FBITS + 8 * OPV_FBITS + 144;      // 9 792 B: sixteen frames per CU = four waves per SIMD
static_assert(OPV_FBITS % 8 == 0, "decision words are 8-byte aligned behind the 1072 value pairs");
static_assert(16 * kDecodeLds <= 160 * 1024, "sixteen workgroups per CU");

}  // namespace

// Pre-pass, one FRAME per lane: scale of frames dec_from .. n_frames-1 of every stream into st.fscale (same flattening and
// striding as k_frame_decode below: thread id -> stream id / per_stream, frames dec_from + id % per_stream, + per_stream, ...)
extern "C" __global__ __launch_bounds__(64) void k_frame_scale(OpvStream* __restrict__ streams, uint32_t per_stream, uint32_t n_streams) {
    const uint32_t id = blockIdx.x * 64u + threadIdx.x;
    const uint32_t sidx = id / per_stream;
    if (sidx >= n_streams) return;
    OpvStream& st = streams[sidx];
    const uint32_t n_frames = st.n_frames;
    for (uint32_t f = st.dec_from + id % per_stream; f < n_frames; f += per_stream) {
        const uint32_t slot = f % st.cap_frames;
        st.fscale[slot] = payload_scale(st.soft, (uint32_t)st.frec[slot].payload_sym, (uint32_t)(st.cap_soft - 1));
    }
}

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
        decode_one(st.soft, (uint32_t)rec.payload_sym, (uint32_t)(st.cap_soft - 1), st.fscale[slot], st.frames + (size_t)slot * OPV_FB,
                   st.metrics + slot, nullptr, nullptr, nullptr, lds);
        __syncthreads();                          // the next frame reuses this workgroup's LDS
    }
}

// stand-alone decoder over caller-provided payloads (parity tap / opv_decode_payloads): scales first, one payload per lane
extern "C" __global__ __launch_bounds__(64) void k_payload_scale(const double* __restrict__ soft, uint32_t n, double* __restrict__ scales) {
    const uint32_t f = blockIdx.x * 64u + threadIdx.x;
    if (f < n) scales[f] = payload_scale(soft + (size_t)f * OPV_CODED, 0u, 0xFFFFFFFFu);
}
extern "C" __global__ __launch_bounds__(64) void k_decode_payloads(const double* __restrict__ soft, uint32_t n,
                                                                    const double* __restrict__ scales,
                                                                    uint8_t* __restrict__ out,
                                                                    int32_t* __restrict__ metrics, int8_t* q,
                                                                    int8_t* deint, uint8_t* bits) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDecodeLds];
    const uint32_t f = blockIdx.x;
    if (f >= n) return;
    decode_one(soft + (size_t)f * OPV_CODED, 0u, 0xFFFFFFFFu, scales[f], out + (size_t)f * OPV_FB, metrics + f,
               q ? q + (size_t)f * OPV_CODED : nullptr, deint ? deint + (size_t)f * OPV_CODED : nullptr,
               bits ? bits + (size_t)f * OPV_FBITS : nullptr, lds);
}
