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

// ---- packed 16-bit helpers: the two frames of a wave live in the two halves of every metric register ------------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }

constexpr int kBlk = 48;                                          // trellis steps per block: a multiple of the six phases, three 16-bit decision words
constexpr int kTabSteps = 24;                                     // steps per branch-metric table refill
constexpr int kDecWords = (OPV_FBITS / kBlk) * 3 * 64 + 64;       // 22 blocks x 3 words x 64 lanes + the 16-step tail: 4 288 words
constexpr int kDecodeLds = 4 * kDecWords + 2 * OPV_FBITS + kTabSteps * 32;   // 17 152 + 2 144 + 768 = 20 064 B: eight workgroups (16 frames) per CU
static_assert(OPV_FBITS == 16 + 22 * kBlk, "22 full blocks and a tail of 16 steps");
static_assert(8 * ((kDecodeLds + 511) / 512) * 512 <= 160 * 1024, "eight workgroups per CU");

struct FrameIo {                      // one of the two frames of a wave
    const double* soft; uint32_t first, mask; double scale;       // payload = soft[(first + i) & mask], i < 2144
    uint8_t* out; int32_t* metric_out;
    int8_t* tq; int8_t* td; uint8_t* tb;                           // parity taps (null in the product path)
    bool present;                                                  // false: the wave's second half is idle (odd frame count)
};

// FrameDecoder::decode for TWO frames behind their scales (ref :859-898), executed by one wave: frame A in the low 16 bits
// of every path-metric register, frame B in the high 16 bits (path metrics stay below 15 008 + 14, the unreachable-state value
// is 0x3FF0: a u32 add of two packed values never carries from A into B).
__device__ __forceinline__ void decode_two(const FrameIo& A, const FrameIo& B, unsigned char* lds) {
    const int lane = threadIdx.x;
    uint32_t* s_dec = reinterpret_cast<uint32_t*>(lds);                                   // 17 152 B decision words
    uint16_t* s_pair = reinterpret_cast<uint16_t*>(lds + 4 * kDecWords);                  //  2 144 B: step t -> A's two 3-bit values | B's << 8
    unsigned char* s_tab = lds + 4 * kDecWords + 2 * OPV_FBITS;                           //    768 B: 24 steps x 4 classes x {x, y}
    uint8_t* s_out = reinterpret_cast<uint8_t*>(s_pair);                                  // 2 x 136 B, over the value pairs once the trellis is done

    const bool liveA = A.present && !(A.scale < 1e-10), liveB = B.present && !(B.scale < 1e-10);   // ref :859 - frame silently dropped
    if (lane == 0) {
        if (A.present && !liveA) *A.metric_out = -1;
        if (B.present && !liveB) *B.metric_out = -1;
    }
    if (!liveA && !liveB) return;
    // ---- quantise (ref :862-866: q=0 confident bit 0 ... q=7 confident bit 1) straight into the deinterleaved
    // position (ref :869-871): value i of the decoder's input is symbol deint_addr(i) of the payload; a lane does the
    // pair (2 t, 2 t + 1) of trellis step t, for both frames
    auto quantise = [&](const FrameIo& F, int t) -> unsigned {
        unsigned pair = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t i = 2u * (uint32_t)t + (uint32_t)h, a = deint_addr(i);
            const double nrm = (-F.soft[(F.first + a) & F.mask] / F.scale) * 3.5 + 3.5;  // contraction is off for this TU
            int v = (int)(nrm + 0.5);                                                  // C truncation toward zero
            v = v < 0 ? 0 : (v > 7 ? 7 : v);
            pair |= (unsigned)v << (4 * h);
            if (F.tq) F.tq[a] = (int8_t)v;
            if (F.td) F.td[i] = (int8_t)v;
        }
        return pair;
    };
    for (int t = lane; t < OPV_FBITS; t += 64) {
        const unsigned pa = liveA ? quantise(A, t) : 0u, pb = liveB ? quantise(B, t) : 0u;
        s_pair[t] = (uint16_t)(pa | (pb << 8));
    }
    __syncthreads();

    // ---- add-compare-select, 1072 steps (ref :810-833), as XOR butterflies on packed metrics ------------------------
    // The metric of state s at time t lives in lane rotr6(s, t): the predecessors (s>>1) and (s>>1)+32 (:815-816) of the
    // state that will sit in lane l at time t+1 are then lane l ITSELF and lane l ^ (1 << K), K = (5 - t) mod 6. Bit K of l
    // (u) is the state's input bit (:817) and says whether the lane's own predecessor is the upper one; ties keep the
    // lower (:829). The expected code bits (e1, e2) of the own predecessor (:821-822) are a per-lane constant c = 2 e1 + e2
    // for each of the six phases; the other predecessor's differ in e2 only (G1 = 0x4F has no tap on state bit 5, G2 = 0x6D
    // has), the partner lane's own class is c ^ 2 and its other class c ^ 3.
    // Branch metrics: a step's two received values give four sums bm(j) = (j&2 ? 7-sg1 : sg1) + (j&1 ? 7-sg2 : sg2)
    // (:823-824) for BOTH frames (packed); 24 lanes build the tables of the next 24 steps in LDS - entry j = {bm(j),
    // bm(j ^ flip)}, flip = 1 for the DPP phases and 3 for the swap phases - and every lane fetches ITS entry of a step with
    // one ds_read_b64 (address = per-lane, per-phase constant + immediate): no per-step arithmetic on the received values.
    //   K < 4  own = m + T.x;  oth = m[lane ^ (1 << K)] + T.y   (v_add_u32 with the DPP move folded in)
    //          raw = sign(own - oth - beta), beta = 1 - u: "the own predecessor wins" with the tie rule in it
    //   K >= 4 a = m + T.x;  b = m + T.y;  v_permlane{16,32}_swap(a, b) puts own / other side by side: for lanes with
    //          u = 0 (a, b) = (own, oth), for u = 1 (oth, own) - their table index is flipped so that each side received the
    //          addend the partner needs; raw = sign(a - b - 1) is "own wins" for u = 0 and its complement for u = 1
    //   m' = min of the two (v_pk_min_u16); the step's raw bit of both frames (bits 15 and 31) is shifted into a per-lane
    //   16-step decision word (v_lshrrev + v_bfi) - no ballots, no lane-indexed writes.
    // Unreachable states carry 0x3FF0 instead of the reference's saturating 0x7FFFFFFF (:826-827): they vanish after six
    // steps (every state is reachable then), never win against a reachable one (those are <= 6 x 14 by then, and
    // 0x3FF0 + 6 x 14 stays below 2^15 so that the 16-bit differences keep their sign), and their decisions are never
    // visited by the traceback. Reachable metrics are <= 1072 x 14 = 15 008: 16 bits hold them and differences of them.
    uint32_t tabofs[6], beta[4], flipm[3] = {0u, 0u, 0u};
#pragma unroll
    for (int ph = 0; ph < 6; ++ph) {
        const int K = (5 - ph + 6) % 6, r = (ph + 1) % 6;
        const int st = r ? (((lane << r) | (lane >> (6 - r))) & 63) : lane;   // state in this lane at time t+1
        const int b0 = st & 1, pown = (st >> 1) | (b0 << 5), f = (b0 << 6) | pown;
        const int c = (__builtin_parity((unsigned)(f & 0x4F)) << 1) | __builtin_parity((unsigned)(f & 0x6D));
        const int u = (lane >> K) & 1;                                        // == b0
        tabofs[ph] = (uint32_t)(((K >= 4 && u) ? (c ^ 3) : c) * 8);
        if (K < 4) beta[K] = u ? 0u : 0x00010001u;
        // what the traceback wants from a decision bit is "the walk moves to the partner lane": !raw for K < 4,
        // raw ^ !u for K >= 4 - folded into the stored words (slots 16 w + j of a block have phase (16 w + j) % 6)
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((16 * w + j) % 6 == ph && (K < 4 || !u)) flipm[w] |= 0x00010001u << j;
    }
    uint32_t metric = (lane == 0) ? 0u : 0x3FF03FF0u;            // ref :805-806
    uint32_t acc[3] = {0u, 0u, 0u};
    const uint32_t kSignMask = 0x80008000u;

    auto build_tables = [&](int t0) {                             // steps t0 .. t0 + 23 (fewer at the tail), one per lane
        if (lane < kTabSteps && t0 + lane < OPV_FBITS) {
            const uint32_t w = s_pair[t0 + lane];
            const uint32_t s1 = (w & 0xFu) | ((w & 0xF00u) << 8), s2 = ((w >> 4) & 0xFu) | ((w & 0xF000u) << 4);
            const uint32_t n1 = 0x00070007u - s1, n2 = 0x00070007u - s2;
            const uint32_t bm0 = s1 + s2, bm1 = s1 + n2, bm2 = n1 + s2, bm3 = n1 + n2;
            const int ph = (t0 + lane) % 6;
            const bool swp = ph < 2;                              // K = 5, 4: the permlane-swap phases pair j with j ^ 3
            uint4* row = reinterpret_cast<uint4*>(s_tab + lane * 32);
            row[0] = make_uint4(bm0, swp ? bm3 : bm1, bm1, swp ? bm2 : bm0);
            row[1] = make_uint4(bm2, swp ? bm1 : bm3, bm3, swp ? bm0 : bm2);
        }
    };
    auto acs = [&](auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;           // step inside the block (blocks start on a multiple of six steps)
        constexpr int PH = SLOT % 6, K = (5 - PH + 6) % 6, W = SLOT / 16;
        const uint2 e = *reinterpret_cast<const uint2*>(s_tab + (SLOT % kTabSteps) * 32 + tabofs[PH]);
        uint32_t x, y, bsub;
        if constexpr (K < 4) {
            x = metric + e.x;
            if constexpr (K == 0) y = (uint32_t)__builtin_amdgcn_mov_dpp((int)metric, 0xB1, 0xF, 0xF, true) + e.y;        // quad_perm [1,0,3,2]
            else if constexpr (K == 1) y = (uint32_t)__builtin_amdgcn_mov_dpp((int)metric, 0x4E, 0xF, 0xF, true) + e.y;   // quad_perm [2,3,0,1]
            else if constexpr (K == 3) y = (uint32_t)__builtin_amdgcn_mov_dpp((int)metric, 0x128, 0xF, 0xF, true) + e.y;  // row_ror:8
            else {                                                 // lane ^ 4: banks 0, 2 read four lanes up, banks 1, 3 four lanes down
                asm("v_add_u32_dpp %0, %1, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                    "v_add_u32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                    : "=&v"(y) : "v"(metric), "v"(e.y));
            }
            bsub = beta[K];
        } else {
            x = metric + e.x;
            y = metric + e.y;
            if constexpr (K == 4) { auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); x = r[0]; y = r[1]; }
            else { auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0]; y = r[1]; }
            bsub = 0x00010001u;
        }
        const uint32_t d = pk_sub(pk_sub(x, y), bsub);
        metric = pk_min(x, y);
        // the step's two sign bits (15 and 31) into the word, which moves down a place: step j of the word ends in bits j and 16 + j
        // (locals: asm operands cannot name captures of a generic lambda)
        const uint32_t sh = acc[W] >> 1, sm = kSignMask;
        uint32_t merged;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(merged) : "s"(sm), "v"(d), "v"(sh));
        acc[W] = merged;
    };
#define OPV_ACS6(G)                                                                                            \
    acs(std::integral_constant<int, (G)>{}); acs(std::integral_constant<int, (G) + 1>{});                      \
    acs(std::integral_constant<int, (G) + 2>{}); acs(std::integral_constant<int, (G) + 3>{});                  \
    acs(std::integral_constant<int, (G) + 4>{}); acs(std::integral_constant<int, (G) + 5>{})
    for (int blk = 0; blk < OPV_FBITS / kBlk; ++blk) {
        build_tables(blk * kBlk);
        __syncthreads();
        OPV_ACS6(0); OPV_ACS6(6); OPV_ACS6(12); OPV_ACS6(18);
        __syncthreads();                                          // every lane has read its entries of the first 24 steps
        build_tables(blk * kBlk + kTabSteps);
        __syncthreads();
        OPV_ACS6(24); OPV_ACS6(30); OPV_ACS6(36); OPV_ACS6(42);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 3; ++w) s_dec[(blk * 3 + w) * 64 + lane] = acc[w] ^ flipm[w];
    }
    {
        build_tables(OPV_FBITS - 16);
        __syncthreads();
        OPV_ACS6(0); OPV_ACS6(6);
        acs(std::integral_constant<int, 12>{}); acs(std::integral_constant<int, 13>{});
        acs(std::integral_constant<int, 14>{}); acs(std::integral_constant<int, 15>{});
        s_dec[(OPV_FBITS / kBlk) * 3 * 64 + lane] = acc[0] ^ flipm[0];
    }
#undef OPV_ACS6
    __syncthreads();

    // ---- best end state of each frame: first minimum in STATE order (ref :835-837) ---------------------------
    constexpr int kEndRot = OPV_FBITS % 6;                       // lane l holds state rotl6(l, 4) at the end
    const int my_state = ((lane << kEndRot) | (lane >> (6 - kEndRot))) & 63;
    int bmA = (int)(metric & 0xFFFFu), bsA = my_state, blA = lane, bmB = (int)(metric >> 16), bsB = my_state, blB = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int om = __shfl_xor(bmA, off, 64), os = __shfl_xor(bsA, off, 64), ol = __shfl_xor(blA, off, 64);
        if (om < bmA || (om == bmA && os < bsA)) { bmA = om; bsA = os; blA = ol; }
        const int pm = __shfl_xor(bmB, off, 64), ps = __shfl_xor(bsB, off, 64), pl = __shfl_xor(blB, off, 64);
        if (pm < bmB || (pm == bmB && ps < bsB)) { bmB = pm; bsB = ps; blB = pl; }
    }

    // ---- traceback + pack (ref :839-843, :878-884) on the SCALAR unit, both frames interleaved ----------------
    // In lane space: step t's decision word of lane `cur` (one v_readlane with a scalar lane select) says whether the walk
    // moves to the partner lane, cur ^= bit << K - three scalar instructions per step and frame. The decoded bit of step t
    // is bit K of the lane the walk stands on (bits[t] = s % 2, :841), and a step changes bit K only: the six bits of `cur`
    // at a step with K = 0 ARE the next six decoded bits in walk order, so the output accumulates 6 bits at a time. Walk
    // order = bit order of the packer (byte 0 bit 0 is t = 1071, :878-884). Bytes go to LDS still randomised; the LFSR
    // table is applied by all lanes at once on the way out.
    uint32_t curA = uni32((uint32_t)blA), curB = uni32((uint32_t)blB);
    uint8_t* s_outA = s_out;
    uint8_t* s_outB = s_out + 136;
    uint8_t* const tbA = liveA ? A.tb : nullptr;                  // parity tap: the 1072 hard decisions (null in the product path)
    uint8_t* const tbB = liveB ? B.tb : nullptr;
    {   // the 16 steps t = 1071 .. 1056 (slots 15 .. 0 of the tail word): two bytes, bit by bit
        const int wv = (int)s_dec[(OPV_FBITS / kBlk) * 3 * 64 + lane];
        uint32_t oa = 0, ob = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int slot = 15 - j, K = (5 - slot % 6 + 6) % 6;
            oa |= ((curA >> K) & 1u) << j;
            ob |= ((curB >> K) & 1u) << j;
            if (tbA) { if (lane == 0) tbA[OPV_FBITS - 1 - j] = (uint8_t)((curA >> K) & 1u); }
            if (tbB) { if (lane == 0) tbB[OPV_FBITS - 1 - j] = (uint8_t)((curB >> K) & 1u); }
            const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane(wv, (int)curA), wb = (uint32_t)__builtin_amdgcn_readlane(wv, (int)curB);
            curA ^= ((wa >> slot) & 1u) << K;
            curB ^= ((wb >> (16 + slot)) & 1u) << K;
        }
        if (lane < 2) { s_outA[lane] = (uint8_t)(oa >> (8 * lane)); s_outB[lane] = (uint8_t)(ob >> (8 * lane)); }
    }
    for (int blk = OPV_FBITS / kBlk - 1; blk >= 0; --blk) {
        int wv[3];
#pragma unroll
        for (int w = 0; w < 3; ++w) wv[w] = (int)s_dec[(blk * 3 + w) * 64 + lane];
        unsigned long long oa = 0, ob = 0;                       // 48 decoded bits per frame in walk order
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
            const int slot = kBlk - 1 - j, K = (5 - slot % 6 + 6) % 6;
            if (j % 6 == 0) {                                     // K == 0 here: the walk's next six bits
                oa |= (unsigned long long)curA << j;
                ob |= (unsigned long long)curB << j;
            }
            const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane(wv[slot / 16], (int)curA);
            const uint32_t wb = (uint32_t)__builtin_amdgcn_readlane(wv[slot / 16], (int)curB);
            curA ^= ((wa >> (slot % 16)) & 1u) << K;
            curB ^= ((wb >> (16 + slot % 16)) & 1u) << K;
        }
        const int byte0 = 2 + 6 * (OPV_FBITS / kBlk - 1 - blk);
        if (lane < 6) { s_outA[byte0 + lane] = (uint8_t)(oa >> (8 * lane)); s_outB[byte0 + lane] = (uint8_t)(ob >> (8 * lane)); }
        if (tbA || tbB) {                                         // the 48 decisions of this block, one per lane
            const int t = blk * kBlk + kBlk - 1 - lane;
            if (lane < kBlk) {
                if (tbA) tbA[t] = (uint8_t)((oa >> lane) & 1u);
                if (tbB) tbB[t] = (uint8_t)((ob >> lane) & 1u);
            }
        }
    }
    __syncthreads();
    for (int q = lane; q < OPV_FB; q += 64) {                    // derandomise (ref :887-895)
        if (liveA) A.out[q] = (uint8_t)(s_outA[q] ^ kLfsr.b[q]);
        if (liveB) B.out[q] = (uint8_t)(s_outB[q] ^ kLfsr.b[q]);
    }
    if (lane == 0) {
        if (liveA) *A.metric_out = bmA;
        if (liveB) *B.metric_out = bmB;
    }
}

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

// grid = n_streams x ceil(max new frames per stream / 2), flattened (stream-major: a stream's frames are neighbours, so are
// their soft symbols in L2); a workgroup decodes frames dec_from + 2 j and dec_from + 2 j + 1 of its stream, then strides on
extern "C" __global__ __launch_bounds__(64) void k_frame_decode(OpvStream* __restrict__ streams, uint32_t pairs_per_stream) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDecodeLds];
    OpvStream& st = streams[blockIdx.x / pairs_per_stream];
    // pairs_per_stream comes from the host's ESTIMATE of the frames a stream releases in a round (from the samples it pushed); a
    // stream that was held back by back-pressure releases its backlog in one round, so the workgroups stride on
    const uint32_t n_frames = st.n_frames, mask = (uint32_t)(st.cap_soft - 1);
    for (uint32_t f = st.dec_from + 2u * (blockIdx.x % pairs_per_stream); f < n_frames; f += 2u * pairs_per_stream) {
        const uint32_t sa = f % st.cap_frames, sb = (f + 1u) % st.cap_frames;  // frame records / frames / metrics are rings
        const bool two = f + 1u < n_frames;
        const FrameIo A{st.soft, (uint32_t)st.frec[sa].payload_sym, mask, st.fscale[sa], st.frames + (size_t)sa * OPV_FB, st.metrics + sa,
                        nullptr, nullptr, nullptr, true};
        const FrameIo B{st.soft, two ? (uint32_t)st.frec[sb].payload_sym : 0u, mask, two ? st.fscale[sb] : 0.0, st.frames + (size_t)sb * OPV_FB,
                        st.metrics + sb, nullptr, nullptr, nullptr, two};
        decode_two(A, B, lds);
        __syncthreads();                          // the next pair reuses this workgroup's LDS
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
    const uint32_t f = 2u * blockIdx.x;                     // payloads f and f + 1 share the wave
    if (f >= n) return;
    auto io = [&](uint32_t k, bool present) {
        return FrameIo{soft + (size_t)k * OPV_CODED, 0u, 0xFFFFFFFFu, present ? scales[k] : 0.0, out + (size_t)k * OPV_FB, metrics + k,
                       q && present ? q + (size_t)k * OPV_CODED : nullptr, deint && present ? deint + (size_t)k * OPV_CODED : nullptr,
                       bits && present ? bits + (size_t)k * OPV_FBITS : nullptr, present};
    };
    const bool two = f + 1u < n;
    decode_two(io(f, true), io(two ? f + 1u : f, two), lds);
}
