// k_frame_decode.hip — scale -> 3-bit quantise -> 67x32 deinterleave -> soft-decision K=7 r=1/2
// Viterbi (64 states) -> bit-pack -> CCSDS derandomise. One wavefront per TWO frames: a lane owns one
// trellis state of each (packed 16-bit path metrics).
//
// Replaces FrameDecoder::decode (reference src/opv-demod.cpp:854-898), deinterleave_addr
// (:792-795) and ViterbiDecoder::decode (:800-847).
//
// Bit-exactness. Everything after the quantiser is integer. The quantiser itself
// (:856-866) is reproduced operation for operation: the scale is the SEQUENTIAL fp64 sum of
// |soft| in index order - 2144 dependent additions, which a wave-per-frame kernel can only run
// redundantly on its 64 lanes (round 2: 4.8 k of the 35 k instructions of a frame). They now run in a
// pre-pass with one FRAME per lane (k_frame_scale / k_payload_scale: 64 frames' sums side by side, same
// additions in the same order), and the decoder reads the scale. The quantised value itself - one IEEE divide, one
// multiply, two adds and a truncation per symbol in the reference, FMA contraction disabled - is decided by ONE multiply-add
// wherever that provably gives the same integer, and by the reference's own sequence inside a guard band around the
// quantiser's boundaries (decode_two). Given the same 2144 doubles this kernel returns the same bytes, decisions and
// metric as the reference, always.
//
// Viterbi on a wave, TWO frames per wave (round 4): the 64 path metrics of a frame live one per lane under a ROTATING
// state-to-lane map (state s at time t in lane rotr6(s, t)), frame A in the low 16 bits of the register and frame B in
// the high 16 bits (metrics stay below 15 008 + 14, so packed 16-bit adds / min serve both). The rotating map turns the
// trellis step into an XOR butterfly: the two predecessors of the state a lane will hold are the lane itself and
// lane ^ (1 << k), k = (5 - t) mod 6 - a DPP modifier on the add (k < 4) or one v_permlane{16,32}_swap of the two candidate
// registers (k >= 4), no LDS-crossbar permute and no select on the step's dependency chain. Branch metrics come from
// per-step class tables in LDS (four sums of the step's two received values, both frames packed), read a group of 24 steps
// ahead. Add-compare-select with the reference's tie rule (m0 <= m1 -> lower predecessor, :829). Survivors are kept as
// trace-forward pointers (the lane a lane's survivor stood in six steps ago, 6 bits per six steps = the same 8.6 KB per
// frame as one decision bit per step; the reference keeps a 68.6 KB byte matrix), so the walk back from the
// first-minimum end state (:835-843) hops six steps per scalar round trip and reads its six decoded bits off the lane
// number. Bytes are emitted MSB-of-byte-133-first exactly as the packer does (:878-884) and XORed with the LFSR table
// (:887-895; the LFSR restarts at 0xFF every frame so it is a constant 134-byte table).
// The index algebra was checked against the oracle's decoder in a numpy model before the kernel was written
// (scripts/models/viterbi_packed_model.py: decision-bit and trace-forward forms).
//
// Bytes: 17 152 B of soft symbols in (read twice: pre-pass and quantiser, the second time from L2), 134 B out per
// frame. The soft doubles are never staged in LDS: a lane quantises straight into the deinterleaved position (a gather of
// 8-byte words inside the frame's 17 KB; a trellis step's two 3-bit values share a byte). A wave needs 20.3 KB of LDS:
// eight workgroups = sixteen frames per CU. Integer ACS rate: 68 608 ACS/frame. No MFMA.
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
        // sixteen 16-byte loads per batch, and the next batch is requested before this one is summed (64 lanes walk 64
        // payloads: nothing but the lane's own requests hides the memory latency; with eight loads in flight and none
        // ahead, a lone round of 64 frames spent 68 us here, ten times the 2144 dependent additions)
        const double2u* p = reinterpret_cast<const double2u*>(soft + first);
        constexpr uint32_t kB = 16, kN = OPV_CODED / 2 / kB;                                // 67 batches
        static_assert(kB * kN * 2 == OPV_CODED, "batches tile the payload");
        double2u cur[kB], nxt[kB];
#pragma unroll
        for (uint32_t j = 0; j < kB; ++j) cur[j] = p[j];
        for (uint32_t b = 0; b < kN; ++b) {
            const uint32_t nb = b + 1 < kN ? b + 1 : b;                                     // (the last trip re-reads its own batch)
#pragma unroll
            for (uint32_t j = 0; j < kB; ++j) nxt[j] = p[nb * kB + j];
#pragma unroll
            for (uint32_t j = 0; j < kB; ++j) { sum += fabs(cur[j].x); sum += fabs(cur[j].y); }    // strictly in index order, like the reference's loop
#pragma unroll
            for (uint32_t j = 0; j < kB; ++j) cur[j] = nxt[j];
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

constexpr int kTabSteps = 24;                                     // steps per branch-metric table refill (four groups of six)
constexpr int kChunk = 96;                                        // steps per chunk: 16 groups of six = 16 six-bit fields = three words per frame
constexpr int kChunks = OPV_FBITS / kChunk;                       // 11 full chunks; 16 steps remain (groups 176, 177 and the 4-step group 178)
constexpr int kGroups = (OPV_FBITS + 5) / 6;                      // 179 pointer fields per frame
constexpr int kDecWords = kChunks * 6 * 64 + 2 * 64;              // [chunk][frame][3][lane] + [frame][lane] for the remainder: 4 352 words
constexpr int kDecodeLds = 4 * kDecWords + 2 * OPV_FBITS + kTabSteps * 32;   // 17 408 + 2 144 + 768 = 20 320 B: eight workgroups (16 frames) per CU
static_assert(OPV_FBITS == kChunks * kChunk + 16 && kGroups == 16 * kChunks + 3, "11 chunks of 96 steps, then 6 + 6 + 4 steps");
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
    uint32_t* s_dec = reinterpret_cast<uint32_t*>(lds);                                   // 17 408 B survivor pointer fields
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
    // pairs (2 t, 2 t + 1) of trellis steps t = lane, lane + 64, ... of one frame after the other.
    // The reference's expression per value (:864-865) is int((-soft / scale) * 3.5 + 3.5 + 0.5), four rounded operations of which
    // the IEEE division alone costs a wave ~30 fp64 instructions. Its result is decided by ONE multiply-add,
    // x = fma(soft, -3.5 / scale, 4.0), whenever x is not within 1e-6 of an integer: both expressions are within ~1e-14 of
    // the exact value 4 - 3.5 soft / scale for |x| < 9 (|soft / scale| <= 2144 always, so nothing overflows), hence they
    // truncate alike unless the exact value is within 1e-14 of an integer - and then x is within the guard band and the
    // reference's own sequence is evaluated (about one value in 10^5 on noise; every value of a few-level input). Outside
    // (-1, 9) both clamp to the same end. NaN (never produced by the front-end) converts to 0 on both paths, like
    // the reference's cvttsd2si + clamp.
    // All of a frame's loads of a lane (34 gathered doubles, from L2 / HBM) are requested before the first is used: a wave
    // that waited for them one trellis step at a time spent 30 000 cycles here (17 round trips), more than in the trellis.
    auto quantise_frame = [&](const FrameIo& F, int byte_of_pair) {
        constexpr int kIter = (OPV_FBITS + 63) / 64;              // 17 steps per lane (the last one for lanes < 48 only)
        double sv[2 * kIter];
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const int t = lane + 64 * k;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t a = deint_addr(2u * (uint32_t)(t < OPV_FBITS ? t : 0) + (uint32_t)h);
                sv[2 * k + h] = F.soft[(F.first + a) & F.mask];
            }
        }
        const double inv = -3.5 / F.scale;
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const int t = lane + 64 * k;
            if (t < OPV_FBITS) {
                unsigned pair = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t i = 2u * (uint32_t)t + (uint32_t)h, a = deint_addr(i);
                    const double x = fma(sv[2 * k + h], inv, 4.0);
                    int v = (int)x;
                    if (__builtin_expect(fabs(x - rint(x)) < 1.0e-6 && x > -1.0 && x < 9.0, 0)) {
                        const double nrm = (-sv[2 * k + h] / F.scale) * 3.5 + 3.5;         // contraction is off for this TU
                        v = (int)(nrm + 0.5);                                              // C truncation toward zero
                    }
                    v = v < 0 ? 0 : (v > 7 ? 7 : v);
                    pair |= (unsigned)v << (4 * h);
                    if (F.tq) F.tq[a] = (int8_t)v;
                    if (F.td) F.td[i] = (int8_t)v;
                }
                reinterpret_cast<uint8_t*>(s_pair)[2 * t + byte_of_pair] = (uint8_t)pair;
            }
        }
    };
    if (liveA) quantise_frame(A, 0);
    else for (int t = lane; t < OPV_FBITS; t += 64) reinterpret_cast<uint8_t*>(s_pair)[2 * t] = 0;
    if (liveB) quantise_frame(B, 1);
    else for (int t = lane; t < OPV_FBITS; t += 64) reinterpret_cast<uint8_t*>(s_pair)[2 * t + 1] = 0;
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
    // bm(j ^ flip)}, flip = 1 for the DPP phases and 3 for the swap phases - and every lane fetches ITS entries of those steps
    // (address = per-lane, per-phase constant + immediate) a whole group of 24 ahead: no arithmetic on the received values
    // and no LDS round trip on the metric's dependency chain.
    //   K < 4  x = m + T.x;  y = m[lane ^ (1 << K)] + T.y   (v_add_u32 with the DPP move folded in): own / other candidate
    //          raw = sign(x - y - beta), beta = 1 - u: "x wins" with the tie rule in it
    //   K >= 4 a = m + T.x;  b = m + T.y;  v_permlane{16,32}_swap(a, b) puts own / other side by side: for lanes with
    //          u = 0 (x, y) = (own, other), for u = 1 (other, own) - their table index is flipped so that each side received the
    //          addend the partner needs; raw = sign(x - y - 1) is again "x wins"
    //   m' = min(x, y) (v_pk_min_u16).
    // Survivors are kept as TRACE-FORWARD POINTERS, not as decision bits: register P holds, per frame, the lane this lane's
    // survivor stood in at the last multiple of six steps. A step moves pointers exactly as it moves metrics - P' = raw ?
    // Px : Py with Px / Py fetched by the same DPP move / swap as x / y (v_pk_ashrrev_i16 turns the two sign bits of the
    // difference into select masks, one v_bfi_b32 selects for both frames) - and every sixth step P is filed as a 6-bit field
    // (16 fields = 96 bits = three words per chunk of 96 steps, lane and frame) and reset to the lane's own number. The walk
    // back then hops SIX steps per scalar round trip (below); with one decision bit per step it took one v_readlane -> scalar
    // -> v_readlane round trip per step, 87 000 cycles a wave, more than the trellis itself.
    // Unreachable states carry 0x3FF0 instead of the reference's saturating 0x7FFFFFFF (:826-827): they vanish after six
    // steps (every state is reachable then), never win against a reachable one (those are <= 6 x 14 by then, and
    // 0x3FF0 + 6 x 14 stays below 2^15 so that the 16-bit differences keep their sign), and their pointers are never
    // followed by the walk. Reachable metrics are <= 1072 x 14 = 15 008: 16 bits hold them and differences of them.
    uint32_t tabofs[6], beta[4];
#pragma unroll
    for (int ph = 0; ph < 6; ++ph) {
        const int K = (5 - ph + 6) % 6, r = (ph + 1) % 6;
        const int st = r ? (((lane << r) | (lane >> (6 - r))) & 63) : lane;   // state in this lane at time t+1
        const int b0 = st & 1, pown = (st >> 1) | (b0 << 5), f = (b0 << 6) | pown;
        const int c = (__builtin_parity((unsigned)(f & 0x4F)) << 1) | __builtin_parity((unsigned)(f & 0x6D));
        const int u = (lane >> K) & 1;                                        // == b0
        tabofs[ph] = (uint32_t)(((K >= 4 && u) ? (c ^ 3) : c) * 8);
        if (K < 4) beta[K] = u ? 0u : 0x00010001u;
    }
    uint32_t metric = (lane == 0) ? 0u : 0x3FF03FF0u;            // ref :805-806
    const uint32_t kIdent = (uint32_t)lane * 0x00010001u;        // "my survivor stood in this very lane", both frames
    uint32_t ptr = kIdent;
    const uint32_t kFifteen = 0x000F000Fu;                        // shift count of v_pk_ashrrev_i16, both halves
    uint32_t fldA[3] = {0u, 0u, 0u}, fldB[3] = {0u, 0u, 0u};     // the chunk's 16 fields per frame

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
    uint2 eA[kTabSteps], eB[kTabSteps];
    auto fetch = [&](uint2 (&e)[kTabSteps]) {                     // this lane's entries of the 24 steps the table holds (it starts on a multiple of 6)
#pragma unroll
        for (int i = 0; i < kTabSteps; ++i) e[i] = *reinterpret_cast<const uint2*>(s_tab + i * 32 + tabofs[i % 6]);
    };
    auto acs = [&](auto slot_tag, const uint2 e) {
        constexpr int SLOT = decltype(slot_tag)::value;           // step inside the chunk (chunks start on a multiple of six steps)
        constexpr int PH = SLOT % 6, K = (5 - PH + 6) % 6;
        uint32_t x, y, px, py, bsub;
        if constexpr (K < 4) {
            x = metric + e.x;
            px = ptr;
            if constexpr (K == 0) {                                // quad_perm [1,0,3,2]
                y = (uint32_t)__builtin_amdgcn_mov_dpp((int)metric, 0xB1, 0xF, 0xF, true) + e.y;
                py = (uint32_t)__builtin_amdgcn_mov_dpp((int)ptr, 0xB1, 0xF, 0xF, true);
            } else if constexpr (K == 1) {                         // quad_perm [2,3,0,1]
                y = (uint32_t)__builtin_amdgcn_mov_dpp((int)metric, 0x4E, 0xF, 0xF, true) + e.y;
                py = (uint32_t)__builtin_amdgcn_mov_dpp((int)ptr, 0x4E, 0xF, 0xF, true);
            } else if constexpr (K == 3) {                         // row_ror:8
                y = (uint32_t)__builtin_amdgcn_mov_dpp((int)metric, 0x128, 0xF, 0xF, true) + e.y;
                py = (uint32_t)__builtin_amdgcn_mov_dpp((int)ptr, 0x128, 0xF, 0xF, true);
            } else {                                               // lane ^ 4: banks 0, 2 read four lanes up, banks 1, 3 four lanes down
                const uint32_t m = metric, ey = e.y, p = ptr;     // (locals: asm operands cannot name captures of a generic lambda)
                asm("v_add_u32_dpp %0, %1, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                    "v_add_u32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                    : "=&v"(y) : "v"(m), "v"(ey));
                asm("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                    "v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                    : "=&v"(py) : "v"(p));
            }
            bsub = beta[K];
        } else {
            x = metric + e.x;
            y = metric + e.y;
            if constexpr (K == 4) {
                auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); x = r[0]; y = r[1];
                auto q = __builtin_amdgcn_permlane16_swap(ptr, ptr, false, false); px = q[0]; py = q[1];
            } else {
                auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0]; y = r[1];
                // K = 5 is the FIRST step of a group: every pointer is still its lane's own number, so the swapped pair would be
                // (lane & 31, lane | 32) on both sides - no move needed, the select below is done on constants
                px = kIdent & 0x001F001Fu; py = kIdent | 0x00200020u;
            }
            bsub = 0x00010001u;
        }
        const uint32_t d = pk_sub(pk_sub(x, y), bsub);
        metric = pk_min(x, y);
        // 0xFFFF per frame whose x candidate won, then one select for both frames (as instructions: hipcc turns the C form
        // into two 16-bit compares, two selects per half and a byte permute)
        uint32_t xwins, sel;
        const uint32_t k15 = kFifteen;                            // (locals: asm operands cannot name captures of a generic lambda)
        asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(xwins) : "s"(k15), "v"(d));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(xwins), "v"(px), "v"(py));
        ptr = sel;
        if constexpr (PH == 5) {                                  // six steps done: file the pointer as field SLOT / 6, start afresh
            constexpr int I = SLOT / 6, O = 6 * I, W = O / 32, SH = O % 32;
            const uint32_t pa = ptr & 63u, pb = ptr >> 16;
            fldA[W] |= pa << SH;
            fldB[W] |= pb << SH;
            if constexpr (SH > 26) { fldA[W + 1] |= pa >> (32 - SH); fldB[W + 1] |= pb >> (32 - SH); }
            // (no reset: the next step is a K = 5 step, which starts the pointers afresh by itself)
        }
    };
#define OPV_ACS6(G, E, O)                                                                                                          \
    acs(std::integral_constant<int, (G)>{}, E[(G) - (O)]); acs(std::integral_constant<int, (G) + 1>{}, E[(G) + 1 - (O)]);          \
    acs(std::integral_constant<int, (G) + 2>{}, E[(G) + 2 - (O)]); acs(std::integral_constant<int, (G) + 3>{}, E[(G) + 3 - (O)]);  \
    acs(std::integral_constant<int, (G) + 4>{}, E[(G) + 4 - (O)]); acs(std::integral_constant<int, (G) + 5>{}, E[(G) + 5 - (O)])
#define OPV_ACS24(G, E) OPV_ACS6(G, E, G); OPV_ACS6((G) + 6, E, G); OPV_ACS6((G) + 12, E, G); OPV_ACS6((G) + 18, E, G)
    // software pipeline over groups of 24 steps: while a group runs from registers, the next group's tables are built and read
    auto refill = [&](uint2 (&e)[kTabSteps], int t0) {
        __syncthreads();                                          // (a single wave: orders this lane's earlier reads before the rows are rewritten)
        build_tables(t0);
        __syncthreads();
        fetch(e);
    };
    refill(eA, 0);
    for (int c = 0; c < kChunks; ++c) {
        const int t0 = c * kChunk;
        refill(eB, t0 + 24);
        OPV_ACS24(0, eA);
        refill(eA, t0 + 48);
        OPV_ACS24(24, eB);
        refill(eB, t0 + 72);
        OPV_ACS24(48, eA);
        refill(eA, t0 + 96);                                      // (the last chunk: the 16 steps of the remainder)
        OPV_ACS24(72, eB);
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            s_dec[(c * 6 + w) * 64 + lane] = fldA[w];
            s_dec[(c * 6 + 3 + w) * 64 + lane] = fldB[w];
            fldA[w] = 0u; fldB[w] = 0u;
        }
    }
    {   // steps 1056 .. 1071: groups 176 and 177, then the four steps of group 178 (fields 0, 1, 2 of a last word)
        OPV_ACS6(0, eA, 0); OPV_ACS6(6, eA, 0);
        acs(std::integral_constant<int, 12>{}, eA[12]); acs(std::integral_constant<int, 13>{}, eA[13]);
        acs(std::integral_constant<int, 14>{}, eA[14]); acs(std::integral_constant<int, 15>{}, eA[15]);
        s_dec[(kChunks * 6) * 64 + lane] = fldA[0] | ((ptr & 63u) << 12);
        s_dec[(kChunks * 6 + 1) * 64 + lane] = fldB[0] | ((ptr >> 16) << 12);
    }
#undef OPV_ACS24
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

    // ---- the walk back + pack (ref :839-843, :878-884), six steps per hop, both frames interleaved ----------------
    // In lane space a step changes bit K of the lane the walk stands on, and the decoded bit of that step IS that bit before
    // the change (bits[t] = s % 2, :841); K runs 0, 1, .. 5 over the six steps below a multiple of six. So the six decoded
    // bits of a hop are the six bits of the lane it starts from, in walk order, and the hop itself is one field lookup:
    // v_readlane (scalar lane select) + s_bfe. Walk order = bit order of the packer (byte 0 bit 0 is t = 1071, :878-884).
    // Bytes go to LDS still randomised; the LFSR table is applied by all lanes at once on the way out.
    uint32_t curA = uni32((uint32_t)blA), curB = uni32((uint32_t)blB);
    uint8_t* s_outA = s_out;
    uint8_t* s_outB = s_out + 136;
    uint8_t* const tbA = liveA ? A.tb : nullptr;                  // parity tap: the 1072 hard decisions (null in the product path)
    uint8_t* const tbB = liveB ? B.tb : nullptr;
    {   // t = 1071 .. 1056: the 4-step group (K = 2, 3, 4, 5: bits 2..5 of the end lane), then groups 177 and 176
        const int wa = (int)s_dec[(kChunks * 6) * 64 + lane], wb = (int)s_dec[(kChunks * 6 + 1) * 64 + lane];
        uint32_t oa = curA >> 2, ob = curB >> 2;
        curA = __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(wa, (int)curA), 12u, 6u);
        curB = __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(wb, (int)curB), 12u, 6u);
        oa |= curA << 4; ob |= curB << 4;
        curA = __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(wa, (int)curA), 6u, 6u);
        curB = __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(wb, (int)curB), 6u, 6u);
        oa |= curA << 10; ob |= curB << 10;
        curA = __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(wa, (int)curA), 0u, 6u);
        curB = __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(wb, (int)curB), 0u, 6u);
        if (lane < 2) { s_outA[lane] = (uint8_t)(oa >> (8 * lane)); s_outB[lane] = (uint8_t)(ob >> (8 * lane)); }
        if (lane < 16) {
            if (tbA) tbA[OPV_FBITS - 1 - lane] = (uint8_t)((oa >> lane) & 1u);
            if (tbB) tbB[OPV_FBITS - 1 - lane] = (uint8_t)((ob >> lane) & 1u);
        }
    }
    for (int c = kChunks - 1; c >= 0; --c) {
        int wa[3], wb[3];
#pragma unroll
        for (int w = 0; w < 3; ++w) { wa[w] = (int)s_dec[(c * 6 + w) * 64 + lane]; wb[w] = (int)s_dec[(c * 6 + 3 + w) * 64 + lane]; }
        uint32_t oa[3] = {0u, 0u, 0u}, ob[3] = {0u, 0u, 0u};     // 96 decoded bits per frame in walk order
#pragma unroll
        for (int j = 0; j < 16; ++j) {                            // hop j starts at time t0 + 96 - 6 j and uses field 15 - j
            const int o = 6 * j, ow = o / 32, osh = o % 32;
            oa[ow] |= curA << osh; ob[ow] |= curB << osh;
            if (osh > 26) { oa[ow + 1] |= curA >> (32 - osh); ob[ow + 1] |= curB >> (32 - osh); }
            const int f = 6 * (15 - j), fw = f / 32, fsh = f % 32;
            uint32_t na = (uint32_t)__builtin_amdgcn_readlane(wa[fw], (int)curA) >> fsh;
            uint32_t nb = (uint32_t)__builtin_amdgcn_readlane(wb[fw], (int)curB) >> fsh;
            if (fsh > 26) {
                na |= (uint32_t)__builtin_amdgcn_readlane(wa[fw + 1], (int)curA) << (32 - fsh);
                nb |= (uint32_t)__builtin_amdgcn_readlane(wb[fw + 1], (int)curB) << (32 - fsh);
            }
            curA = na & 63u; curB = nb & 63u;
        }
        const int byte0 = 2 + 12 * (kChunks - 1 - c);
        if (lane < 12) {
            const int w = lane >> 2, sh = 8 * (lane & 3);
            s_outA[byte0 + lane] = (uint8_t)((w == 0 ? oa[0] : w == 1 ? oa[1] : oa[2]) >> sh);
            s_outB[byte0 + lane] = (uint8_t)((w == 0 ? ob[0] : w == 1 ? ob[1] : ob[2]) >> sh);
        }
        if (tbA || tbB) {                                         // the 96 decisions of this chunk
            for (int j = lane; j < kChunk; j += 64) {
                const int t = c * kChunk + kChunk - 1 - j, w = j >> 5, sh = j & 31;
                if (tbA) tbA[t] = (uint8_t)(((w == 0 ? oa[0] : w == 1 ? oa[1] : oa[2]) >> sh) & 1u);
                if (tbB) tbB[t] = (uint8_t)(((w == 0 ? ob[0] : w == 1 ? ob[1] : ob[2]) >> sh) & 1u);
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

// The same pre-pass for SMALL rounds (a live round of one frame per stream): one WAVE per frame. With one frame per lane a
// round of 64 frames is a single wave whose lanes walk 64 different payloads - 46 us of memory latency for 2144 additions.
// Here the wave fetches its frame's 17 KB at once (34 coalesced 512-byte loads, all in flight) into LDS and then adds the
// 2144 values from there, in index order like the reference (:856-858; every lane runs the same chain, lane 0 stores).
extern "C" __global__ __launch_bounds__(64) void k_frame_scale_wave(OpvStream* __restrict__ streams, uint32_t per_stream) {
    __shared__ __attribute__((aligned(16))) double vals[OPV_CODED];
    OpvStream& st = streams[blockIdx.x / per_stream];
    const uint32_t n_frames = st.n_frames, mask = (uint32_t)(st.cap_soft - 1);
    const int lane = threadIdx.x;
    for (uint32_t f = st.dec_from + blockIdx.x % per_stream; f < n_frames; f += per_stream) {
        const uint32_t slot = f % st.cap_frames, first = (uint32_t)st.frec[slot].payload_sym;
        constexpr int kIter = (OPV_CODED + 63) / 64;            // 34 (the last one for lanes < 32 only)
        double v[kIter];
#pragma unroll
        for (int k = 0; k < kIter; ++k) { const int i = lane + 64 * k; v[k] = st.soft[(first + (uint32_t)(i < OPV_CODED ? i : 0)) & mask]; }
#pragma unroll
        for (int k = 0; k < kIter; ++k) { const int i = lane + 64 * k; if (i < OPV_CODED) vals[i] = fabs(v[k]); }
        __syncthreads();
        double sum = 0.0;
#pragma unroll 16
        for (int i = 0; i < OPV_CODED; ++i) sum += vals[i];     // strictly in index order
        if (lane == 0) st.fscale[slot] = sum / (double)OPV_CODED;
        __syncthreads();                                         // the next frame reuses the buffer
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
