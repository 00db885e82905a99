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
// Viterbi on a wave. metrics[64] live one per lane (int32, with the reference's 0x7FFFFFFF
// sentinel logic, :805,:826-827). Step t: the four possible branch metrics are wave-uniform
// (only (e1,e2) in {0,1}^2 exist); lane s selects the two it needs by its constant parity
// pattern (G1=0x4F has no tap on state bit 5, G2=0x6D has, so the upper predecessor flips
// e2 only), fetches the predecessor metrics from lanes s>>1 and (s>>1)+32 with ds_bpermute,
// add-compare-selects with the reference's tie rule (m0 <= m1 -> lower predecessor, :829)
// and the 64 decision bits of the step are one __ballot -> one 64-bit word in LDS
// (1072 x 8 B = 8.6 KB/frame instead of the reference's 68.6 KB byte matrix). Traceback is a
// serial walk over those words from the first-minimum end state (:835-843), emitting bytes
// MSB-of-byte-133-first exactly as the packer does (:878-884), XORed with the LFSR table
// (:887-895; the LFSR restarts at 0xFF every frame so it is a constant 134-byte table).
//
// Bytes: 17 152 B of soft symbols in, 134 B out per frame (L2-resident right after the
// front-end). Integer ACS rate: 68 608 ACS/frame. No MFMA.
#include <hip/hip_runtime.h>
#include <math.h>

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
__device__ inline void decode_one(const double* __restrict__ soft, uint32_t first, uint32_t mask,
                                  uint8_t* __restrict__ out,
                                  int32_t* __restrict__ metric_out, int8_t* tq, int8_t* td, uint8_t* tb,
                                  unsigned char* lds) {
    const int lane = threadIdx.x;
    double* s_soft = reinterpret_cast<double*>(lds);                                 // 17 152 B
    unsigned long long* s_dec = reinterpret_cast<unsigned long long*>(lds + 17152);  //  8 576 B
    uint8_t* s_q = lds + 17152 + 8576;                                               //  2 144 B
    uint8_t* s_d = s_q + OPV_CODED;                                                  //  2 144 B
    uint8_t* s_out = s_d + OPV_CODED;                                                //    136 B

    for (int i = lane; i < OPV_CODED; i += 64) s_soft[i] = soft[(first + (uint32_t)i) & mask];  // ring or linear (mask = ~0)
    __syncthreads();

    // ---- scale = mean |soft|, summed in index order (ref :856-858) --------------------------
    double scale = 0.0;
#pragma unroll 8
    for (int i = 0; i < OPV_CODED; ++i) scale += fabs(s_soft[i]);
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

    // ---- add-compare-select, 1072 steps (ref :810-833) ---------------------------------------
    const int s = lane;
    const int p0 = s >> 1, p1 = p0 + 32, inb = s & 1;
    const int f0 = (inb << 6) | p0;
    const bool e1 = __builtin_parity((unsigned)(f0 & 0x4F));  // same for both predecessors
    const bool e2 = __builtin_parity((unsigned)(f0 & 0x6D));  // flipped for p1 (bit 5 of 0x6D)
    int metric = (s == 0) ? 0 : 0x7FFFFFFF;                   // ref :805-806
    const uint16_t* s_d2 = reinterpret_cast<const uint16_t*>(s_d);
#pragma unroll 4
    for (int t = 0; t < OPV_FBITS; ++t) {
        const unsigned pair = s_d2[t];  // sg1 | sg2<<8, wave-uniform LDS broadcast
        const int sg1 = (int)(pair & 0xFF), sg2 = (int)(pair >> 8);
        const int b1 = e1 ? 7 - sg1 : sg1;                    // ref :823-824
        const int bm0 = b1 + (e2 ? 7 - sg2 : sg2);
        const int bm1 = b1 + (e2 ? sg2 : 7 - sg2);
        const int mp0 = __shfl(metric, p0, 64);
        const int mp1 = __shfl(metric, p1, 64);
        const int m0 = (mp0 < 0x7FFFFFF0) ? mp0 + bm0 : 0x7FFFFFFF;  // ref :826-827
        const int m1 = (mp1 < 0x7FFFFFF0) ? mp1 + bm1 : 0x7FFFFFFF;
        const bool take1 = !(m0 <= m1);                       // ref :829: ties keep p0
        metric = take1 ? m1 : m0;
        const unsigned long long word = __ballot(take1);
        if (lane == 0) s_dec[t] = word;
    }
    __syncthreads();

    // ---- best end state: first minimum (ref :835-837) ----------------------------------------
    int bm = metric, bs = s;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int om = __shfl_xor(bm, off, 64), os = __shfl_xor(bs, off, 64);
        if (om < bm || (om == bm && os < bs)) { bm = om; bs = os; }
    }

    // ---- traceback + pack + derandomise (ref :839-843, :878-895), uniform on all lanes ---------
    int cur = bs;
    for (int i = 0; i < OPV_FB; ++i) {
        unsigned byte = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = OPV_FBITS - 1 - 8 * i - j;
            byte |= (unsigned)(cur & 1) << j;                 // bits[t] = s % 2 -> bit j of byte i
            if (tb && lane == 0) tb[t] = (uint8_t)(cur & 1);
            const unsigned d = (unsigned)((s_dec[t] >> cur) & 1ull);
            cur = (cur >> 1) + (d ? 32 : 0);
        }
        if (lane == 0) s_out[i] = (uint8_t)(byte ^ kLfsr.b[i]);
    }
    __syncthreads();
    for (int i = lane; i < OPV_FB; i += 64) out[i] = s_out[i];
    if (lane == 0) *metric_out = bm;
}

constexpr int kDecodeLds = 17152 + 8576 + 2 * OPV_CODED + 144;

}  // namespace

// grid = (max new frames per stream, n_streams); frames dec_from .. n_frames-1 of each stream
extern "C" __global__ __launch_bounds__(64) void k_frame_decode(OpvStream* __restrict__ streams) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kDecodeLds];
    OpvStream& st = streams[blockIdx.y];
    const uint32_t f = st.dec_from + blockIdx.x;
    if (f >= st.n_frames) return;
    const uint32_t slot = f % st.cap_frames;  // frame records / frames / metrics are rings
    const OpvFrameRec rec = st.frec[slot];
    decode_one(st.soft, (uint32_t)rec.payload_sym, (uint32_t)(st.cap_soft - 1), st.frames + (size_t)slot * OPV_FB,
               st.metrics + slot, nullptr, nullptr, nullptr, lds);
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
