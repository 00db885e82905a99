// k_tx_modulate.hip — the OPV transmit chain on the device: everything `opv-mod` does between a
// 134-byte frame and int16 I/Q (reference src/opv-mod.cpp:97-291), written for HBM. Four kernels:
//
//   k_tx_encode         one workgroup per FRAME: CCSDS randomiser (:97-113), K=7 r=1/2 encoder (:120-136; last byte first,
//                       MSB first :186-196), 67x32 interleaver with in-byte bit reversal (:142-153), sync word in front
//                       (:315-321) -> one code byte per on-air symbol: bit 0 = the symbol's bit, bit 1 = XOR of the frame's
//                       bits in front of it; + the frame's parity. The encoder is a sliding window (coded bit t depends on
//                       input bits t-6..t only), so all 1072 steps of a frame run in parallel.
//   k_tx_scan_frames    exclusive XOR scan of the frame parities: the modulator's differential sign T' = d T
//                       (:232-239) is a running +/-1 product over the whole run, i.e. a parity prefix.
//   k_tx_expand_phases  the two NCOs free-run through every sample with rounded additions (:274-279): data-independent but
//                       strictly sequential. Their state at every 128th symbol is tabulated at build time
//                       (tools/gen_tx_checkpoints.cpp); one thread per table entry replays the 128 x 40 additions that
//                       follow (bit-identical IEEE adds and wraps) and leaves (ph1, ph2) at every symbol start. Once per
//                       context and run length; every stream modulated afterwards shares it.
//   k_tx_modulate       one thread per symbol: tone / sign from the code byte and the frame prefix (:241-257), 40 samples
//                       with sin / cos of the ACTIVE tone only, truncation like the reference (:268-272), 64 symbols per
//                       workgroup staged through LDS so that HBM sees 16-byte-per-lane stores (4 B/sample, nothing else).
//
// Exactness: device sincos and glibc agree to ~1 ulp, so 16383*x can only truncate differently when it lies within ~1e-11
// of an integer. Such samples (none in practice) are reported in a small list and re-evaluated by the host with libm, so
// the result is identical to `opv-mod` by construction, not by luck.
// One place is not "in practice none": the FLAT TOPS. A symbol lasts a quarter period of either tone, so at every symbol
// start sin or cos of each NCO sits at +/-1 up to the phase's accumulated rounding drift eps (6.74e-12 rad per frame, from
// the recurrence itself: data-independent and monotone), i.e. at 1 - eps^2/2 - and 16383 x truncates to 16383 only if the
// library returns exactly 1.0. glibc does while eps^2/2 < 2^-54 (eps < 1.05e-8: 1563 frames into a run) and returns
// 1 - 2^-53 from there on (16382). The kernel therefore decides flat tops by the symbol index, not by its own sincos:
// 16383 before `flat_lo` (eps < 0.90e-8: within 0.37 ulp of 1.0), 16382 from `flat_hi` on (eps > 1.25e-8: 0.70 ulp below),
// and in the ~500 frames between by a bit per symbol and tone that the host made with libm itself (opv_capi.hip; once per
// context and run length). tests/test_capi_and_host.py::test_flat_top_zones_hold_for_this_libm scans both zones on the
// host's libm.
//
// Roofline: HBM write, 4 B/sample (a 1000-frame run: 347 MB); the phase table adds 16 B/symbol of reads (35 MB).
#include <hip/hip_runtime.h>
#include <math.h>

#include "opv_device.h"
#include "opv_tx_internal.h"

namespace {
constexpr double kPi = 3.14159265358979323846;  // opv-mod.cpp:43
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kFs = 2168000.0;
constexpr double kDev = 54200.0 / 4.0;
static_assert(((OPV_SYNC_WORD >> 23) & 1u) == 0u, "symbol 0 of a run (first sync bit) is taken to be 0: it never enters the sign product");
static_assert(OPV_FSYMS % 2 == 0, "b_n toggles per symbol: a frame's first symbol always sees b_n = 1");

__device__ inline void advance(double& ph, double inc) {  // opv-mod.cpp:274-279
    ph += inc;
    while (ph > kPi) ph -= kTwoPi;
    while (ph < -kPi) ph += kTwoPi;
}

// the randomiser's byte sequence (opv-mod.cpp:97-113: state 0xFF, taps 7 6 4 2, reset per frame): a constant table
struct LfsrTable { uint8_t b[OPV_FB]; };
constexpr LfsrTable make_lfsr() {
    LfsrTable t{};
    unsigned st = 0xFF;
    for (int i = 0; i < OPV_FB; ++i) {
        unsigned v = 0;
        for (int k = 0; k < 8; ++k) {
            v = (v << 1) | ((st >> 7) & 1u);
            st = ((st << 1) | (((st >> 7) ^ (st >> 6) ^ (st >> 4) ^ (st >> 2)) & 1u)) & 0xFFu;
        }
        t.b[i] = (uint8_t)v;
    }
    return t;
}
__constant__ LfsrTable kTxLfsr = make_lfsr();
}  // namespace

// frames: [n_frames][134]; codes: [n_frames * 2168] (bit 0: symbol bit, bit 1: parity of the frame's bits before it);
// frame_par: [n_frames] parity of all 2168 bits of the frame
extern "C" __global__ __launch_bounds__(256) void k_tx_encode(const uint8_t* __restrict__ frames, uint32_t n_frames,
                                                               uint8_t* __restrict__ codes, uint8_t* __restrict__ frame_par) {
    __shared__ uint8_t u[OPV_FBITS];                // encoder input bits in encoding order
    __shared__ uint8_t sym[OPV_FSYMS + 8];          // on-air bits: 24 sync + 2144 interleaved coded
    __shared__ uint32_t part[256];
    const uint32_t f = blockIdx.x;
    const int tid = threadIdx.x;
    if (f >= n_frames) return;
    const uint8_t* p = frames + (size_t)f * OPV_FB;
    if (tid < OPV_FB) {
        // byte 133 goes first, MSB first (opv-mod.cpp:186-196): input bit t = bit (7 - t % 8) of byte (133 - t / 8)
        const unsigned v = p[tid] ^ kTxLfsr.b[tid];
        const int t0 = 8 * (OPV_FB - 1 - tid);
#pragma unroll
        for (int b = 0; b < 8; ++b) u[t0 + b] = (uint8_t)((v >> (7 - b)) & 1u);
    }
    if (tid < OPV_SYNC_BITS) sym[tid] = (uint8_t)((OPV_SYNC_WORD >> (23 - tid)) & 1u);
    __syncthreads();
    for (int t = tid; t < OPV_FBITS; t += 256) {
        // reg = (in << 6) | sr, sr bit k = input bit t - 1 - k (zero before the frame: the encoder is reset per frame, :161)
        unsigned reg = (unsigned)u[t] << 6;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (t - 1 - k >= 0) reg |= (unsigned)u[t - 1 - k] << k;
        const unsigned g1 = __popc(reg & 0x4Fu) & 1u, g2 = __popc(reg & 0x6Du) & 1u;   // opv-mod.cpp:124-128
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned i = 2u * (unsigned)t + (unsigned)h;                        // coded bit index, order g1, g2
            const unsigned q = (i % 32u) * 67u + i / 32u;                            // opv-mod.cpp:145-149
            sym[OPV_SYNC_BITS + ((q & ~7u) | (7u - (q & 7u)))] = (uint8_t)(h ? g2 : g1);
        }
    }
    __syncthreads();
    // exclusive parity prefix over the frame's 2168 bits: 9 consecutive symbols per thread, scan of the 256 partials
    constexpr int kPer = (OPV_FSYMS + 255) / 256;
    const int k0 = tid * kPer;
    uint32_t mine = 0;
    for (int k = k0; k < k0 + kPer && k < OPV_FSYMS; ++k) mine ^= sym[k];
    part[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] ^= v;
        __syncthreads();
    }
    uint32_t run = part[tid] ^ mine;                // exclusive prefix of this thread's first symbol
    uint8_t* out = codes + (size_t)f * OPV_FSYMS;
    for (int k = k0; k < k0 + kPer && k < OPV_FSYMS; ++k) {
        out[k] = (uint8_t)(sym[k] | (run << 1));
        run ^= sym[k];
    }
    if (tid == 255) frame_par[f] = (uint8_t)part[255];
}

// in place: frame_par[f] <- XOR of the parities of frames 0..f-1 (one workgroup; runs of any length in slices of 1024)
extern "C" __global__ __launch_bounds__(1024) void k_tx_scan_frames(uint8_t* __restrict__ frame_par, uint32_t n_frames) {
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_frames; base += 1024u) {
        const uint32_t i = base + tid;
        const uint32_t own = i < n_frames ? frame_par[i] : 0u;
        s[tid] = own;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t v = tid >= off ? s[tid - off] : 0u;
            __syncthreads();
            s[tid] ^= v;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        if (i < n_frames) frame_par[i] = (uint8_t)(s[tid] ^ own ^ carry);
        __syncthreads();
        if (tid == 1023) carry_s = carry ^ s[1023];
        __syncthreads();
    }
}

// ckpt: [n_ckpt] (ph1, ph2) at symbol j * OPV_TX_CKPT_SYMS; phases: [nsym] (ph1, ph2) at every symbol start
extern "C" __global__ __launch_bounds__(64) void k_tx_expand_phases(const double2* __restrict__ ckpt, uint32_t n_ckpt,
                                                                     uint64_t first_ckpt, uint64_t nsym,
                                                                     double2* __restrict__ phases) {
    const uint64_t j = first_ckpt + (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (j >= n_ckpt) return;
    const double inc1 = kTwoPi * (-kDev) / kFs, inc2 = kTwoPi * (+kDev) / kFs;  // opv-mod.cpp:259-260
    double2 p = ckpt[j];
    const uint64_t s0 = j * OPV_TX_CKPT_SYMS;
    for (uint32_t s = 0; s < OPV_TX_CKPT_SYMS && s0 + s < nsym; ++s) {
        phases[s0 + s] = p;
        for (int i = 0; i < OPV_SPS; ++i) { advance(p.x, inc1); advance(p.y, inc2); }
    }
}

// codes: [nsym] from k_tx_encode, frame_pre: [n_frames] from k_tx_scan_frames, phases: [nsym] (ph1, ph2) pairs;
// symbols nsym .. nsym_total-1 are the silent tail (opv-mod.cpp:528-529); out: packed int16 I | Q << 16
extern "C" __global__ __launch_bounds__(64) void k_tx_modulate(const uint8_t* __restrict__ codes,
                                                                const uint8_t* __restrict__ frame_pre,
                                                                const double2* __restrict__ phases, uint64_t nsym,
                                                                uint64_t nsym_total, int* __restrict__ out,
                                                                uint32_t* __restrict__ amb_count,
                                                                uint64_t* __restrict__ amb_list, uint32_t amb_cap,
                                                                uint64_t flat_lo, uint64_t flat_hi,
                                                                const uint8_t* __restrict__ flat_bits) {
    __shared__ __attribute__((aligned(16))) int stage[64 * OPV_SPS];
    const uint64_t sym0 = (uint64_t)blockIdx.x * 64u;
    const uint64_t sym = sym0 + threadIdx.x;
    const double inc1 = kTwoPi * (-kDev) / kFs, inc2 = kTwoPi * (+kDev) / kFs;  // opv-mod.cpp:259-260
    int a = 0;                                      // +/-1: tone 1 with that sign, +/-2: tone 2, 0: silent
    double ph1 = 0.0, ph2 = 0.0;
    if (sym < nsym && sym != 0) {                   // (symbol 0: T = 0 right after the modulator's reset, :221-226 - silent)
        const unsigned c = codes[sym];
        // T(k) = prod_{0<j<k} d_j, d = -1 for a 1 bit (:232-239): its sign is the parity of the bits in front of the symbol
        const int T = ((frame_pre[sym / OPV_FSYMS] ^ (c >> 1)) & 1u) ? -1 : 1;
        if ((c & 1u) == 0u) a = T;                                          // bit 0: d_s1 = T (:241-245)
        else a = 2 * ((sym & 1u) ? -T : T);                                 // bit 1: d_s2 = -/+T by b_n (:246-257); b_n = 0 on odd symbols
        const double2 p = phases[sym];
        ph1 = p.x; ph2 = p.y;
    }
    int* mine = stage + threadIdx.x * OPV_SPS;
    if (a == 0) {
        for (int i = 0; i < OPV_SPS; ++i) mine[i] = 0;
    } else {
        const bool tone1 = (a == 1 || a == -1);
        const double sgn = (a > 0) ? 1.0 : -1.0;
        for (int i = 0; i < OPV_SPS; ++i) {
            double sn, cs;
            sincos(tone1 ? ph1 : ph2, &sn, &cs);
            const double vi = 16383.0 * (sgn * sn), vq = 16383.0 * (sgn * cs);  // opv-mod.cpp:268-272
            int I = (int)vi, Q = (int)vq;                                         // truncation toward zero
            // a flat top (see the file comment): |sin| or |cos| = 1 - eps^2/2; whether libm's value IS 1.0 is a function
            // of the symbol index
            const bool top_i = fabs(vi) > 16382.5, top_q = fabs(vq) > 16382.5;
            if (top_i | top_q) {
                int mag = 16383;
                if (sym >= flat_hi) mag = 16382;
                else if (sym >= flat_lo) mag = ((flat_bits[sym - flat_lo] >> (tone1 ? 0 : 1)) & 1u) ? 16383 : 16382;
                if (top_i) I = vi < 0.0 ? -mag : mag;
                else Q = vq < 0.0 ? -mag : mag;
            }
            mine[i] = (I & 0xFFFF) | (Q << 16);
            // Could libm's value truncate differently? Only if v sits within the two libraries' ~1e-11
            // disagreement of a NON-ZERO integer without being exactly on it (|v| < 1 truncates to 0
            // from either side); the flat tops are decided above.
            const double ri = rint(vi), rq = rint(vq);
            if ((!top_i && ri != 0.0 && vi != ri && fabs(vi - ri) < 1e-9) || (!top_q && rq != 0.0 && vq != rq && fabs(vq - rq) < 1e-9)) {
                const uint32_t k = atomicAdd(amb_count, 1u);
                if (k < amb_cap) amb_list[k] = sym * OPV_SPS + (uint64_t)i;
            }
            advance(ph1, inc1);
            advance(ph2, inc2);
        }
    }
    __syncthreads();
    // 64 symbols x 40 samples = 2560 dwords = 10 x (64 lanes x 16 B)
    const uint64_t base = sym0 * OPV_SPS;
    const uint64_t total = nsym_total * OPV_SPS;
    const int4* s4 = reinterpret_cast<const int4*>(stage);
    int4* o4 = reinterpret_cast<int4*>(out + base);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const int q = r * 64 + threadIdx.x;
        if (base + 4u * q + 3u < total) o4[q] = s4[q];
    }
}
