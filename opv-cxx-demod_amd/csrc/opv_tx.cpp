// opv_tx.cpp — host-side OPV transmit chain (signal source for tests, the CLI and the bench).
//
// Behaviourally identical, bit for bit, to the reference modulator program
// (reference src/opv-mod.cpp; behaviour spec in SURVEY.md Appendix A): BERT frame builder
// (:339-361), CCSDS randomiser (:97-113), K=7 r=1/2 encoder (:120-136, last byte first, MSB
// first :186-196), 67x32 interleaver with in-byte bit reversal (:142-153), sync word MSB
// first (:315-321) and the parallel-tone MSK modulator (:219-291) followed by 100 zero
// symbols (:528-529). Pinned by sha256 against `opv-mod` output (tests/test_capi_and_host.py).
//
// Design (not the reference's): the per-symbol tone/sign sequence and the NCO phases at each
// frame boundary are produced by one cheap sequential pass (the NCOs free-run with the
// reference's repeated-addition + wrap arithmetic, so the values are reproduced exactly);
// the expensive part — one libm sin/cos pair per sample for the ACTIVE tone only, the other
// tone contributes an exact +/-0 — then runs frame-parallel on host threads.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/opv_demod.h"
#include "opv_tx_internal.h"

namespace {

constexpr double kPi = 3.14159265358979323846;  // opv-mod.cpp:43
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kFs = 2168000.0;
constexpr double kDev = 54200.0 / 4.0;  // opv-mod.cpp:36-37
constexpr uint32_t kSync = 0x02B8DB;
constexpr int kSps = OPV_SAMPLES_PER_SYMBOL;

struct LfsrBytes {
    uint8_t b[OPV_FRAME_BYTES];
    LfsrBytes() {
        unsigned st = 0xFF;
        for (auto& o : b) {
            unsigned v = 0;
            for (int k = 0; k < 8; ++k) {
                v = (v << 1) | ((st >> 7) & 1u);
                st = ((st << 1) | (((st >> 7) ^ (st >> 6) ^ (st >> 4) ^ (st >> 2)) & 1u)) & 0xFFu;
            }
            o = (uint8_t)v;
        }
    }
};
const LfsrBytes kLfsr;

inline size_t interleave_pos(size_t i) {  // opv-mod.cpp:145-149
    const size_t p = (i % 32) * 67 + i / 32;
    return (p & ~size_t(7)) | (7 - (p & 7));
}

// 24 sync bits + 2144 interleaved coded bits of one frame, in on-air order; `linear` (optional): the coded bits in encoder order
void frame_symbols(const uint8_t* payload, uint8_t* sym /*2168*/, uint8_t* linear = nullptr /*2144*/) {
    for (int b = 0; b < 24; ++b) sym[b] = (kSync >> (23 - b)) & 1u;
    uint8_t* coded = sym + 24;
    unsigned sr = 0;
    size_t o = 0;
    for (int byte = OPV_FRAME_BYTES - 1; byte >= 0; --byte) {
        const unsigned v = payload[byte] ^ kLfsr.b[byte];
        for (int bit = 7; bit >= 0; --bit) {
            const unsigned in = (v >> bit) & 1u;
            const unsigned reg = (in << 6) | sr;
            const uint8_t g1 = (uint8_t)__builtin_parity(reg & 0x4F), g2 = (uint8_t)__builtin_parity(reg & 0x6D);
            if (linear) { linear[o] = g1; linear[o + 1] = g2; }
            coded[interleave_pos(o++)] = g1;
            coded[interleave_pos(o++)] = g2;
            sr = ((sr << 1) | in) & 0x3F;
        }
    }
}

inline void advance(double& ph, double inc) {  // opv-mod.cpp:274-279
    ph += inc;
    while (ph > kPi) ph -= kTwoPi;
    while (ph < -kPi) ph += kTwoPi;
}

struct FrameStart {
    double ph1, ph2;
};

}  // namespace

extern "C" void opv_tx_bert_frame(const char* callsign, uint32_t token, uint32_t frame_num,
                                  uint8_t out[OPV_FRAME_BYTES]) {
    std::memset(out, 0, OPV_FRAME_BYTES);
    size_t len = std::strlen(callsign);
    if (len > 9) len = 9;  // opv-mod.cpp:451-454
    uint64_t v = 0;
    for (size_t k = len; k-- > 0;) {  // first character least significant (opv-mod.cpp:66-70)
        const char c = callsign[k];
        unsigned d = 0;
        if (c >= 'A' && c <= 'Z') d = c - 'A' + 1;
        else if (c >= 'a' && c <= 'z') d = c - 'a' + 1;
        else if (c >= '0' && c <= '9') d = c - '0' + 27;
        else if (c == '-') d = 37;
        else if (c == '/') d = 38;
        else if (c == '.') d = 39;
        v = v * 40 + d;
    }
    for (int b = 0; b < 6; ++b) out[b] = (uint8_t)(v >> (40 - 8 * b));
    out[6] = (uint8_t)(token >> 16);
    out[7] = (uint8_t)(token >> 8);
    out[8] = (uint8_t)token;
    for (unsigned i = 0; i < OPV_FRAME_BYTES - 12; ++i) out[12 + i] = (uint8_t)(frame_num + i);
}

extern "C" void opv_tap_tx_frame(const uint8_t* frame134, uint8_t* randomized134, uint8_t* coded2144, uint8_t* interleaved2144) {
    if (!frame134) return;
    if (randomized134) for (int i = 0; i < OPV_FRAME_BYTES; ++i) randomized134[i] = frame134[i] ^ kLfsr.b[i];
    if (coded2144 || interleaved2144) {
        uint8_t sym[OPV_FRAME_SYMBOLS], lin[OPV_ENCODED_BITS];
        frame_symbols(frame134, sym, lin);
        if (coded2144) std::memcpy(coded2144, lin, OPV_ENCODED_BITS);
        if (interleaved2144) std::memcpy(interleaved2144, sym + 24, OPV_ENCODED_BITS);
    }
}

extern "C" void opv_tx_bert_frames(const char* callsign, uint32_t token, uint32_t first_frame, size_t n_frames,
                                   uint8_t* out) {
    for (size_t k = 0; k < n_frames; ++k) opv_tx_bert_frame(callsign, token, first_frame + (uint32_t)k, out + k * OPV_FRAME_BYTES);
}

extern "C" size_t opv_tx_modulated_samples(size_t n_frames) {
    return n_frames * (size_t)OPV_FRAME_SYMBOLS * kSps + 100u * kSps;
}

// ---- pieces shared with the device modulator (opv_tx_internal.h) ----------------------------
// per-symbol tone / sign of n_frames more frames of a run whose modulator stands at (T, bn) (opv-mod.cpp:228-257)
static void symbol_codes_from(int& T, int& bn, const uint8_t* frames, size_t n_frames, int8_t* amp) {
    std::vector<uint8_t> sym(OPV_FRAME_SYMBOLS);
    for (size_t f = 0; f < n_frames; ++f) {
        frame_symbols(frames + f * OPV_FRAME_BYTES, sym.data());
        for (int k = 0; k < OPV_FRAME_SYMBOLS; ++k) {
            const int d = sym[k] ? -1 : 1;
            int a;
            if (d == 1) a = T;                                  // tone 1, sign T (0 right after reset)
            else a = 2 * ((bn == 0) ? -T : T);                  // tone 2, sign +/-T by symbol parity
            amp[f * OPV_FRAME_SYMBOLS + k] = (int8_t)a;
            T = (T == 0) ? 1 : d * T;
            bn ^= 1;
        }
    }
}

void opv_tx_symbol_codes(const uint8_t* frames, size_t n_frames, int8_t* amp) {
    int T = 0, bn = 1;  // opv-mod.cpp:221-226: one modulator reset per run
    symbol_codes_from(T, bn, frames, n_frames, amp);
}

void opv_tx_symbol_phases(size_t first_symbol, size_t n_symbols, double* ph1_io, double* ph2_io, double* out2) {
    // free-running NCOs, the reference's repeated addition + wrap (opv-mod.cpp:274-279); data-independent
    const double inc1 = kTwoPi * (-kDev) / kFs, inc2 = kTwoPi * (+kDev) / kFs;
    double ph1 = *ph1_io, ph2 = *ph2_io;
    (void)first_symbol;
    for (size_t k = 0; k < n_symbols; ++k) {
        out2[2 * k] = ph1;
        out2[2 * k + 1] = ph2;
        for (int i = 0; i < kSps; ++i) { advance(ph1, inc1); advance(ph2, inc2); }
    }
    *ph1_io = ph1;
    *ph2_io = ph2;
}

#ifndef OPV_TX_NO_EMBEDDED_TABLE   // (the build-time tool that MAKES the table links this file without it)
void opv_tx_checkpoint_range(size_t first, size_t count, double* out2) {
    size_t n_tab = 0;
    const double* tab = opv_tx_checkpoints(&n_tab);
    size_t k = 0;
    for (; k < count && first + k < n_tab; ++k) { out2[2 * k] = tab[2 * (first + k)]; out2[2 * k + 1] = tab[2 * (first + k) + 1]; }
    if (k == count) return;
    // beyond the build-time table: the same recurrence, continued from its last entry, kept for the life of the process
    static std::mutex mu;
    static std::vector<double> ext;      // entries n_tab, n_tab + 1, ... (2 doubles each)
    static double ph1, ph2;              // state at entry n_tab + ext.size() / 2
    std::lock_guard<std::mutex> lock(mu);
    if (ext.empty() && n_tab) { ph1 = tab[2 * (n_tab - 1)]; ph2 = tab[2 * (n_tab - 1) + 1]; }
    std::vector<double> tmp(2 * OPV_TX_CKPT_SYMS);
    for (; k < count; ++k) {
        const size_t e = first + k - n_tab;                  // index into ext
        while (ext.size() / 2 <= e) {
            // entry n_tab + j is one checkpoint interval behind entry n_tab + j - 1 (the table's last entry for j = 0)
            opv_tx_symbol_phases(0, OPV_TX_CKPT_SYMS, &ph1, &ph2, tmp.data());
            ext.push_back(ph1);
            ext.push_back(ph2);
        }
        out2[2 * k] = ext[2 * e];
        out2[2 * k + 1] = ext[2 * e + 1];
    }
}

extern "C" void opv_tap_tx_checkpoints(size_t first, size_t count, double* out2) { opv_tx_checkpoint_range(first, count, out2); }
#endif

void opv_tx_sample_exact(double ph1_sym, double ph2_sym, int a, int i, int16_t* I, int16_t* Q) {
    const double inc1 = kTwoPi * (-kDev) / kFs, inc2 = kTwoPi * (+kDev) / kFs;
    double ph1 = ph1_sym, ph2 = ph2_sym;
    for (int k = 0; k < i; ++k) { advance(ph1, inc1); advance(ph2, inc2); }
    double vi = 0.0, vq = 0.0;
    if (a == 1 || a == -1) { vi = a * std::sin(ph1); vq = a * std::cos(ph1); }
    else if (a == 2 || a == -2) { const int sg = a / 2; vi = sg * std::sin(ph2); vq = sg * std::cos(ph2); }
    *I = (int16_t)(16383.0 * vi);
    *Q = (int16_t)(16383.0 * vq);
}

// The modulator object of the reference (HDLModulator, opv-mod.cpp:219-291): the two free-running NCOs, the differential
// sign T and the symbol parity b_n. A run is a reset, any number of frames, and - in `opv-mod` - 100 silent symbols.
struct opv_tx_stream {
    double ph1 = 0.0, ph2 = 0.0;
    int T = 0, bn = 1;   // :221-226
};

// n_frames more frames of a run: samples into iq (n_frames * 2168 * 40), the state advanced past them
static void modulate_frames(opv_tx_stream& st, const uint8_t* frames, size_t n_frames, int16_t* iq) {
    if (n_frames == 0) return;
    const size_t nsym = n_frames * OPV_FRAME_SYMBOLS;
    // pass 1 (sequential, cheap): per-symbol active tone + sign, NCO phases at frame starts
    std::vector<int8_t> amp(nsym);  // +/-1: tone 1 active with that sign; +/-2: tone 2; 0: silent
    std::vector<FrameStart> fs(n_frames);
    symbol_codes_from(st.T, st.bn, frames, n_frames, amp.data());
    {
        const double inc1 = kTwoPi * (-kDev) / kFs, inc2 = kTwoPi * (+kDev) / kFs;
        for (size_t f = 0; f < n_frames; ++f) {
            fs[f] = {st.ph1, st.ph2};
            for (int k = 0; k < OPV_FRAME_SYMBOLS * kSps; ++k) { advance(st.ph1, inc1); advance(st.ph2, inc2); }
        }
    }
    // pass 2 (frame-parallel): samples
    auto work = [&](size_t f0, size_t f1) {
        const double inc1 = kTwoPi * (-kDev) / kFs, inc2 = kTwoPi * (+kDev) / kFs;
        for (size_t f = f0; f < f1; ++f) {
            double ph1 = fs[f].ph1, ph2 = fs[f].ph2;
            int16_t* o = iq + f * (size_t)OPV_FRAME_SYMBOLS * kSps * 2;
            for (int k = 0; k < OPV_FRAME_SYMBOLS; ++k) {
                const int a = amp[f * OPV_FRAME_SYMBOLS + k];
                for (int i = 0; i < kSps; ++i) {
                    double I = 0.0, Q = 0.0;
                    if (a == 1 || a == -1) { I = a * std::sin(ph1); Q = a * std::cos(ph1); }
                    else if (a == 2 || a == -2) { const int s = a / 2; I = s * std::sin(ph2); Q = s * std::cos(ph2); }
                    *o++ = (int16_t)(16383.0 * I);  // truncation toward zero (opv-mod.cpp:271-272)
                    *o++ = (int16_t)(16383.0 * Q);
                    advance(ph1, inc1);
                    advance(ph2, inc2);
                }
            }
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 16) nt = 16;
    if (nt > n_frames) nt = (unsigned)n_frames;
    if (nt == 1) { work(0, n_frames); return; }         // (a live source hands over one frame at a time: no thread for it)
    std::vector<std::thread> pool;
    const size_t per = (n_frames + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t a = t * per, b = (a + per < n_frames) ? a + per : n_frames;
        if (a < b) pool.emplace_back(work, a, b);
    }
    for (auto& th : pool) th.join();
}

extern "C" size_t opv_tx_modulate(const uint8_t* frames, size_t n_frames, int16_t* iq) {
    opv_tx_stream st;
    modulate_frames(st, frames, n_frames, iq);
    const size_t body = n_frames * (size_t)OPV_FRAME_SYMBOLS * kSps;
    std::memset(iq + 2 * body, 0, sizeof(int16_t) * 2u * 100u * kSps);
    return body + 100u * kSps;
}

extern "C" opv_tx_stream* opv_tx_stream_create(void) { return new (std::nothrow) opv_tx_stream; }
extern "C" void opv_tx_stream_destroy(opv_tx_stream* st) { delete st; }
extern "C" void opv_tx_stream_reset(opv_tx_stream* st) { if (st) *st = opv_tx_stream{}; }
extern "C" size_t opv_tx_stream_frames(opv_tx_stream* st, const uint8_t* frames, size_t n_frames, int16_t* iq) {
    if (!st || !iq || (!frames && n_frames)) return 0;
    modulate_frames(*st, frames, n_frames, iq);
    return n_frames * (size_t)OPV_FRAME_SYMBOLS * kSps;
}
extern "C" size_t opv_tx_stream_tail(int16_t* iq) {
    if (!iq) return 0;
    std::memset(iq, 0, sizeof(int16_t) * 2u * 100u * kSps);
    return 100u * kSps;
}
