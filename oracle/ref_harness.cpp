// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-callable window onto the *real* reference classes. The reference translation
// unit is compiled from where it lies (the Makefile passes -I$(REF)/src, default
// /root/reference/src); nothing of it is copied into this repository. The resulting
// shared object goes to oracle/_ref/ (git-ignored) and is used to
//   (1) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py),
//   (2) pin oracle/opv_oracle.c bit-for-bit while /root/reference is present.
//
// Classes reached (reference file:line): MSKDemodulatorAFC src/opv-demod.cpp:108-348,
// CoherentMSKDemodulator :365-572,
// SyncTracker :587-787, deinterleave_addr :792-795, ViterbiDecoder :800-847,
// FrameDecoder :852-902.
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

// SyncTracker::process logs its state transitions with fprintf(stderr, ...)
// (src/opv-demod.cpp:651,677,695,699,705). Route that stream into a buffer we own so the
// exact text can be captured as a fixture instead of polluting the test runner's stderr.
static FILE* g_ref_log = nullptr;
static char* g_ref_log_buf = nullptr;
static size_t g_ref_log_len = 0;
static FILE* ref_log_stream() {
    if (!g_ref_log) g_ref_log = open_memstream(&g_ref_log_buf, &g_ref_log_len);
    return g_ref_log;
}
#undef stderr
#define stderr ref_log_stream()
#define main opv_reference_main
#include "opv-demod.cpp"
#undef main
#undef stderr

namespace {
std::vector<sample_t> widen(const int16_t* iq, size_t n) {
    std::vector<sample_t> v;
    v.reserve(n);
    for (size_t i = 0; i < n; ++i) v.push_back(sample_t(iq[2 * i], iq[2 * i + 1]));  // :1023
    return v;
}
}  // namespace

extern "C" {

// ---- log capture ---------------------------------------------------------------------
size_t ref_log_take(char* out, size_t cap) {
    FILE* f = ref_log_stream();
    fflush(f);
    size_t n = g_ref_log_len < cap ? g_ref_log_len : cap;
    if (out && n) memcpy(out, g_ref_log_buf, n);
    size_t total = g_ref_log_len;
    fclose(f);
    free(g_ref_log_buf);
    g_ref_log = nullptr; g_ref_log_buf = nullptr; g_ref_log_len = 0;
    return total;
}

// ---- MSKDemodulatorAFC ----------------------------------------------------------------
void* ref_demod_create() { return new MSKDemodulatorAFC(); }
void ref_demod_destroy(void* h) { delete static_cast<MSKDemodulatorAFC*>(h); }
void ref_demod_set_freq_offset(void* h, double hz) { static_cast<MSKDemodulatorAFC*>(h)->set_freq_offset(hz); }
void ref_demod_set_afc(void* h, double a) { static_cast<MSKDemodulatorAFC*>(h)->set_afc_bandwidth(a); }
double ref_demod_freq_offset(void* h) { return static_cast<MSKDemodulatorAFC*>(h)->get_freq_offset(); }
double ref_demod_timing_freq(void* h) { return static_cast<MSKDemodulatorAFC*>(h)->get_timing_freq(); }
size_t ref_demod_leftover(void* h) { return static_cast<MSKDemodulatorAFC*>(h)->get_leftover(); }

double ref_demod_estimate_offset(void* h, const int16_t* iq, size_t n) {
    auto v = widen(iq, n);
    return static_cast<MSKDemodulatorAFC*>(h)->estimate_offset(v.data(), v.size());
}

// returns the number of soft symbols produced; writes at most cap of them
size_t ref_demod_demodulate(void* h, const int16_t* iq, size_t n, double* soft, size_t cap) {
    auto v = widen(iq, n);
    std::vector<double> s;
    static_cast<MSKDemodulatorAFC*>(h)->demodulate(v.data(), v.size(), s);
    size_t m = s.size() < cap ? s.size() : cap;
    if (soft && m) memcpy(soft, s.data(), m * sizeof(double));
    return s.size();
}

// ---- CoherentMSKDemodulator (src/opv-demod.cpp:365-572) ---------------------------------
void* ref_coh_create() { return new CoherentMSKDemodulator(); }
void ref_coh_destroy(void* h) { delete static_cast<CoherentMSKDemodulator*>(h); }
void ref_coh_set_freq_offset(void* h, double hz) { static_cast<CoherentMSKDemodulator*>(h)->set_freq_offset(hz); }
void ref_coh_set_afc(void* h, double a) { static_cast<CoherentMSKDemodulator*>(h)->set_afc_bandwidth(a); }
void ref_coh_set_pll(void* h, double bw) { static_cast<CoherentMSKDemodulator*>(h)->set_pll_bandwidth(bw); }
double ref_coh_freq_offset(void* h) { return static_cast<CoherentMSKDemodulator*>(h)->get_freq_offset(); }
double ref_coh_estimate_offset(void* h, const int16_t* iq, size_t n) {
    auto v = widen(iq, n);
    return static_cast<CoherentMSKDemodulator*>(h)->estimate_offset(v.data(), v.size());
}
size_t ref_coh_demodulate(void* h, const int16_t* iq, size_t n, double* soft, size_t cap) {
    auto v = widen(iq, n);
    std::vector<double> s;
    static_cast<CoherentMSKDemodulator*>(h)->demodulate(v.data(), v.size(), s);
    size_t m = s.size() < cap ? s.size() : cap;
    if (soft && m) memcpy(soft, s.data(), m * sizeof(double));
    return s.size();
}

// ---- SyncTracker ----------------------------------------------------------------------
void* ref_tracker_create() { return new SyncTracker(); }
void ref_tracker_destroy(void* h) { delete static_cast<SyncTracker*>(h); }
int ref_tracker_state(void* h) { return (int)static_cast<SyncTracker*>(h)->get_state(); }
int ref_tracker_frames(void* h) { return static_cast<SyncTracker*>(h)->get_total_frames(); }
// returns 1 when a frame was released; payload (2144 doubles) and quality are then filled
int ref_tracker_process(void* h, double soft, size_t sym_idx, double* payload, double* quality) {
    auto r = static_cast<SyncTracker*>(h)->process(soft, sym_idx);
    if (r.frame_ready && !r.payload.empty()) {
        if (payload) memcpy(payload, r.payload.data(), r.payload.size() * sizeof(double));
        if (quality) *quality = r.sync_quality;
        return (int)r.payload.size();
    }
    return 0;
}

// ---- frame decoder pieces -------------------------------------------------------------
size_t ref_deinterleave_addr(size_t i) { return deinterleave_addr(i); }

int ref_viterbi(const int* in2144, uint8_t* bits1072) {
    std::array<int, ENCODED_BITS> in;
    std::array<uint8_t, FRAME_BITS> bits;
    memcpy(in.data(), in2144, sizeof(int) * ENCODED_BITS);
    ViterbiDecoder v;
    int m = v.decode(in, bits);
    memcpy(bits1072, bits.data(), FRAME_BITS);
    return m;
}

int ref_frame_decode(const double* soft2144, uint8_t* out134) {
    std::array<uint8_t, FRAME_BYTES> out;
    FrameDecoder fd;
    int m = fd.decode(soft2144, out);
    if (m >= 0) memcpy(out134, out.data(), FRAME_BYTES);
    return m;
}

}  // extern "C"
