// tests/san/fake_device.cpp — TEST INFRASTRUCTURE ONLY: the device half of include/opv_demod.h without a device.
//
// The sanitizer job (make -C opv-cxx-demod_amd san; tests/test_sanitizers.py) builds the product's HOST code - host/opv_demod_main.cpp,
// host/opv_mod_main.cpp, host/opv_rx_bridge.cpp, csrc/opv_tx.cpp - with -fsanitize=address,undefined (and once more with
// -fsanitize=thread) and needs something behind opv_create / opv_push_iq / opv_process / opv_pop_frames that is not a GPU: GPU
// AddressSanitizer is not available on the pool, and the host code is what parses argv, pipes and UDP datagrams from outside.
// This file is that something: every stream collects what is pushed, and the round after its flush runs the CPU oracle's
// whole receiver (oracle/opv_oracle.c: oro_receive, also compiled with the sanitizers) over it; frames, tracker events, the
// chunk log and the state then come back through the same entry points. Results therefore appear at end of stream only (the
// real library releases them as chunks complete), but they are the real ones: the sanitized opv-demod's stdout and stderr equal
// the reference-made fixtures. The four HIP runtime calls the bridge makes for its gather are host memory here.
//
// Never linked into the product: libopv_demod_hip.so has no CPU path (opv_create -> OPV_ENODEV without a gfx950 device).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/opv_demod.h"
#include "../../oracle/opv_oracle.h"

struct FakeStream {
    std::vector<int16_t> iq;
    bool flushed = false, done = false;
    std::vector<uint8_t> frames;
    std::vector<opv_frame_meta> meta;
    std::vector<opv_event> events;
    std::vector<double> chunks;      // 5 per demodulate() call
    size_t frames_popped = 0, events_popped = 0;
    opv_stream_state st{}, st_final{};
    // OPV_FAKE_STALL: the stream ends its first round held back by "back-pressure" (opv_stream_state.stalled bit 1) with half
    // of its frames visible, and finishes in the round after they have been popped - the path host code takes when a ring fills
    size_t vis_frames = 0, vis_events = 0, vis_chunks = 0;
    bool held = false;
};

struct opv_ctx {
    opv_cfg cfg{};
    std::vector<FakeStream> s;
    std::vector<uint8_t> dev_frames;   // [S][cap][134] view for the gather
    std::vector<int32_t> dev_counts;
    size_t cap = 0;
};

namespace {
thread_local std::string g_err;
int fail(int code, const char* what) { g_err = what; return code; }

void run_stream(opv_ctx* c, FakeStream& f) {
    const size_t n = f.iq.size() / 2;
    const size_t cap_frames = n / (OPV_FRAME_SYMBOLS * 38) + 8, cap_ev = 4 * cap_frames + 64, cap_chunks = n / 80000 + 4;
    std::vector<uint8_t> frames(cap_frames * OPV_FRAME_BYTES);
    std::vector<int32_t> metrics(cap_frames);
    std::vector<double> quality(cap_frames);
    std::vector<uint64_t> fsym(cap_frames);
    std::vector<oro_event> ev(cap_ev);
    f.chunks.assign(cap_chunks * 5, 0.0);
    oro_rx_cfg rc{};
    rc.streaming = c->cfg.streaming;
    rc.have_init_offset = c->cfg.have_init_offset;
    rc.init_offset = c->cfg.init_offset_hz;
    rc.afc_alpha = c->cfg.afc_alpha;
    rc.coherent = c->cfg.coherent;
    rc.pll_bw = c->cfg.pll_bw_hz;
    oro_rx_out out{};
    out.frames = frames.data(); out.metrics = metrics.data(); out.quality = quality.data(); out.frame_sym = fsym.data();
    out.cap_frames = cap_frames;
    out.events = ev.data(); out.cap_events = cap_ev;
    out.chunk_state = f.chunks.data(); out.cap_chunks = cap_chunks;
    static const int16_t none[2] = {0, 0};
    oro_receive(n ? f.iq.data() : none, n, &rc, &out);
    f.frames.assign(frames.begin(), frames.begin() + out.n_frames * OPV_FRAME_BYTES);
    f.meta.resize(out.n_frames);
    for (size_t k = 0; k < out.n_frames; ++k) {
        f.meta[k] = opv_frame_meta{metrics[k], 1, quality[k], fsym[k], fsym[k] >= OPV_ENCODED_BITS ? fsym[k] - OPV_ENCODED_BITS : 0};
    }
    f.events.resize(out.n_events);
    for (size_t k = 0; k < out.n_events; ++k) f.events[k] = opv_event{ev[k].kind, ev[k].count, ev[k].sym_idx, ev[k].corr, ev[k].raw};
    f.chunks.resize(out.n_chunks * 5);
    f.st.freq_offset_hz = out.final_freq_offset;
    f.st.timing_freq = out.final_timing_freq;
    f.st.est_offset_hz = out.est_offset;
    f.st.total_symbols = out.n_soft;
    f.st.total_samples = n;
    f.st.sync_state = out.final_state;
    f.st.frames_released = f.st.frames_decoded = (int32_t)out.n_frames;
    f.st.frames_perfect = (int32_t)out.n_perfect;
    f.st.n_chunks = (int32_t)out.n_chunks;
    f.done = true;
    f.vis_frames = f.meta.size(); f.vis_events = f.events.size(); f.vis_chunks = out.n_chunks;
    if (getenv("OPV_FAKE_STALL") && f.meta.size() >= 2) {
        f.st_final = f.st;
        f.held = true;
        f.vis_frames = f.meta.size() / 2;
        const uint64_t upto = f.meta[f.vis_frames - 1].release_symbol;
        f.vis_events = 0;
        while (f.vis_events < f.events.size() && f.events[f.vis_events].sym_idx <= upto) ++f.vis_events;
        f.vis_chunks = 0;
        f.st.stalled = 2;
        f.st.total_symbols = upto + 1;
        f.st.frames_released = f.st.frames_decoded = (int32_t)f.vis_frames;
        f.st.n_chunks = 0;
        f.st.freq_offset_hz = 12345.6;              // (an intermediate value nobody should print)
    }
}

FakeStream* stream_of(opv_ctx* c, int stream) {
    if (!c || stream < 0 || (size_t)stream >= c->s.size()) { fail(OPV_EINVAL, "bad context or stream index"); return nullptr; }
    return &c->s[(size_t)stream];
}
}  // namespace

extern "C" {

int opv_abi_version(void) { return OPV_ABI_VERSION; }
const char* opv_last_error(void) { return g_err.c_str(); }

int opv_create(opv_ctx** out, int n_streams, const opv_cfg* cfg) {
    if (!out || !cfg || n_streams < 1 || cfg->max_samples == 0 || cfg->max_samples >= (1ull << 31)) return fail(OPV_EINVAL, "opv_create: bad arguments");
    if (getenv("OPV_FAKE_NODEV")) return fail(OPV_ENODEV, "no HIP device (fake)");
    opv_ctx* c = new opv_ctx;
    c->cfg = *cfg;
    c->s.resize((size_t)n_streams);
    for (auto& f : c->s) f.st.est_offset_hz = NAN;
    *out = c;
    return OPV_OK;
}
void opv_destroy(opv_ctx* c) { delete c; }

int opv_push_iq(opv_ctx* c, int stream, const int16_t* iq, size_t n) {
    FakeStream* f = stream_of(c, stream);
    if (!f) return OPV_EINVAL;
    if (f->flushed) return fail(OPV_ESTATE, "push after flush");
    if (n && !iq) return fail(OPV_EINVAL, "null samples");
    f->iq.insert(f->iq.end(), iq, iq + 2 * n);            // reads every byte the host says is there: ASan checks the host's buffers
    return OPV_OK;
}
int opv_push_iq_batch(opv_ctx* c, int count, const int* streams, const int16_t* const* iq, const size_t* n) {
    for (int i = 0; i < count; ++i)
        if (int r = opv_push_iq(c, streams[i], iq[i], n[i])) return r;
    return OPV_OK;
}
int opv_push_iq_batch_async(opv_ctx* c, int count, const int* streams, const int16_t* const* iq, const size_t* n) {
    return opv_push_iq_batch(c, count, streams, iq, n);   // (the stand-in has copied everything when it returns)
}
int opv_push_wait(opv_ctx* c) { return c ? OPV_OK : OPV_EINVAL; }
int opv_flush(opv_ctx* c, int stream) {
    FakeStream* f = stream_of(c, stream);
    if (!f) return OPV_EINVAL;
    f->flushed = true;
    return OPV_OK;
}
int opv_process(opv_ctx* c) {
    if (!c) return fail(OPV_EINVAL, "null context");
    for (auto& f : c->s) {
        if (f.flushed && !f.done) run_stream(c, f);
        else if (f.held && f.frames_popped == f.vis_frames) {      // the frames were popped: the stream resumes and finishes
            f.held = false;
            f.st = f.st_final;
            f.vis_frames = f.meta.size(); f.vis_events = f.events.size(); f.vis_chunks = f.chunks.size() / 5;
        }
    }
    return OPV_OK;
}
int opv_sync(opv_ctx* c) { return c ? OPV_OK : OPV_EINVAL; }

long opv_pop_frames(opv_ctx* c, int stream, uint8_t* out134, size_t cap, opv_frame_meta* meta) {
    FakeStream* f = stream_of(c, stream);
    if (!f) return OPV_EINVAL;
    size_t n = f->vis_frames - f->frames_popped;
    if (n > cap) n = cap;
    if (n) memcpy(out134, f->frames.data() + f->frames_popped * OPV_FRAME_BYTES, n * OPV_FRAME_BYTES);
    for (size_t k = 0; meta && k < n; ++k) meta[k] = f->meta[f->frames_popped + k];
    f->frames_popped += n;
    return (long)n;
}
long opv_pop_events(opv_ctx* c, int stream, opv_event* out, size_t cap) {
    FakeStream* f = stream_of(c, stream);
    if (!f) return OPV_EINVAL;
    size_t n = f->vis_events - f->events_popped;
    if (n > cap) n = cap;
    for (size_t k = 0; k < n; ++k) out[k] = f->events[f->events_popped + k];
    f->events_popped += n;
    return (long)n;
}
int opv_get_state(opv_ctx* c, int stream, opv_stream_state* out) {
    FakeStream* f = stream_of(c, stream);
    if (!f || !out) return OPV_EINVAL;
    *out = f->st;
    out->flushed = f->flushed;
    return OPV_OK;
}
long opv_tap_chunks(opv_ctx* c, int stream, uint32_t first, double* out5, size_t cap) {
    FakeStream* f = stream_of(c, stream);
    if (!f) return OPV_EINVAL;
    const size_t have = f->vis_chunks;
    if (first >= have) return 0;
    const size_t n = have - first < cap ? have - first : cap;
    memcpy(out5, f->chunks.data() + 5 * (size_t)first, n * 5 * sizeof(double));
    return (long)n;
}

// ---- the bridge's --gather leg: "device" buffers are host memory, the collective is a copy
int opv_device_frames(opv_ctx* c, const uint8_t** d_frames, const int32_t** d_metrics, const int32_t** d_counts, size_t* cap) {
    if (!c) return OPV_EINVAL;
    if (!c->cap) {
        c->cap = (size_t)(c->cfg.max_samples / (uint64_t)(OPV_FRAME_SYMBOLS * 38) + 4);
        c->dev_frames.assign(c->s.size() * c->cap * OPV_FRAME_BYTES, 0);
        c->dev_counts.assign(c->s.size(), 0);
    }
    for (size_t k = 0; k < c->s.size(); ++k) {
        const FakeStream& f = c->s[k];
        c->dev_counts[k] = f.st.frames_released;
        for (size_t j = 0; j < f.meta.size(); ++j)          // a ring of the most recent `cap` frames, like the library's
            memcpy(&c->dev_frames[(k * c->cap + j % c->cap) * OPV_FRAME_BYTES], &f.frames[j * OPV_FRAME_BYTES], OPV_FRAME_BYTES);
    }
    if (d_frames) *d_frames = c->dev_frames.data();
    if (d_metrics) *d_metrics = nullptr;
    if (d_counts) *d_counts = c->dev_counts.data();
    if (cap) *cap = c->cap;
    return OPV_OK;
}
int opv_comm_init_all(void** comms, int n, const int* devices) {
    for (int i = 0; i < n; ++i) comms[i] = new int(devices[i]);
    return OPV_OK;
}
void opv_comm_destroy(void* comm) { delete static_cast<int*>(comm); }
int opv_gather_frames_all(opv_ctx* const* ctxs, void* const* comms, int n, int root, uint8_t* d_frames_all, int32_t* d_counts_all) {
    (void)comms; (void)root;
    for (int r = 0; r < n; ++r) {
        const uint8_t* fr; const int32_t* cnt; size_t cap;
        opv_device_frames(ctxs[r], &fr, nullptr, &cnt, &cap);
        const size_t S = ctxs[r]->s.size();
        memcpy(d_frames_all + (size_t)r * S * cap * OPV_FRAME_BYTES, fr, S * cap * OPV_FRAME_BYTES);
        memcpy(d_counts_all + (size_t)r * S, cnt, S * sizeof(int32_t));
    }
    return OPV_OK;
}

// the device transmit chain of `opv-mod -G`: the host modulator (csrc/opv_tx.cpp, the same bytes)
long opv_tx_modulate_device_to_host(opv_ctx* c, const uint8_t* frames134, size_t n_frames, int16_t* iq_out) {
    if (!c) return OPV_EINVAL;
    opv_tx_modulate(frames134, n_frames, iq_out);
    return 0;
}

// ---- the HIP runtime calls host/opv_rx_bridge.cpp makes itself
int hipSetDevice(int) { return 0; }
int hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
int hipHostFree(void* p) { free(p); return 0; }
int hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
int hipFree(void* p) { free(p); return 0; }
int hipMemcpy(void* dst, const void* src, size_t n, int) { memcpy(dst, src, n); return 0; }

}  // extern "C"
