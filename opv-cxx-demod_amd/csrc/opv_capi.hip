// opv_capi.hip — the thin C-ABI HIP shim behind include/opv_demod.h.
//
// Owns device memory and the per-stream device contexts, stages host IQ, and launches the
// four hot-path kernels on one HIP stream. There is NO CPU implementation of the hot path in
// this library: without a usable HIP device opv_create fails with OPV_ENODEV.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <atomic>

#include <climits>
#include <cstddef>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/opv_demod.h"
#include "opv_device.h"
#include "opv_offset_host.h"
#include "opv_tx_internal.h"

extern "C" __global__ void k_offset_search(OpvStream*, OpvGlobalCfg, const double*, uint32_t*);
extern "C" __global__ void k_tie_collect(const OpvStream*, const uint32_t*, uint32_t, uint32_t, uint32_t, OpvTieStage*);
extern "C" __global__ void k_tie_apply(OpvStream*, const OpvTieStage*, uint32_t, int);
extern "C" __global__ void k_msk_frontend_rb(OpvStream*, OpvGlobalCfg, int);
extern "C" __global__ void k_msk_frontend_rb_wg4(OpvStream*, OpvGlobalCfg, int);
extern "C" __global__ void k_msk_frontend_x4_wg4(OpvStream*, OpvGlobalCfg, int);
extern "C" __global__ void k_msk_frontend_x16(OpvStream*, OpvGlobalCfg, int);
extern "C" __global__ void k_msk_frontend_x16_wg4(OpvStream*, OpvGlobalCfg, int);
extern "C" __global__ void k_msk_frontend_x16_wg8(OpvStream*, OpvGlobalCfg, int);
extern "C" __global__ void k_coherent_frontend(OpvStream*, double, double);
extern "C" __global__ void k_sync_track(OpvStream*);
extern "C" __global__ void k_frame_scale(OpvStream*, uint32_t, uint32_t);
extern "C" __global__ void k_frame_scale_wave(OpvStream*, uint32_t);
extern "C" __global__ void k_frame_decode(OpvStream*, uint32_t);
extern "C" __global__ void k_payload_scale(const double*, uint32_t, double*);
extern "C" __global__ void k_decode_payloads(const double*, uint32_t, const double*, uint8_t*, int32_t*, int8_t*, int8_t*, uint8_t*);
extern "C" __global__ void k_channel(const int4*, int4*, uint64_t, double, double, double, uint64_t);
extern "C" __global__ void k_resample_clock(const int*, uint64_t, int*, uint64_t, double);
extern "C" __global__ void k_tx_encode(const uint8_t*, uint32_t, uint8_t*, uint8_t*);
extern "C" __global__ void k_tx_scan_frames(uint8_t*, uint32_t);
extern "C" __global__ void k_tx_expand_phases(const double2*, uint32_t, uint64_t, uint64_t, double2*);
extern "C" __global__ void k_tx_modulate(const uint8_t*, const uint8_t*, const double2*, uint64_t, uint64_t, int*, uint32_t*, uint64_t*, uint32_t,
                                         uint64_t, uint64_t, const uint8_t*);

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                              \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return fail(OPV_EHIP, #expr, _e);   \
    } while (0)

// one wave per stream, four of them per workgroup: from the stream count at which single-wave workgroups start to
// double up on SIMDs (k_frontend.hip: msk_frontend_body)
constexpr uint64_t kScaleWaveMaxFrames = 4096;    // frames (upper estimate) per round up to which the scale pre-pass runs one wave per frame
constexpr int kFrontendWg4MinStreams = 512;
constexpr int kFrontendX16MinStreams = 8192;      // measured on MI355X (NOTEBOOK.md §3.1): 8192 streams x 16 frames: four per wave 260 GS/s, sixteen per wave 247 (512 waves: half the SIMDs idle;
                                                  // with 7 frames per stream, bench.py's sweep, 221 / 255: the cross-over IS about 8192); 16 384 x 8: 187 / 478; 32 768 x 8: 204 / 574
constexpr int kFrontendX16Wg8MinStreams = 16384;  // 1024 waves of sixteen streams = one per SIMD; beyond that eight waves (two per SIMD) per workgroup
constexpr int kFrontendX4MinStreams = 2049;      // measured on MI355X (NOTEBOOK.md §3.1): one wave per stream runs 1024 streams at a time (46 ms per 2048 x 30 frames, 69 ms from 2049 on), four per wave 4096 (47.5 ms)

struct StreamIn {  // host -> device per-round update
    const int16_t* iq;
    uint64_t n_avail;
    int32_t eof;
    int32_t dirty;
    uint32_t frames_popped, events_popped;  // consumer cursors of the record rings
};

__global__ void k_apply_inputs(OpvStream* streams, const StreamIn* in, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (in[i].dirty) {
            streams[i].iq = in[i].iq;
            streams[i].n_avail = in[i].n_avail;
            streams[i].eof = in[i].eof;
        }
        streams[i].frames_popped = in[i].frames_popped;
        streams[i].events_popped = in[i].events_popped;
    }
}

// per round: frames released so far per stream (the zero-copy consumers' counts), and whether ANY stream ended the round
// held back by back-pressure, OR-ed into a pinned host word (opv_ctx::h_stall) that the next opv_process reads without a copy
__global__ void k_collect_counts(const OpvStream* streams, int32_t* counts, int n, int* any_stalled) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        counts[i] = (int32_t)streams[i].n_frames;
        if (streams[i].stalled) *(volatile int*)any_stalled = 1;   // (every writer stores the same 1: no atomic needed across PCIe)
    }
}

// ---- the serving path's two bulk moves (a live server pushes one 40 ms chunk per stream and round: N separate 347 KB blocks) ----
// k_push_gather: host -> device for MANY streams in one launch. The sources are the caller's buffers in pinned host memory, read
// through their device-visible addresses (16 B per lane over PCIe); the table itself lives in pinned memory too. One
// hipMemcpyAsync per stream moves 1536 such blocks at 21 GB/s (per-copy overhead), this kernel at 57 - the link's rate for one
// large copy (scripts/microbench/h2d_many.hip). A pair is split into `slices` work items so that few streams still keep
// enough loads in flight.
struct PushPair {
    const void* src;     // device-visible address of the caller's samples
    void* dst;           // where they go in the stream's device buffer
    uint32_t bytes;      // multiple of 4 (one sample)
    uint32_t wide;       // 1: src and dst are 16-byte aligned (int4 moves + a tail of ints), 0: 4-byte moves
};
__global__ __launch_bounds__(256) void k_push_gather(const PushPair* tab, uint32_t n_pairs, uint32_t slices) {
    for (uint32_t w = blockIdx.x; w < n_pairs * slices; w += gridDim.x) {
        const PushPair p = tab[w / slices];
        const uint32_t sl = w % slices;
        if (p.wide) {
            const uint32_t quads = p.bytes >> 4, per = (quads + slices - 1) / slices;
            const uint32_t lo = sl * per, hi = lo + per < quads ? lo + per : quads;
            const int4* s4 = (const int4*)p.src;
            int4* d4 = (int4*)p.dst;
            for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) d4[i] = s4[i];
            if (sl == 0) {                                  // at most three samples behind the last full quad
                const uint32_t rem = (p.bytes & 15u) >> 2;
                if (threadIdx.x < rem) ((int*)p.dst)[quads * 4 + threadIdx.x] = ((const int*)p.src)[quads * 4 + threadIdx.x];
            }
        } else {
            const uint32_t words = p.bytes >> 2, per = (words + slices - 1) / slices;
            const uint32_t lo = sl * per, hi = lo + per < words ? lo + per : words;
            for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) ((int*)p.dst)[i] = ((const int*)p.src)[i];
        }
    }
}
// k_compact: the retained tails of MANY streams' staging buffers across to their second buffers in one launch, and the streams'
// device contexts re-based (what compact_stream does for one stream with two copies). Exactly `samples` samples move: what lies
// behind them belongs to the pushes that follow on the copy stream.
struct CompactItem {
    const int* src;      // retained tail in the full buffer (16-byte aligned: `keep` is a multiple of 4 samples)
    int* dst;            // head of the other buffer
    uint32_t samples;
    uint32_t stream;
    uint64_t keep;       // samples dropped in front
};
__global__ __launch_bounds__(256) void k_compact(OpvStream* streams, const CompactItem* items, uint32_t n_items, uint32_t slices) {
    for (uint32_t w = blockIdx.x; w < n_items * slices; w += gridDim.x) {
        const CompactItem it = items[w / slices];
        const uint32_t sl = w % slices;
        const uint32_t quads = it.samples >> 2, per = (quads + slices - 1) / slices;
        const uint32_t lo = sl * per, hi = lo + per < quads ? lo + per : quads;
        const int4* s4 = (const int4*)it.src;
        int4* d4 = (int4*)it.dst;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) d4[i] = s4[i];
        if (sl == 0) {
            if (threadIdx.x < (it.samples & 3u)) it.dst[quads * 4 + threadIdx.x] = it.src[quads * 4 + threadIdx.x];
            if (threadIdx.x == 0) {
                OpvStream& st = streams[it.stream];
                st.iq = (const int16_t*)it.dst;
                st.origin -= it.keep;
                st.n_avail -= it.keep;
                st.iq_base += it.keep;
            }
        }
    }
}

__global__ void k_fill_i32(int32_t* p, int32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

struct HostStream {
    int16_t* d_iq_owned = nullptr;  // push path buffer
    int16_t* d_iq_alt = nullptr;    // second buffer of the same size: compaction copies the retained tail across and swaps
    size_t iq_cap = 0;              // samples
    const int16_t* d_iq = nullptr;  // what the kernels read
    uint64_t n_avail = 0;
    int eof = 0;
    bool dirty = false;
    bool attached = false;
    bool search_seen = false;       // a round in which k_offset_search could run for this stream has been launched
    uint64_t last_round_avail = 0;
    uint32_t popped = 0;        // frame records already handed out
    uint32_t events_popped = 0;
    uint32_t events_dropped = 0;       // overwritten in the (lossy) event ring before they were read
    int32_t decoded = 0, perfect = 0;  // over popped frames (`decoded` / `perfect` of main(), ref :1053-1054)
};

}  // namespace

struct opv_ctx {
    int n_streams = 0;
    opv_cfg cfg{};
    hipStream_t stream = nullptr;       // kernels, in order
    hipStream_t copy_stream = nullptr;  // host -> device IQ of opv_push_iq: overlaps the kernels of the previous round
    OpvStream* d_streams = nullptr;
    StreamIn* d_in = nullptr;
    // per-round stream updates travel through pinned host memory (no host sync in opv_process):
    // kInSlots rounds may be in flight before the host has to wait for the oldest upload
    static constexpr int kInSlots = 4;
    StreamIn* h_in = nullptr;           // kInSlots x n_streams, pinned
    hipEvent_t in_ev[kInSlots] = {};
    int* h_stall = nullptr;             // kInSlots words, pinned: round r's "a stream ended stalled" (k_collect_counts), slot r % kInSlots
    hipEvent_t done_ev = nullptr;       // behind the last round's k_collect_counts
    unsigned round_no = 0;
    std::vector<OpvStream> mirror;  // host copy, refreshed by refresh()
    std::vector<OpvStream> initial; // as created (for reset)
    std::vector<HostStream> hs;
    // pooled logs
    double* d_soft = nullptr;
    OpvFrameRec* d_frec = nullptr;
    OpvEventRec* d_events = nullptr;
    double* d_chunks = nullptr;
    uint8_t* d_frames = nullptr;
    int32_t* d_metrics = nullptr;
    double* d_fscale = nullptr;
    int32_t* d_counts = nullptr;
    double* d_offs_wtab = nullptr;      // k_offset_search's moment weights: [40 taps][cos, sin][OPV_OFFS_TERMS]
    // offset-search ties are decided with the HOST's libm (opv_offset_host.cpp), in stream order: streams whose search could not
    // decide put themselves on d_tie_list ([0] = count, then indices); behind the search opv_process enqueues, per pass of
    // tie_slots streams, k_tie_collect -> a host function -> k_tie_apply (TieWork below), and the front-end behind those
    uint32_t* d_tie_list = nullptr;
    bool host_ties = false;             // the host's libm reproduces the pinned reference energy (probed at opv_create)
    bool tx_trust_libm = false;         // the host's sin / cos behave at the transmit NCOs' flat tops as k_tx_modulate.hip assumes (probed at opv_create)
    bool push_gather = true;            // opv_push_iq_batch moves pinned blocks with one gather kernel (else: one copy per block)
    uint32_t push_gather_blocks = 32;   // workgroups of that kernel (see push_deferred)
    struct TieWork {                    // what the host function gets: stable for the context's life (opv_destroy drains the stream first)
        OpvTieStage* stage = nullptr;   // pinned: header + tie_slots slots
        uint32_t slots = 0;
        std::atomic<uint64_t> decided{0};   // streams the host has decided so far (opv_offset_ties_decided_on_host)
        std::atomic<uint64_t> left{0};      // streams listed beyond what a round's passes stage: the device's decision stood (opv_offset_ties_left_to_device)
    } tie;
    std::vector<int> search_now;        // streams whose offset search can run in the round being enqueued
    // opv_push_iq_batch: the table of the gather kernel / of the batched compaction, in pinned memory (the kernels read it in place)
    void* h_bulk_tab = nullptr;
    size_t bulk_tab_bytes = 0;
    // opv_push_iq_batch_async: moves enqueued on the copy stream that no host wait has covered yet; opv_process orders its kernels
    // behind push_ev on the device, every other push entry point waits on the host first
    bool push_pending = false;
    hipEvent_t push_ev = nullptr;
    uint64_t cap_soft = 0;
    uint32_t cap_frames = 0, cap_events = 0, cap_chunks = 0;
    bool mirror_valid = false;
    // Back-pressure: a stream that paused in the last round (OpvStream.stalled) must be retried by the next
    // opv_process even if the caller pushed nothing new. Unknown (= true) from a launch until the next refresh().
    bool maybe_stalled = false;
    // device transmit chain: NCO phases at symbol starts (data-independent: expanded once per context and run length from
    // the build-time checkpoint table, shared by every stream modulated afterwards) + grow-only scratch
    double* d_tx_ckpt = nullptr;         // checkpoints uploaded so far, 2 doubles each
    size_t tx_ckpt_cap = 0, tx_ckpt_have = 0;
    double* d_tx_phases = nullptr;       // (ph1, ph2) at every symbol start
    size_t tx_phases_cap = 0, tx_phases_have = 0;   // symbols
    uint8_t* d_tx_frames = nullptr;      // [frames][134]
    uint8_t* d_tx_codes = nullptr;       // one code byte per symbol
    uint8_t* d_tx_fpar = nullptr;        // per-frame parity / prefix
    size_t tx_frames_cap = 0;            // frames the three scratch buffers hold
    uint32_t* d_tx_cnt = nullptr;        // ambiguous-sample counter + list
    uint64_t* d_tx_list = nullptr;
    // flat tops (k_tx_modulate.hip): symbols [tx_flat_lo, tx_flat_hi) are decided by a bit per tone made with the host's libm;
    // d_tx_flat holds them for [tx_flat_lo, tx_flat_have)
    uint8_t* d_tx_flat = nullptr;
    uint64_t tx_flat_lo = ~0ull, tx_flat_hi = ~0ull, tx_flat_have = 0, tx_flat_cap = 0;
    const char* last_frontend = "";   // kernel the last opv_process launched for the front-end (opv_frontend_kernel)
    int frontend = 0;  // 0: by stream count, 1: one wave per stream, 4: four streams per wave, -1 / -2: see opv_set_frontend
    bool timing = false;
    bool timing_valid = false;
    hipEvent_t ev[8] = {};

    // Host copies of the record pools (frames, metrics, frame records, events), fetched once per round
    // with four copies for ALL streams instead of three or four small copies per stream and pop: a
    // server popping 256 live streams spent 10 ms per 40 ms round in those copies. Pools above
    // kPoolMirrorMax (big offline captures) are read per stream as before.
    static constexpr size_t kPoolMirrorMax = 64u << 20;
    std::vector<uint8_t> m_frames;
    std::vector<int32_t> m_metrics;
    std::vector<OpvFrameRec> m_frec;
    std::vector<OpvEventRec> m_events;
    unsigned mirror_epoch = 0, pools_epoch = ~0u;

    int refresh() {
        if (mirror_valid) return OPV_OK;
        HIPCHK(hipSetDevice(cfg.device));       // (every pop / state / tap comes through here: a host with contexts on several GPUs)
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipMemcpy(mirror.data(), d_streams, sizeof(OpvStream) * n_streams, hipMemcpyDeviceToHost));
        mirror_valid = true;
        ++mirror_epoch;
        maybe_stalled = false;
        for (const OpvStream& m : mirror) maybe_stalled |= m.stalled != 0;
        return OPV_OK;
    }
    size_t pool_bytes() const {
        const size_t S = (size_t)n_streams;
        return S * ((size_t)cap_frames * (OPV_FB + sizeof(int32_t) + sizeof(OpvFrameRec)) + (size_t)cap_events * sizeof(OpvEventRec));
    }
    // after refresh(): bring the pools over if they are small enough; returns whether host views exist
    int ensure_pools(bool* have) {
        *have = false;
        if (pool_bytes() > kPoolMirrorMax) return OPV_OK;
        if (pools_epoch != mirror_epoch) {
            const size_t S = (size_t)n_streams;
            m_frames.resize((size_t)OPV_FB * cap_frames * S);
            m_metrics.resize((size_t)cap_frames * S);
            m_frec.resize((size_t)cap_frames * S);
            m_events.resize((size_t)cap_events * S);
            HIPCHK(hipMemcpy(m_frames.data(), d_frames, m_frames.size(), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(m_metrics.data(), d_metrics, m_metrics.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(m_frec.data(), d_frec, m_frec.size() * sizeof(OpvFrameRec), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(m_events.data(), d_events, m_events.size() * sizeof(OpvEventRec), hipMemcpyDeviceToHost));
            pools_epoch = mirror_epoch;
        }
        *have = true;
        return OPV_OK;
    }
    // host address of a device pointer into one of the mirrored pools (nullptr if it is not one)
    const void* host_view(const void* d) const {
        auto in = [&](const void* base, size_t bytes) { return (const char*)d >= (const char*)base && (const char*)d < (const char*)base + bytes; };
        if (in(d_frames, m_frames.size())) return m_frames.data() + ((const char*)d - (const char*)d_frames);
        if (in(d_metrics, m_metrics.size() * sizeof(int32_t))) return (const char*)m_metrics.data() + ((const char*)d - (const char*)d_metrics);
        if (in(d_frec, m_frec.size() * sizeof(OpvFrameRec))) return (const char*)m_frec.data() + ((const char*)d - (const char*)d_frec);
        if (in(d_events, m_events.size() * sizeof(OpvEventRec))) return (const char*)m_events.data() + ((const char*)d - (const char*)d_events);
        return nullptr;
    }
};

// k_tx_modulate.hip decides the flat tops of the transmit NCOs by the symbol index: |sin| / |cos| at a quarter-period point
// +/- eps is exactly 1.0 while eps < kFlatExact and below 1.0 from kFlatBelow on. True of glibc's correctly rounded-in-practice
// sin / cos (1 - eps^2/2 rounds to 1.0 while eps^2/2 < 2^-54); a libm with 1 ulp of slack may answer differently, so it is
// asked, once per process: 5 tops x 2 sides x 2 x 256 offsets.
constexpr double kFlatExact = 0.90e-8, kFlatBelow = 1.25e-8;
static bool libm_flat_tops_as_assumed() {
    static const bool ok = [] {
        const double pi = 3.14159265358979323846;
        const struct { bool is_sin; double x0; } tops[] = {{true, pi / 2}, {true, -pi / 2}, {false, 0.0}, {false, pi}, {false, -pi}};
        for (const auto& t : tops)
            for (int k = 0; k <= 256; ++k) {
                const double e = kFlatExact * k / 256.0;                                       // must be exactly +/-1
                const double b = kFlatBelow * std::pow(1.0e-5 / kFlatBelow, k / 256.0);       // must be below: 1.25e-8 .. 1e-5, log-spaced
                for (double sgn : {1.0, -1.0}) {
                    const double ve = std::fabs(t.is_sin ? std::sin(t.x0 + sgn * e) : std::cos(t.x0 + sgn * e));
                    const double vb = std::fabs(t.is_sin ? std::sin(t.x0 + sgn * b) : std::cos(t.x0 + sgn * b));
                    if (ve != 1.0 || !(vb < 1.0)) return false;
                }
            }
        return true;
    }();
    return ok;
}

extern "C" const char* opv_last_error(void) { return g_err.c_str(); }
extern "C" int opv_abi_version(void) { return OPV_ABI_VERSION; }

extern "C" int opv_create(opv_ctx** out, int n_streams, const opv_cfg* cfg) {
    if (!out || !cfg || n_streams <= 0 || cfg->max_samples == 0) return fail(OPV_EINVAL, "opv_create: bad arguments");
    if (cfg->max_samples >= (1ull << 31)) return fail(OPV_EINVAL, "opv_create: max_samples must be < 2^31");
    // the reference's phase wraps are `while (ph > pi) ph -= 2 pi` loops (src/opv-demod.cpp:255-262,488-493): with an infinite -o / -a / -p
    // it spins forever on its CPU; here that would be a wave that never finishes, so such values are refused up front
    if ((cfg->have_init_offset && !std::isfinite(cfg->init_offset_hz)) || !std::isfinite(cfg->afc_alpha) || (cfg->coherent && !std::isfinite(cfg->pll_bw_hz)))
        return fail(OPV_EINVAL, "opv_create: init_offset_hz / afc_alpha / pll_bw_hz must be finite");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(OPV_ENODEV, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(OPV_ENODEV, "opv_cfg.device out of range");
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(OPV_ENODEV, "device is not gfx950 (MI355X); this library carries gfx950 code objects only");

    opv_ctx* c = new (std::nothrow) opv_ctx();
    if (!c) return fail(OPV_ENOMEM, "host allocation failed");
    c->n_streams = n_streams;
    c->cfg = *cfg;
    c->hs.resize(n_streams);
    c->mirror.resize(n_streams);

    const uint64_t M = cfg->max_samples;
    c->cap_soft = 1;  // power-of-two ring: one opv_process worth of symbols (<= M/38) + a frame in flight
    while (c->cap_soft < M / 38 + 4096) c->cap_soft <<= 1;
    c->cap_frames = (uint32_t)(M / (uint64_t)(OPV_FSYMS * 38) + 4);
    c->cap_events = 4 * c->cap_frames + 64;
    c->cap_chunks = (uint32_t)(M / 80000 + 4);

    auto cleanup = [&](int code) { opv_destroy(c); return code; };
#define HIPCHK_C(expr)                                                        \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return cleanup(fail(_e == hipErrorOutOfMemory ? OPV_ENOMEM : OPV_EHIP, #expr, _e)); \
    } while (0)

    HIPCHK_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {   // The copy stream also carries a KERNEL (k_push_gather). The runtime has a handful of hardware queues per priority and
        // deals streams onto them round-robin: in a process with several contexts the kernel stream and the copy stream of one
        // context can land on the same queue, and the moves of round r + 1 then wait for the kernels of round r (bench.py's
        // pcie_inclusive: 109 ms instead of 70). A stream of another PRIORITY comes from another set of queues.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();                       // (no priorities here: an ordinary stream - correct, the overlap is then up to the queue deal)
            c->copy_stream = nullptr;
            HIPCHK_C(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        }
    }
    const size_t S = (size_t)n_streams;
    HIPCHK_C(hipHostMalloc(&c->h_in, sizeof(StreamIn) * S * opv_ctx::kInSlots, hipHostMallocDefault));
    for (auto& e : c->in_ev) HIPCHK_C(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK_C(hipHostMalloc(&c->h_stall, sizeof(int) * opv_ctx::kInSlots, hipHostMallocDefault));
    for (int i = 0; i < opv_ctx::kInSlots; ++i) c->h_stall[i] = 0;
    HIPCHK_C(hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming));
    HIPCHK_C(hipMalloc(&c->d_streams, sizeof(OpvStream) * S));
    HIPCHK_C(hipMalloc(&c->d_in, sizeof(StreamIn) * S));
    HIPCHK_C(hipMalloc(&c->d_soft, sizeof(double) * c->cap_soft * S));
    HIPCHK_C(hipMalloc(&c->d_frec, sizeof(OpvFrameRec) * c->cap_frames * S));
    HIPCHK_C(hipMalloc(&c->d_events, sizeof(OpvEventRec) * c->cap_events * S));
    HIPCHK_C(hipMalloc(&c->d_chunks, sizeof(double) * 5 * c->cap_chunks * S));
    HIPCHK_C(hipMalloc(&c->d_frames, (size_t)OPV_FB * c->cap_frames * S));
    HIPCHK_C(hipMalloc(&c->d_metrics, sizeof(int32_t) * c->cap_frames * S));
    HIPCHK_C(hipMalloc(&c->d_fscale, sizeof(double) * c->cap_frames * S));
    HIPCHK_C(hipMalloc(&c->d_counts, sizeof(int32_t) * S));
    {   // cos / sin(pi i / 80) x u^k / k!, u = i - 19.5: the tone tables of the offset search folded into its Taylor weights
        std::vector<double> w((size_t)OPV_SPS * 2 * OPV_OFFS_TERMS);
        for (int i = 0; i < OPV_SPS; ++i) {
            const double u = (double)i - 19.5, cs = std::cos(M_PI * i / 80.0), sn = std::sin(M_PI * i / 80.0);
            double t = 1.0;                                // u^k / k!
            for (int k = 0; k < OPV_OFFS_TERMS; ++k) {
                w[((size_t)i * 2 + 0) * OPV_OFFS_TERMS + k] = cs * t;
                w[((size_t)i * 2 + 1) * OPV_OFFS_TERMS + k] = sn * t;
                t = t * u / (double)(k + 1);
            }
        }
        HIPCHK_C(hipMalloc(&c->d_offs_wtab, sizeof(double) * w.size()));
        HIPCHK_C(hipMemcpy(c->d_offs_wtab, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
    }
    // The library's four environment switches, all TEST HOOKS (include/opv_demod.h has the table): read here, once per context,
    // and nowhere else - a context never changes its behaviour after it has been created.
    c->host_ties = opv_offset_host_libm_matches_reference() && !std::getenv("OPV_OFFSET_DISTRUST_LIBM");
    c->tx_trust_libm = libm_flat_tops_as_assumed() && !std::getenv("OPV_TX_DISTRUST_LIBM");
    c->push_gather = !std::getenv("OPV_PUSH_NO_GATHER");
    if (const char* e = std::getenv("OPV_PUSH_GATHER_BLOCKS")) { const int v = atoi(e); if (v > 0) c->push_gather_blocks = (uint32_t)v; }
    if (c->host_ties) {
        HIPCHK_C(hipMalloc(&c->d_tie_list, sizeof(uint32_t) * (S + 1)));
        c->tie.slots = (uint32_t)(S < (size_t)OPV_TIE_SLOTS_MAX ? S : (size_t)OPV_TIE_SLOTS_MAX);
        // (82 MB of pinned memory for 512 slots: a host that cannot pin that much keeps working - the device then decides the ties
        // itself, and opv_offset_ties_on_host() says so)
        if (hipHostMalloc((void**)&c->tie.stage, offsetof(OpvTieStage, slot) + sizeof(OpvTieSlot) * c->tie.slots, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->tie.stage = nullptr;
            c->host_ties = false;
        } else {
            c->tie.stage->n = c->tie.stage->listed = c->tie.stage->beyond = 0;
        }
    }
    HIPCHK_C(hipMemsetAsync(c->d_frames, 0, (size_t)OPV_FB * c->cap_frames * S, c->stream));
    HIPCHK_C(hipMemsetAsync(c->d_counts, 0, sizeof(int32_t) * S, c->stream));
    k_fill_i32<<<256, 256, 0, c->stream>>>(c->d_metrics, INT32_MIN, (size_t)c->cap_frames * S);

    for (size_t i = 0; i < S; ++i) {
        OpvStream s;
        std::memset(&s, 0, sizeof s);
        s.soft = c->d_soft + c->cap_soft * i;
        s.cap_soft = c->cap_soft;
        s.frec = c->d_frec + (size_t)c->cap_frames * i;
        s.events = c->d_events + (size_t)c->cap_events * i;
        s.chunk_log = c->d_chunks + (size_t)5 * c->cap_chunks * i;
        s.frames = c->d_frames + (size_t)OPV_FB * c->cap_frames * i;
        s.metrics = c->d_metrics + (size_t)c->cap_frames * i;
        s.fscale = c->d_fscale + (size_t)c->cap_frames * i;
        s.cap_frames = c->cap_frames;
        s.cap_events = c->cap_events;
        s.cap_chunks = c->cap_chunks;
        // demod.set_freq_offset(init_offset) only in streaming mode (ref :1004-1005)
        s.freq_offset = (cfg->streaming && cfg->have_init_offset) ? cfg->init_offset_hz : 0.0;
        s.afc_alpha = cfg->afc_alpha;  // set_afc_bandwidth (ref :1009 / :1172)
        s.est_offset = NAN;
        s.x40c = 1.0;  // X[40] of a (non-existent) previous symbol; only ever multiplies a zero prev
        s.trk_state = OPV_HUNTING;
        c->mirror[i] = s;
    }
    c->initial = c->mirror;
    HIPCHK_C(hipMemcpyAsync(c->d_streams, c->mirror.data(), sizeof(OpvStream) * S, hipMemcpyHostToDevice, c->stream));
    HIPCHK_C(hipStreamSynchronize(c->stream));
    c->mirror_valid = true;
    *out = c;
    return OPV_OK;
#undef HIPCHK_C
}

extern "C" void opv_destroy(opv_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);       // (a host with contexts on several GPUs: free on the context's own device)
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& e : c->in_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_stall) (void)hipHostFree(c->h_stall);
    if (c->tie.stage) (void)hipHostFree(c->tie.stage);
    if (c->h_bulk_tab) (void)hipHostFree(c->h_bulk_tab);
    if (c->push_ev) (void)hipEventDestroy(c->push_ev);
    if (c->done_ev) (void)hipEventDestroy(c->done_ev);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& h : c->hs) {
        if (h.d_iq_owned) (void)hipFree(h.d_iq_owned);
        if (h.d_iq_alt) (void)hipFree(h.d_iq_alt);
    }
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    void* ptrs[] = {c->d_streams, c->d_in, c->d_soft, c->d_frec, c->d_events, c->d_chunks, c->d_frames, c->d_metrics, c->d_fscale, c->d_counts, c->d_tx_phases, c->d_offs_wtab, c->d_tx_ckpt, c->d_tx_frames, c->d_tx_codes, c->d_tx_fpar, c->d_tx_cnt, c->d_tx_list, c->d_tx_flat, c->d_tie_list};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int check_stream(opv_ctx* c, int s) {
    if (!c) return fail(OPV_EINVAL, "null context");
    if (s < 0 || s >= c->n_streams) return fail(OPV_EINVAL, "stream index out of range");
    return OPV_OK;
}

// Long-running streams: the per-stream device buffers are bounded (opv_cfg.max_samples). When a
// push would overflow the IQ buffer, everything the kernels can no longer touch is dropped:
// IQ before the current chunk origin (minus the 16-sample reach of the early gate / tile
// alignment). The soft-symbol log and the frame / event / chunk records are rings on the device
// and need no host action. Indices handed to the caller stay absolute.
static int compact_stream(opv_ctx* c, int s) {
    if (int r = c->refresh()) return r;  // synchronises (once per round: the mirror stays valid for the other streams)
    HostStream& h = c->hs[s];
    OpvStream st = c->mirror[s];
    const uint64_t keep = st.origin >= 16 ? ((st.origin - 16) & ~3ull) : 0;
    if (keep == 0 || !h.d_iq_owned) return OPV_OK;
    const uint64_t len = h.n_avail - keep;  // samples to retain (less than two chunks in steady state)
    if (!h.d_iq_alt) HIPCHK(hipMalloc(&h.d_iq_alt, h.iq_cap * 4 + 16384));
    // tail -> head of the other buffer, in stream order before the next kernels; later pushes write behind it
    if (len) HIPCHK(hipMemcpyAsync(h.d_iq_alt, h.d_iq_owned + 2 * keep, len * 4, hipMemcpyDeviceToDevice, c->stream));
    std::swap(h.d_iq_owned, h.d_iq_alt);
    h.d_iq = h.d_iq_owned;
    h.n_avail -= keep;
    h.last_round_avail = h.last_round_avail > keep ? h.last_round_avail - keep : 0;
    h.dirty = true;
    st.iq = h.d_iq;
    st.origin -= keep;
    st.n_avail = h.n_avail;
    st.iq_base += keep;
    c->mirror[s] = st;  // the mirror is the staging area of this (pageable -> device, staged at once) copy
    HIPCHK(hipMemcpyAsync(c->d_streams + s, &c->mirror[s], sizeof st, hipMemcpyHostToDevice, c->stream));
    return OPV_OK;
}

// grow-only pinned table for the two bulk kernels: [CompactItem x n][PushPair x n]. k_compact (on `stream`) may still read its
// half when the next batch arrives; that half is rewritten only behind a refresh() (which waits for `stream`), the other half
// only behind the wait for the copy stream that ends every batch.
static int bulk_table(opv_ctx* c, size_t n, CompactItem** items, PushPair** pairs) {
    const size_t need = n * (sizeof(CompactItem) + sizeof(PushPair));
    if (c->bulk_tab_bytes < need) {
        HIPCHK(hipStreamSynchronize(c->copy_stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->h_bulk_tab) HIPCHK(hipHostFree(c->h_bulk_tab));
        c->h_bulk_tab = nullptr;
        c->bulk_tab_bytes = 0;
        const size_t cap = need < 65536 ? 65536 : 2 * need;
        HIPCHK(hipHostMalloc(&c->h_bulk_tab, cap, hipHostMallocDefault));
        c->bulk_tab_bytes = cap;
    }
    const size_t per = c->bulk_tab_bytes / (sizeof(CompactItem) + sizeof(PushPair));
    *items = (CompactItem*)c->h_bulk_tab;
    *pairs = (PushPair*)((char*)c->h_bulk_tab + per * sizeof(CompactItem));
    return OPV_OK;
}

// compact_stream for MANY streams at once (a live server's staging buffers fill in the same round): one wait, one launch,
// no per-stream copies. Streams that cannot be compacted (nothing consumed yet) are left to push_enqueue's own check.
static int compact_streams(opv_ctx* c, const std::vector<int>& which) {
    HIPCHK(hipStreamSynchronize(c->copy_stream));         // copies enqueued earlier land first
    if (int r = c->refresh()) return r;
    CompactItem* items = nullptr;
    PushPair* pairs = nullptr;
    if (int r = bulk_table(c, (size_t)c->n_streams, &items, &pairs)) return r;
    uint32_t m = 0, most = 0;
    for (int s : which) {
        HostStream& h = c->hs[s];
        OpvStream& st = c->mirror[s];
        const uint64_t keep = st.origin >= 16 ? ((st.origin - 16) & ~3ull) : 0;
        if (keep == 0 || !h.d_iq_owned) continue;
        const uint64_t len = h.n_avail - keep;
        if (!h.d_iq_alt) HIPCHK(hipMalloc(&h.d_iq_alt, h.iq_cap * 4 + 16384));
        items[m++] = {(const int*)(h.d_iq_owned + 2 * keep), (int*)h.d_iq_alt, (uint32_t)len, (uint32_t)s, keep};
        if ((uint32_t)len > most) most = (uint32_t)len;
        std::swap(h.d_iq_owned, h.d_iq_alt);
        h.d_iq = h.d_iq_owned;
        h.n_avail -= keep;
        h.last_round_avail = h.last_round_avail > keep ? h.last_round_avail - keep : 0;
        h.dirty = true;                                    // (iq and n_avail reach the device with the next round's inputs)
        st.iq = h.d_iq;
        st.origin -= keep;
        st.n_avail -= keep;
        st.iq_base += keep;
    }
    if (m) {
        void* d_items = nullptr;
        HIPCHK(hipHostGetDevicePointer(&d_items, items, 0));
        uint32_t slices = m >= 1024 ? 1 : (1024 + m - 1) / m;
        const uint32_t by_size = most / (256u * 4u);                  // (samples: one 16-byte move per lane of a block at least)
        if (slices > by_size) slices = by_size ? by_size : 1;
        uint32_t grid = m * slices;
        if (grid > 2048) grid = 2048;
        k_compact<<<grid, 256, 0, c->stream>>>(c->d_streams, (const CompactItem*)d_items, m, slices);   // in stream order before the next kernels
        HIPCHK(hipGetLastError());
    }
    return OPV_OK;
}

struct DeferredCopy { void* dst; const void* src; size_t bytes; };
static int push_deferred(opv_ctx* c, const std::vector<DeferredCopy>& copies);

static int push_enqueue(opv_ctx* c, int s, const int16_t* iq, size_t n, std::vector<DeferredCopy>* defer = nullptr) {
    if (int r = check_stream(c, s)) return r;
    HostStream& h = c->hs[s];
    if (h.attached) return fail(OPV_ESTATE, "stream has an attached device capture");
    if (h.eof) return fail(OPV_ESTATE, "push after flush");
    if (n == 0) return OPV_OK;
    if (!iq) return fail(OPV_EINVAL, "null IQ pointer");
    if (h.n_avail + n > c->cfg.max_samples) {
        if (defer && !defer->empty()) {                // blocks of this batch not yet under way (one of them may be this stream's) go first
            if (int r = push_deferred(c, *defer)) return r;
            defer->clear();
        }
        HIPCHK(hipStreamSynchronize(c->copy_stream));  // copies enqueued earlier in this batch land first
        if (int r = compact_stream(c, s)) return r;
        if (h.n_avail + n > c->cfg.max_samples)
            return fail(OPV_ECAPACITY, "opv_cfg.max_samples exceeded (unprocessed samples + this push do not fit; "
                                       "call opv_process between pushes or raise max_samples)");
    }
    if (!h.d_iq_owned) {
        h.iq_cap = c->cfg.max_samples;
        HIPCHK(hipMalloc(&h.d_iq_owned, h.iq_cap * 4 + 16384));  // + slack for whole-tile reads
        h.d_iq = h.d_iq_owned;
    }
    // On the copy stream: the kernels of an opv_process still in flight only read samples below the
    // n_avail they were launched with, so new samples land behind them while they run (H2D over
    // PCIe overlaps compute).
    if (defer) defer->push_back({h.d_iq_owned + 2 * h.n_avail, iq, n * 4});   // (opv_push_iq_batch moves its blocks together)
    else HIPCHK(hipMemcpyAsync(h.d_iq_owned + 2 * h.n_avail, iq, n * 4, hipMemcpyHostToDevice, c->copy_stream));
    h.n_avail += n;
    h.dirty = true;
    return OPV_OK;
}

// The deferred copies of one batch. Blocks in pinned (device-visible) host memory go through ONE gather kernel on the copy
// stream; anything else - pageable memory, a source that is not 4-byte aligned - takes hipMemcpyAsync as before.
static int push_deferred(opv_ctx* c, const std::vector<DeferredCopy>& copies) {
    if (copies.empty()) return OPV_OK;
    CompactItem* items = nullptr;
    PushPair* pairs = nullptr;
    uint32_t m = 0, most = 0;
    const bool try_gather = copies.size() >= 2 && c->push_gather;
    if (try_gather)
        if (int r = bulk_table(c, copies.size() > (size_t)c->n_streams ? copies.size() : (size_t)c->n_streams, &items, &pairs)) return r;
    for (const DeferredCopy& k : copies) {
        bool gathered = false;
        if (try_gather && ((uintptr_t)k.src & 3u) == 0 && k.bytes < (1ull << 32)) {
            hipPointerAttribute_t at;
            // (pinned memory that belongs to ANOTHER device's context is left to hipMemcpyAsync: whether this device may read it in a
            // kernel depends on how it was allocated, and a wrong guess is a page fault, not an error code)
            if (hipPointerGetAttributes(&at, k.src) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer && at.device == c->cfg.device) {
                const uint32_t wide = (((uintptr_t)at.devicePointer | (uintptr_t)k.dst) & 15u) == 0 ? 1u : 0u;
                pairs[m++] = {at.devicePointer, k.dst, (uint32_t)k.bytes, wide};
                if ((uint32_t)k.bytes > most) most = (uint32_t)k.bytes;
                gathered = true;
            } else {
                (void)hipGetLastError();                   // (pageable memory: "invalid value" - not an error of ours)
            }
        }
        if (!gathered) HIPCHK(hipMemcpyAsync(k.dst, k.src, k.bytes, hipMemcpyHostToDevice, c->copy_stream));
    }
    if (m) {
        void* d_pairs = nullptr;
        HIPCHK(hipHostGetDevicePointer(&d_pairs, pairs, 0));
        // A SMALL grid on purpose: PCIe's latency x bandwidth product is ~120 KB, and 32 blocks x 256 lanes x 16 B = 128 KB of
        // reads in flight already run the link at its rate (56 GB/s; 16 blocks: 45). More does not move more - but it fills the
        // memory system's request queues with reads that take microseconds, and the kernels of an opv_process running beside an
        // asynchronous batch then crawl (5120 streams: round kernels 3.5 ms beside 32 blocks, 18 ms beside 128, 27 ms beside 256).
        uint32_t slices = m >= 1024 ? 1 : (1024 + m - 1) / m;         // work items of <= 347 KB / slices: an even load over the blocks ...
        const uint32_t by_size = most / (256u * 16u);                 // ... but never thinner than one 16-byte move per lane of a block
        if (slices > by_size) slices = by_size ? by_size : 1;
        uint32_t grid = m * slices;
        if (grid > c->push_gather_blocks) grid = c->push_gather_blocks;
        k_push_gather<<<grid, 256, 0, c->copy_stream>>>((const PushPair*)d_pairs, m, slices);
        HIPCHK(hipGetLastError());
    }
    return OPV_OK;
}

// moves of an opv_push_iq_batch_async still under way: every other push entry point (and reset / destroy) waits for them first
static int settle_pushes(opv_ctx* c) {
    if (c->push_pending) {
        HIPCHK(hipStreamSynchronize(c->copy_stream));
        c->push_pending = false;
    }
    return OPV_OK;
}

extern "C" int opv_push_wait(opv_ctx* c) {
    if (!c) return fail(OPV_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->cfg.device));
    return settle_pushes(c);
}

extern "C" int opv_push_iq(opv_ctx* c, int s, const int16_t* iq, size_t n) {
    if (!c) return fail(OPV_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->cfg.device));
    if (int r = settle_pushes(c)) return r;
    if (int r = push_enqueue(c, s, iq, n)) return r;
    // The caller keeps ownership of `iq`: the copy must have left the host buffer before we return,
    // which is a wait for THIS copy only, not for the kernels.
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    return OPV_OK;
}

static int push_batch(opv_ctx* c, int count, const int* streams, const int16_t* const* iq, const size_t* n_samples, bool wait);

extern "C" int opv_push_iq_batch(opv_ctx* c, int count, const int* streams, const int16_t* const* iq, const size_t* n_samples) {
    return push_batch(c, count, streams, iq, n_samples, true);
}

extern "C" int opv_push_iq_batch_async(opv_ctx* c, int count, const int* streams, const int16_t* const* iq, const size_t* n_samples) {
    return push_batch(c, count, streams, iq, n_samples, false);
}

static int push_batch(opv_ctx* c, int count, const int* streams, const int16_t* const* iq, const size_t* n_samples, bool wait) {
    if (!c) return fail(OPV_EINVAL, "null context");
    if (count < 0 || (count > 0 && (!streams || !iq || !n_samples))) return fail(OPV_EINVAL, "opv_push_iq_batch: bad arguments");
    HIPCHK(hipSetDevice(c->cfg.device));
    if (int r = settle_pushes(c)) return r;                // (the table of an earlier asynchronous batch may still be read)
    // staging buffers that would overflow are compacted together first (a live server's streams fill in the same round)
    std::vector<int> full;
    for (int i = 0; i < count; ++i) {
        const int s = streams[i];
        if (s < 0 || s >= c->n_streams) continue;          // (reported by push_enqueue below, at its place in the order)
        const HostStream& h = c->hs[s];
        if (!h.attached && !h.eof && n_samples[i] && h.n_avail + n_samples[i] > c->cfg.max_samples) full.push_back(s);
    }
    if (full.size() >= 2)
        if (int r = compact_streams(c, full)) return r;
    std::vector<DeferredCopy> copies;
    copies.reserve((size_t)count);
    int rc = OPV_OK;
    for (int i = 0; i < count && rc == OPV_OK; ++i) rc = push_enqueue(c, streams[i], iq[i], n_samples[i], &copies);
    const int rc2 = push_deferred(c, copies);              // (streams before an error have been pushed)
    if (wait || rc != OPV_OK || rc2 != OPV_OK) {
        HIPCHK(hipStreamSynchronize(c->copy_stream));      // one wait for all copies; the buffers are the caller's again
    } else {
        if (!c->push_ev) HIPCHK(hipEventCreateWithFlags(&c->push_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->push_ev, c->copy_stream)); // opv_process orders its kernels behind this; opv_push_wait waits on the host
        c->push_pending = true;
    }
    return rc != OPV_OK ? rc : rc2;
}

extern "C" int opv_flush(opv_ctx* c, int s) {
    if (int r = check_stream(c, s)) return r;
    c->hs[s].eof = 1;
    c->hs[s].dirty = true;
    return OPV_OK;
}

extern "C" int opv_attach_device_iq(opv_ctx* c, int s, const int16_t* d_iq, size_t n, int eof) {
    if (int r = check_stream(c, s)) return r;
    HostStream& h = c->hs[s];
    if (h.d_iq_owned && h.n_avail && !h.attached) return fail(OPV_ESTATE, "stream already has pushed samples (reset it first)");
    if (!d_iq && n) return fail(OPV_EINVAL, "null device pointer");
    if (((uintptr_t)d_iq & 15u) != 0) return fail(OPV_EINVAL, "device IQ pointer must be 16-byte aligned");
    if (n > c->cfg.max_samples) return fail(OPV_ECAPACITY, "opv_cfg.max_samples exceeded");
    if (h.attached && (d_iq != h.d_iq || n < h.n_avail)) return fail(OPV_ESTATE, "attached capture may only grow");
    h.attached = true;
    h.d_iq = d_iq;
    h.n_avail = n;
    h.eof = eof ? 1 : 0;
    h.dirty = true;
    return OPV_OK;
}

// The host function of one pass of the tie decision (hipLaunchHostFunc, behind k_tie_collect; k_tie_apply follows): the
// contenders of every filled slot are evaluated by the reference's loop on the host's libm (opv_offset_host.cpp). Runs on a
// thread of the runtime in stream order; touches pinned memory and the context's counter only - no HIP call.
static void tie_host_fn(void* p) {
    opv_ctx::TieWork* w = (opv_ctx::TieWork*)p;
    uint32_t n = w->stage->n;
    if (n > w->slots) n = w->slots;
    if (w->stage->beyond) w->left.fetch_add(w->stage->beyond, std::memory_order_relaxed);   // (the round's last pass reports what no pass staged)
    if (n == 0) return;
    opv_offset_decide_slots(w->stage->slot, n);
    w->decided.fetch_add(n, std::memory_order_relaxed);
}

extern "C" int opv_process(opv_ctx* c) {
    if (!c) return fail(OPV_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->cfg.device));
    const int S = c->n_streams;
    const int slot = (int)(c->round_no % opv_ctx::kInSlots);
    if (c->round_no >= (unsigned)opv_ctx::kInSlots) HIPCHK(hipEventSynchronize(c->in_ev[slot]));  // upload of round_no-kInSlots done
    StreamIn* in = c->h_in + (size_t)slot * S;
    uint64_t max_new = 0;
    bool any = false;
    c->search_now.clear();
    for (int i = 0; i < S; ++i) {
        const HostStream& h = c->hs[i];
        in[i] = {h.d_iq, h.n_avail, h.eof, h.dirty ? 1 : 0, h.popped, h.events_popped};
        if (h.dirty) any = true;
        // k_offset_search's own condition (first full chunk in streaming mode, EOF in batch mode), once per stream
        if (!h.search_seen && (c->cfg.streaming ? h.n_avail >= (uint64_t)OPV_CHUNK : h.eof != 0)) c->search_now.push_back(i);
        const uint64_t fresh = h.n_avail - h.last_round_avail;
        if (fresh > max_new) max_new = fresh;
    }
    // nothing new: still run the round while a stream may be waiting behind back-pressure (it resumes by itself once
    // the tracker has consumed soft symbols / the caller has popped frames; the cursors travel with every round).
    // Whether one is: the last round's kernels said so in a pinned word (a caller on the zero-copy path - opv_device_frames +
    // opv_sync, no pops - never refreshes the mirror); while that round is still running the answer is "maybe".
    if (!any && c->maybe_stalled && c->round_no > 0 && hipEventQuery(c->done_ev) == hipSuccess)
        c->maybe_stalled = c->h_stall[(c->round_no - 1) % opv_ctx::kInSlots] != 0;
    if (!any && !c->maybe_stalled) return OPV_OK;
    // frames a stream can release this round: new symbols / 2168 plus what was pending (checked before anything is launched)
    uint64_t fr = max_new / (uint64_t)(OPV_FSYMS * 38) + 4;
    if (fr > c->cap_frames) fr = c->cap_frames;
    if (fr * (uint64_t)S > 0x7FFFFFFFull) return fail(OPV_EINVAL, "opv_process: streams x frames per round exceeds the grid limit");
    // the round will be launched: only now is the host's view of the streams advanced (a refused round leaves it untouched)
    for (int i = 0; i < S; ++i) {
        HostStream& h = c->hs[i];
        h.last_round_avail = h.n_avail;
        h.dirty = false;
    }
    for (int i : c->search_now) c->hs[i].search_seen = true;
    c->mirror_valid = false;
    c->maybe_stalled = true;
    c->h_stall[slot] = 0;          // (a straggler of round_no - kInSlots could still set it: then one idle round too many, never one too few)
    // samples of an asynchronous batch still crossing PCIe: this round's kernels read them, so they queue behind the moves (on the device)
    if (c->push_pending) HIPCHK(hipStreamWaitEvent(c->stream, c->push_ev, 0));
    // pinned -> device, in stream order behind the previous round's kernels; no host wait
    HIPCHK(hipMemcpyAsync(c->d_in, in, sizeof(StreamIn) * S, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->in_ev[slot], c->stream));
    ++c->round_no;
    OpvGlobalCfg g{c->cfg.streaming, c->cfg.have_init_offset, c->cfg.init_offset_hz};
    k_apply_inputs<<<(S + 63) / 64, 64, 0, c->stream>>>(c->d_streams, c->d_in, S);
    const bool tm = c->timing;
    if (tm) HIPCHK(hipEventRecord(c->ev[0], c->stream));
    // searches that can run this round (none under a streaming -o: ref :1031), and whether their ties go to the host's libm
    const size_t n_search = (c->cfg.streaming && c->cfg.have_init_offset) ? 0 : c->search_now.size();
    const bool host_ties = n_search != 0 && c->host_ties;
    if (host_ties) HIPCHK(hipMemsetAsync(c->d_tie_list, 0, sizeof(uint32_t), c->stream));
    k_offset_search<<<S, 256, 0, c->stream>>>(c->d_streams, g, c->d_offs_wtab, host_ties ? c->d_tie_list : nullptr);
    if (tm) HIPCHK(hipEventRecord(c->ev[1], c->stream));
    // near-ties are decided with the host's libm before the front-end takes the estimate, IN STREAM ORDER: per pass of
    // tie.slots listed streams a kernel stages their inputs in pinned memory, a host function decides them
    // (opv_offset_host.cpp), a kernel carries the results back. At most n_search streams can be listed, hence the number of
    // passes - but never more than OPV_TIE_PASSES_MAX (8 x 512 = 4096 listed streams per round): a pass nobody is listed for
    // costs the stream ~30 us (scripts/microbench/hostfunc.hip), a 32 768-stream round would pay for 64 of them, and a
    // last-place tie that is NOT an exact mirror happens to about one stream in ten thousand (exact mirrors - real-valued
    // captures, the one systematic source - come out the same on the device: test_offset_search_ties_decided_like_the_reference
    // runs both paths). Streams listed beyond that keep the device's decision and are counted (opv_offset_ties_left_to_device).
    // The caller never waits.
    if (host_ties) {
        const uint32_t K = c->tie.slots;
        uint32_t passes = (uint32_t)((n_search + K - 1) / K);
        if (passes > OPV_TIE_PASSES_MAX) passes = OPV_TIE_PASSES_MAX;
        for (uint32_t p = 0; p < passes; ++p) {
            k_tie_collect<<<K, 256, 0, c->stream>>>(c->d_streams, c->d_tie_list, p, K, p + 1 == passes ? 1u : 0u, c->tie.stage);
            if (hipLaunchHostFunc(c->stream, tie_host_fn, &c->tie) != hipSuccess) {
                // a runtime that cannot enqueue host functions: the round goes on with the device's decisions (the staging kernel
                // left ties = 0 in every slot, so nothing would be applied anyway), and from now on the context says so
                (void)hipGetLastError();
                c->host_ties = false;
                break;
            }
            k_tie_apply<<<K, 192, 0, c->stream>>>(c->d_streams, c->tie.stage, K, S);
        }
    }
    if (tm) HIPCHK(hipEventRecord(c->ev[2], c->stream));
    // one wave per stream has the shortest per-symbol latency (what counts while SIMDs are idle); four streams per
    // wave issue fewer instructions per symbol and stream, which pays once there are more streams than the one-wave
    // kernel's two rounds of 1024 hold
    const bool x4 = c->frontend == 4 || (c->frontend == 0 && S >= kFrontendX4MinStreams);
    if (c->cfg.coherent && !c->cfg.streaming) {           // -c, batch only (ref :1144-1161)
        const double wn = c->cfg.pll_bw_hz * 2.0 * M_PI, zeta = 0.707, fsym = 2168000.0 / 40.0;   // set_pll_bandwidth (ref :551-558)
        c->last_frontend = "k_coherent_frontend";
        k_coherent_frontend<<<S, 64, 0, c->stream>>>(c->d_streams, 2.0 * zeta * wn / fsym, wn * wn / (fsym * fsym));
    }
    else if (c->frontend == 16 || (c->frontend == 0 && S > kFrontendX16MinStreams)) {   // sixteen streams per wave (k_frontend_x16.hip): four waves (64 streams) per workgroup
        if (S > kFrontendX16Wg8MinStreams) { c->last_frontend = "k_msk_frontend_x16_wg8"; k_msk_frontend_x16_wg8<<<(S + 127) / 128, 512, 0, c->stream>>>(c->d_streams, g, S); }
        else if (S > 16) { c->last_frontend = "k_msk_frontend_x16_wg4"; k_msk_frontend_x16_wg4<<<(S + 63) / 64, 256, 0, c->stream>>>(c->d_streams, g, S); }
        else { c->last_frontend = "k_msk_frontend_x16"; k_msk_frontend_x16<<<1, 64, 0, c->stream>>>(c->d_streams, g, S); }
    }
    else if (x4) {                                        // four waves (16 streams) per workgroup, whatever the stream count (automatic: 2049..8192)
        c->last_frontend = "k_msk_frontend_x4_wg4";
        k_msk_frontend_x4_wg4<<<(S + 15) / 16, 256, 0, c->stream>>>(c->d_streams, g, S);
    }
    // one wave per stream, row-broadcast reduction (k_frontend.hip: symbol_r). Its 272-278 registers allow one wave per SIMD;
    // four waves per workgroup (one per SIMD of a CU by construction) as soon as single-wave workgroups could double up
    else if (S > kFrontendWg4MinStreams) {
        c->last_frontend = "k_msk_frontend_rb_wg4";
        k_msk_frontend_rb_wg4<<<(S + 3) / 4, 256, 0, c->stream>>>(c->d_streams, g, S);
    } else {
        c->last_frontend = "k_msk_frontend_rb";
        k_msk_frontend_rb<<<S, 64, 0, c->stream>>>(c->d_streams, g, S);
    }
    if (tm) { HIPCHK(hipEventRecord(c->ev[3], c->stream)); HIPCHK(hipEventRecord(c->ev[4], c->stream)); }
    k_sync_track<<<S, 64, 0, c->stream>>>(c->d_streams);
    if (tm) { HIPCHK(hipEventRecord(c->ev[5], c->stream)); HIPCHK(hipEventRecord(c->ev[6], c->stream)); }
    // the quantiser's scale (2144 dependent additions per frame) with one frame per lane, then one wave per two frames
    // (one frame per lane - the HBM-rate shape - for bulk rounds; one wave per frame while the round is so small that a wave of 64
    // frames would be the whole launch: a live round of 64 frames 46 -> 12 us)
    if (fr * (uint64_t)S <= kScaleWaveMaxFrames) k_frame_scale_wave<<<(unsigned)(fr * (uint64_t)S), 64, 0, c->stream>>>(c->d_streams, (uint32_t)fr);
    else k_frame_scale<<<(unsigned)((fr * (uint64_t)S + 63) / 64), 64, 0, c->stream>>>(c->d_streams, (uint32_t)fr, (uint32_t)S);
    k_frame_decode<<<(unsigned)(((fr + 1) / 2) * (uint64_t)S), 64, 0, c->stream>>>(c->d_streams, (uint32_t)((fr + 1) / 2));   // two frames per wave
    if (tm) { HIPCHK(hipEventRecord(c->ev[7], c->stream)); c->timing_valid = true; }
    k_collect_counts<<<(S + 63) / 64, 64, 0, c->stream>>>(c->d_streams, c->d_counts, S, c->h_stall + slot);
    HIPCHK(hipEventRecord(c->done_ev, c->stream));
    HIPCHK(hipGetLastError());
    return OPV_OK;
}

extern "C" int opv_set_frontend(opv_ctx* c, int streams_per_wave) {
    if (!c) return fail(OPV_EINVAL, "null context");
    if (streams_per_wave != 0 && streams_per_wave != 1 && streams_per_wave != 4 && streams_per_wave != 16)
        return fail(OPV_EINVAL, "opv_set_frontend: 0 (automatic), 1, 4 or 16 streams per wave");
    c->frontend = streams_per_wave;
    return OPV_OK;
}

extern "C" const char* opv_frontend_kernel(opv_ctx* c) { return c ? c->last_frontend : ""; }

extern "C" int opv_sync(opv_ctx* c) {
    if (!c) return fail(OPV_EINVAL, "null context");
    HIPCHK(hipStreamSynchronize(c->stream));
    return OPV_OK;
}

extern "C" int opv_reset_stream(opv_ctx* c, int s) {
    if (!c) return fail(OPV_EINVAL, "null context");
    if (s != -1) { if (int r = check_stream(c, s)) return r; }
    HIPCHK(hipSetDevice(c->cfg.device));
    if (int r = settle_pushes(c)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    const int lo = (s == -1) ? 0 : s, hi = (s == -1) ? c->n_streams : s + 1;
    for (int i = lo; i < hi; ++i) {
        HostStream& h = c->hs[i];
        int16_t* keep = h.d_iq_owned;
        int16_t* keep_alt = h.d_iq_alt;
        const size_t cap = h.iq_cap;
        h = HostStream();
        h.d_iq_owned = keep;
        h.d_iq_alt = keep_alt;
        h.iq_cap = cap;
        h.d_iq = keep;
    }
    HIPCHK(hipMemcpyAsync(c->d_streams + lo, &c->initial[lo], sizeof(OpvStream) * (hi - lo), hipMemcpyHostToDevice, c->stream));
    k_fill_i32<<<256, 256, 0, c->stream>>>(c->d_metrics + (size_t)c->cap_frames * lo, INT32_MIN,
                                          (size_t)c->cap_frames * (hi - lo));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->mirror_valid = false;
    return OPV_OK;
}

extern "C" int opv_enable_timing(opv_ctx* c, int enable) {
    if (!c) return fail(OPV_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->cfg.device));
    if (enable)
        for (auto& e : c->ev)
            if (!e) HIPCHK(hipEventCreate(&e));
    c->timing = enable != 0;
    c->timing_valid = false;
    return OPV_OK;
}

extern "C" int opv_kernel_times(opv_ctx* c, float ms[4]) {
    if (!c || !ms) return fail(OPV_EINVAL, "null argument");
    if (!c->timing_valid) return fail(OPV_ESTATE, "no timed opv_process yet (opv_enable_timing first)");
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 4; ++k) HIPCHK(hipEventElapsedTime(&ms[k], c->ev[2 * k], c->ev[2 * k + 1]));
    return OPV_OK;
}

// copies records [first, first+n) of a device ring with `cap` slots of `elem` bytes into dst (host)
static int ring_to_host(opv_ctx* c, void* dst, const void* d_ring, uint32_t first, uint32_t n, uint32_t cap, size_t elem) {
    const uint32_t p0 = first % cap;
    const uint32_t run = n < cap - p0 ? n : cap - p0;
    bool have = false;
    if (int r = c->ensure_pools(&have)) return r;
    if (have) {
        if (const void* h_ring = c->host_view(d_ring)) {
            std::memcpy(dst, (const char*)h_ring + (size_t)p0 * elem, (size_t)run * elem);
            if (run < n) std::memcpy((char*)dst + (size_t)run * elem, h_ring, (size_t)(n - run) * elem);
            return OPV_OK;
        }
    }
    HIPCHK(hipMemcpy(dst, (const char*)d_ring + (size_t)p0 * elem, (size_t)run * elem, hipMemcpyDeviceToHost));
    if (run < n) HIPCHK(hipMemcpy((char*)dst + (size_t)run * elem, d_ring, (size_t)(n - run) * elem, hipMemcpyDeviceToHost));
    return OPV_OK;
}

extern "C" long opv_pop_frames(opv_ctx* c, int s, uint8_t* out, size_t cap, opv_frame_meta* meta) {
    if (int r = check_stream(c, s)) return r;
    if (int r = c->refresh()) return r;
    const OpvStream& st = c->mirror[s];
    if (st.overflow == 2) return fail(OPV_EHIP, "front-end: the two wavefronts of a stream lost each other (hand-over timed out)");
    if (st.overflow) return fail(OPV_EINVAL, "opv_cfg.max_samples too large for the soft-symbol ring (internal limit 2^28 symbols)");
    HostStream& h = c->hs[s];
    const uint32_t nf = st.n_frames;
    if (h.popped >= nf || cap == 0) return 0;
    // candidates: at most `cap` decodable frames can be returned, dropped ones are skipped, so look at
    // the unread records in slices of up to cap and stop when the caller's buffer is full
    size_t w = 0;
    uint32_t f = h.popped;
    std::vector<int32_t> met;
    std::vector<OpvFrameRec> rec;
    std::vector<uint8_t> fr;
    while (f < nf && w < cap) {
        uint32_t n = nf - f;
        if (n > cap - w) n = (uint32_t)(cap - w);
        if (n > st.cap_frames) n = st.cap_frames;
        met.resize(n);
        rec.resize(n);
        fr.resize((size_t)n * OPV_FB);
        if (int r = ring_to_host(c, met.data(), st.metrics, f, n, st.cap_frames, sizeof(int32_t))) return r;
        if (int r = ring_to_host(c, rec.data(), st.frec, f, n, st.cap_frames, sizeof(OpvFrameRec))) return r;
        if (int r = ring_to_host(c, fr.data(), st.frames, f, n, st.cap_frames, OPV_FB)) return r;
        bool stop = false;
        for (uint32_t k = 0; k < n; ++k, ++f) {
            if (met[k] == INT32_MIN) { stop = true; break; }  // released but not decoded yet (cannot happen after opv_process)
            if (met[k] < 0) continue;                         // dropped silent frame (ref :859, :1052)
            if (out) std::memcpy(out + w * OPV_FB, fr.data() + (size_t)k * OPV_FB, OPV_FB);
            if (meta) {
                meta[w].viterbi_metric = met[k];
                meta[w].sync_ok = rec[k].sync_ok;
                meta[w].sync_quality = rec[k].quality;
                meta[w].release_symbol = rec[k].release_sym;
                meta[w].payload_symbol = rec[k].payload_sym;
            }
            ++h.decoded;
            if (met[k] == 0) ++h.perfect;
            ++w;
        }
        if (stop) break;
    }
    h.popped = f;
    return (long)w;
}

extern "C" long opv_pop_events(opv_ctx* c, int s, opv_event* out, size_t cap) {
    if (int r = check_stream(c, s)) return r;
    if (int r = c->refresh()) return r;
    const OpvStream& st = c->mirror[s];
    HostStream& h = c->hs[s];
    const uint32_t ne = st.n_events;
    if (ne - h.events_popped > st.cap_events) {  // lossy ring: the oldest unread entries have been overwritten
        h.events_dropped += (ne - h.events_popped) - st.cap_events;
        h.events_popped = ne - st.cap_events;
    }
    if (h.events_popped >= ne || cap == 0 || !out) return 0;
    uint32_t n = ne - h.events_popped;
    if (n > cap) n = (uint32_t)cap;
    static_assert(sizeof(opv_event) == sizeof(OpvEventRec), "event layouts must match");
    if (int r = ring_to_host(c, out, st.events, h.events_popped, n, st.cap_events, sizeof(OpvEventRec))) return r;
    h.events_popped += n;
    return (long)n;
}

extern "C" int opv_get_state(opv_ctx* c, int s, opv_stream_state* out) {
    if (int r = check_stream(c, s)) return r;
    if (!out) return fail(OPV_EINVAL, "null out");
    if (int r = c->refresh()) return r;
    const OpvStream& st = c->mirror[s];
    std::memset(out, 0, sizeof *out);
    out->freq_offset_hz = st.freq_offset;
    out->timing_freq = st.timing_freq;
    out->est_offset_hz = st.est_offset;
    out->mu = st.mu;
    out->total_symbols = st.n_soft;
    out->total_samples = st.total_samples;
    out->chunk_origin = st.origin + st.iq_base;  // absolute sample index
    out->sync_state = st.trk_state;
    out->frames_released = (int32_t)st.n_frames;
    out->n_chunks = (int32_t)st.n_chunks;
    out->flushed = st.tail_done;
    const HostStream& h = c->hs[s];
    out->frames_decoded = h.decoded;
    out->frames_perfect = h.perfect;
    out->events_dropped = h.events_dropped;
    if (st.n_events - h.events_popped > st.cap_events) out->events_dropped += (st.n_events - h.events_popped) - st.cap_events;
    out->edge_ties = st.edge_ties;
    out->offset_ties = (int32_t)st.est_ties;
    out->stalled = st.stalled;
    if (st.n_frames > h.popped) {  // released but not popped yet
        uint32_t n = st.n_frames - h.popped;
        if (n > st.cap_frames) n = st.cap_frames;
        std::vector<int32_t> met(n);
        if (int r = ring_to_host(c, met.data(), st.metrics, h.popped, n, st.cap_frames, sizeof(int32_t))) return r;
        for (int32_t m : met)
            if (m >= 0) { out->frames_decoded++; if (m == 0) out->frames_perfect++; }
    }
    return st.overflow ? fail(OPV_EINVAL, "opv_cfg.max_samples too large for the soft-symbol ring") : OPV_OK;
}

extern "C" int opv_device_frames(opv_ctx* c, const uint8_t** d_frames, const int32_t** d_metrics,
                                 const int32_t** d_counts, size_t* frame_capacity) {
    if (!c) return fail(OPV_EINVAL, "null context");
    if (d_frames) *d_frames = c->d_frames;
    if (d_metrics) *d_metrics = c->d_metrics;
    if (d_counts) *d_counts = c->d_counts;
    if (frame_capacity) *frame_capacity = c->cap_frames;
    return OPV_OK;
}

extern "C" void* opv_hip_stream(opv_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" long opv_tap_soft(opv_ctx* c, int s, uint64_t first, double* out, size_t cap) {
    if (int r = check_stream(c, s)) return r;
    if (int r = c->refresh()) return r;
    const OpvStream& st = c->mirror[s];
    if (st.n_soft > st.cap_soft && first < st.n_soft - st.cap_soft)
        return fail(OPV_EINVAL, "soft symbols that old have been overwritten (the log is a ring)");
    if (first >= st.n_soft || cap == 0 || !out) return 0;
    uint64_t n = st.n_soft - first;
    if (n > cap) n = cap;
    const uint64_t p0 = first & (st.cap_soft - 1);
    const uint64_t run = n < st.cap_soft - p0 ? n : st.cap_soft - p0;
    HIPCHK(hipMemcpy(out, st.soft + p0, sizeof(double) * run, hipMemcpyDeviceToHost));
    if (run < n) HIPCHK(hipMemcpy(out + run, st.soft, sizeof(double) * (n - run), hipMemcpyDeviceToHost));
    return (long)n;
}

extern "C" long opv_tap_chunks(opv_ctx* c, int s, uint32_t first, double* out5, size_t cap) {
    if (int r = check_stream(c, s)) return r;
    if (int r = c->refresh()) return r;
    const OpvStream& st = c->mirror[s];
    if (first >= st.n_chunks || cap == 0 || !out5) return 0;
    if (st.n_chunks - first > st.cap_chunks) return fail(OPV_EINVAL, "chunk log entries already overwritten (ring)");
    uint32_t n = st.n_chunks - first;
    if (n > cap) n = (uint32_t)cap;
    for (uint32_t k = 0; k < n; ++k)
        HIPCHK(hipMemcpy(out5 + 5 * (size_t)k, st.chunk_log + 5 * (size_t)((first + k) % st.cap_chunks), sizeof(double) * 5,
                         hipMemcpyDeviceToHost));
    return (long)n;
}

extern "C" int opv_tap_offset_energies(opv_ctx* c, int s, double* out134) {
    if (int r = check_stream(c, s)) return r;
    if (!out134) return fail(OPV_EINVAL, "null out");
    if (int r = c->refresh()) return r;
    std::memcpy(out134, c->mirror[s].energies, sizeof(double) * 134);
    return OPV_OK;
}

extern "C" int opv_offset_ties_on_host(opv_ctx* c) { return c && c->host_ties ? 1 : 0; }
extern "C" uint64_t opv_offset_ties_decided_on_host(opv_ctx* c) { return c ? c->tie.decided.load(std::memory_order_relaxed) : 0; }
extern "C" uint64_t opv_offset_ties_left_to_device(opv_ctx* c) { return c ? c->tie.left.load(std::memory_order_relaxed) : 0; }

extern "C" int opv_tap_wave_info(opv_ctx* c, int s, uint64_t out[4]) {
    if (int r = check_stream(c, s)) return r;
    if (!out) return fail(OPV_EINVAL, "null out");
    if (int r = c->refresh()) return r;
    const OpvStream& st = c->mirror[s];
    out[0] = st.dbg_hw_id; out[1] = st.dbg_xcc_id; out[2] = st.dbg_cycles; out[3] = st.dbg_ticks;
    return OPV_OK;
}

extern "C" int opv_tap_occupancy(opv_ctx* c, int out[6]) {
    if (!c || !out) return fail(OPV_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->cfg.device));
    const struct { const void* k; int threads; } ks[6] = {
        {(const void*)k_msk_frontend_rb, 64}, {(const void*)k_msk_frontend_rb_wg4, 256}, {(const void*)k_msk_frontend_x16_wg4, 256},
        {(const void*)k_msk_frontend_x4_wg4, 256}, {(const void*)k_frame_decode, 64}, {(const void*)k_frame_scale, 64}};
    for (int i = 0; i < 6; ++i) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[i], ks[i].k, ks[i].threads, 0));
    return OPV_OK;
}

extern "C" int opv_decode_payloads(opv_ctx* c, const double* soft, size_t n, uint8_t* out, int32_t* metrics,
                                   int8_t* q, int8_t* deint, uint8_t* bits) {
    if (!c || !soft || !out || !metrics) return fail(OPV_EINVAL, "null argument");
    if (n == 0) return OPV_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    double* d_soft = nullptr;
    uint8_t* d_out = nullptr;
    int32_t* d_met = nullptr;
    int8_t *d_q = nullptr, *d_d = nullptr;
    uint8_t* d_b = nullptr;
    double* d_scale = nullptr;
    int rc = OPV_OK;
    auto chk = [&](hipError_t e, const char* w) { if (e != hipSuccess && rc == OPV_OK) rc = fail(OPV_EHIP, w, e); };
    chk(hipMalloc(&d_soft, sizeof(double) * OPV_CODED * n), "hipMalloc soft");
    chk(hipMalloc(&d_out, (size_t)OPV_FB * n), "hipMalloc out");
    chk(hipMalloc(&d_met, sizeof(int32_t) * n), "hipMalloc metrics");
    chk(hipMalloc(&d_scale, sizeof(double) * n), "hipMalloc scales");
    if (q) chk(hipMalloc(&d_q, (size_t)OPV_CODED * n), "hipMalloc q");
    if (deint) chk(hipMalloc(&d_d, (size_t)OPV_CODED * n), "hipMalloc deint");
    if (bits) chk(hipMalloc(&d_b, (size_t)OPV_FBITS * n), "hipMalloc bits");
    if (rc == OPV_OK) {
        chk(hipMemcpyAsync(d_soft, soft, sizeof(double) * OPV_CODED * n, hipMemcpyHostToDevice, c->stream), "H2D soft");
        chk(hipMemsetAsync(d_out, 0, (size_t)OPV_FB * n, c->stream), "memset");
        k_payload_scale<<<(unsigned)((n + 63) / 64), 64, 0, c->stream>>>(d_soft, (uint32_t)n, d_scale);
        k_decode_payloads<<<(unsigned)((n + 1) / 2), 64, 0, c->stream>>>(d_soft, (uint32_t)n, d_scale, d_out, d_met, d_q, d_d, d_b);   // two payloads per wave
        chk(hipGetLastError(), "k_decode_payloads launch");
        chk(hipMemcpyAsync(out, d_out, (size_t)OPV_FB * n, hipMemcpyDeviceToHost, c->stream), "D2H out");
        chk(hipMemcpyAsync(metrics, d_met, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream), "D2H metrics");
        if (q) chk(hipMemcpyAsync(q, d_q, (size_t)OPV_CODED * n, hipMemcpyDeviceToHost, c->stream), "D2H q");
        if (deint) chk(hipMemcpyAsync(deint, d_d, (size_t)OPV_CODED * n, hipMemcpyDeviceToHost, c->stream), "D2H deint");
        if (bits) chk(hipMemcpyAsync(bits, d_b, (size_t)OPV_FBITS * n, hipMemcpyDeviceToHost, c->stream), "D2H bits");
        chk(hipStreamSynchronize(c->stream), "sync");
    }
    void* ptrs[] = {d_soft, d_out, d_met, d_q, d_d, d_b, d_scale};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    return rc;
}

extern "C" int opv_channel_device(opv_ctx* c, const int16_t* d_in, int16_t* d_out, size_t n, double gain,
                                  double f0_hz, double sigma, uint64_t seed) {
    if (!c || !d_in || !d_out) return fail(OPV_EINVAL, "null argument");
    if ((n & 3u) || (((uintptr_t)d_in | (uintptr_t)d_out) & 15u))
        return fail(OPV_EINVAL, "opv_channel_device: n must be a multiple of 4 and pointers 16-byte aligned");
    HIPCHK(hipSetDevice(c->cfg.device));
    const uint64_t quads = n / 4;
    unsigned blocks = (unsigned)((quads + 255) / 256);
    if (blocks > 256u * 8u) blocks = 256u * 8u;  // 8 blocks per CU, grid-stride the rest
    if (blocks == 0) return OPV_OK;
    k_channel<<<blocks, 256, 0, c->stream>>>((const int4*)d_in, (int4*)d_out, quads, gain, f0_hz / 2168000.0, sigma, seed);
    HIPCHK(hipGetLastError());
    return OPV_OK;
}

extern "C" long opv_resample_device(opv_ctx* c, const int16_t* d_in, size_t n_in, int16_t* d_out, size_t out_cap,
                                    double clock_ppm) {
    if (!c || !d_in || !d_out) return fail(OPV_EINVAL, "null argument");
    if (n_in < 2) return fail(OPV_EINVAL, "opv_resample_device: at least two input samples");
    if (!(clock_ppm > -5e5 && clock_ppm < 5e5)) return fail(OPV_EINVAL, "opv_resample_device: |clock_ppm| must be below 5e5");
    HIPCHK(hipSetDevice(c->cfg.device));
    const double rate = 1.0 + clock_ppm * 1e-6;
    const uint64_t n_out = (uint64_t)((double)n_in / rate);
    if (n_out > out_cap) return fail(OPV_ECAPACITY, "opv_resample_device: output buffer too small");
    if (n_out == 0) return 0;
    unsigned blocks = (unsigned)((n_out + 255) / 256);
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    k_resample_clock<<<blocks, 256, 0, c->stream>>>((const int*)d_in, (uint64_t)n_in, (int*)d_out, n_out, rate);
    HIPCHK(hipGetLastError());
    return (long)n_out;
}

extern "C" long opv_tx_modulate_device(opv_ctx* c, const uint8_t* frames, size_t n_frames, int16_t* d_iq_out) {
    if (!c || !d_iq_out || (!frames && n_frames)) return fail(OPV_EINVAL, "null argument");
    if (((uintptr_t)d_iq_out & 15u) != 0) return fail(OPV_EINVAL, "device IQ pointer must be 16-byte aligned");
    if (n_frames > 0x7FFFFFFFull / OPV_FSYMS) return fail(OPV_EINVAL, "opv_tx_modulate_device: too many frames for one call");
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t nsym = n_frames * OPV_FSYMS, nsym_total = nsym + 100;  // + 100 silent symbols (opv-mod.cpp:528-529)
    constexpr uint32_t kAmbCap = 4096;
    // ---- symbol-start phases: from the checkpoint table, expanded on the device, once per context and run length
    if (c->tx_phases_have < nsym) {
        const size_t need_ck = (nsym + OPV_TX_CKPT_SYMS - 1) / OPV_TX_CKPT_SYMS;
        if (c->tx_ckpt_have < need_ck) {
            if (c->tx_ckpt_cap < need_ck) {
                if (c->d_tx_ckpt) HIPCHK(hipFree(c->d_tx_ckpt));
                c->d_tx_ckpt = nullptr;
                c->tx_ckpt_cap = c->tx_ckpt_have = 0;
                HIPCHK(hipMalloc(&c->d_tx_ckpt, sizeof(double) * 2 * need_ck));
                c->tx_ckpt_cap = need_ck;
            }
            std::vector<double> ck(2 * (need_ck - c->tx_ckpt_have));
            opv_tx_checkpoint_range(c->tx_ckpt_have, need_ck - c->tx_ckpt_have, ck.data());
            HIPCHK(hipMemcpy(c->d_tx_ckpt + 2 * c->tx_ckpt_have, ck.data(), sizeof(double) * ck.size(), hipMemcpyHostToDevice));
            c->tx_ckpt_have = need_ck;
        }
        size_t first_ck = c->tx_phases_have / OPV_TX_CKPT_SYMS;            // whole intervals already expanded stay
        if (c->tx_phases_cap < nsym) {
            HIPCHK(hipStreamSynchronize(c->stream));                        // (an earlier modulation may still read the old table)
            if (c->d_tx_phases) HIPCHK(hipFree(c->d_tx_phases));
            c->d_tx_phases = nullptr;
            c->tx_phases_cap = c->tx_phases_have = 0;
            HIPCHK(hipMalloc(&c->d_tx_phases, sizeof(double) * 2 * nsym));
            c->tx_phases_cap = nsym;
            first_ck = 0;
        }
        const size_t n_new = need_ck - first_ck;
        k_tx_expand_phases<<<(unsigned)((n_new + 63) / 64), 64, 0, c->stream>>>((const double2*)c->d_tx_ckpt, (uint32_t)need_ck, first_ck,
                                                                                 nsym, (double2*)c->d_tx_phases);
        HIPCHK(hipGetLastError());
        c->tx_phases_have = nsym;
    }
    // ---- flat tops: the zone limits from the checkpoint sequence (the drift is monotone), the bits in between from libm.
    // Both zone limits are statements about THIS process's libm (glibc: exactly 1.0 while eps < 1.05e-8); they are probed once
    // per process (libm_flat_tops_as_assumed), and the partition the binary search relies on is spot-checked. If either fails -
    // another libm, a drift that is not monotone - every symbol's bits come from libm (flat_lo = 0, no upper zone): slower to
    // set up (two sincos per symbol of the run on the host), identical to `opv-mod` on this machine by construction.
    if (c->tx_flat_hi == ~0ull) {                            // (both limits found: final - the sequence is data-independent)
        const size_t n_ck = (nsym + OPV_TX_CKPT_SYMS - 1) / OPV_TX_CKPT_SYMS;
        auto drift = [](size_t j) {                          // distance of checkpoint j's phases from a multiple of pi/2
            double p[2];
            opv_tx_checkpoint_range(j, 1, p);
            const double q = 1.57079632679489661923;
            const double d1 = std::fabs(p[0] - std::nearbyint(p[0] / q) * q), d2 = std::fabs(p[1] - std::nearbyint(p[1] / q) * q);
            return d1 > d2 ? d1 : d2;
        };
        auto first_at = [&](double thr) -> size_t {         // smallest checkpoint index with drift >= thr, n_ck if none
            size_t lo = 0, hi = n_ck;
            while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if (drift(mid) >= thr) hi = mid; else lo = mid + 1; }
            return lo;
        };
        const size_t j_lo = n_ck ? first_at(kFlatExact) : 0, j_hi = n_ck ? first_at(kFlatBelow) : 0;
        bool ok = c->tx_trust_libm;
        for (size_t t = 0; ok && n_ck && t < 16; ++t) {     // the partition the search assumed, at 16 places across the run
            const size_t j = t * (n_ck - 1) / 15;
            const double d = drift(j);
            ok = (d >= kFlatExact) == (j >= j_lo) && (d >= kFlatBelow) == (j >= j_hi);
        }
        if (ok) {
            c->tx_flat_lo = j_lo >= n_ck ? ~0ull : (uint64_t)(j_lo ? j_lo - 1 : 0) * OPV_TX_CKPT_SYMS;
            c->tx_flat_hi = j_hi >= n_ck ? ~0ull : (uint64_t)(j_hi + 1) * OPV_TX_CKPT_SYMS;
        } else {
            c->tx_flat_lo = 0;
            c->tx_flat_hi = ~0ull - 1;                       // (not the "unknown" value: final)
        }
    }
    if (c->tx_flat_lo != ~0ull && nsym > c->tx_flat_lo) {
        const uint64_t want = c->tx_flat_hi < (uint64_t)nsym ? c->tx_flat_hi : (uint64_t)nsym;   // bits for [flat_lo, want)
        if (c->tx_flat_have < want) {
            const size_t cnt = (size_t)(want - c->tx_flat_lo);
            if (c->tx_flat_cap < cnt) {
                HIPCHK(hipStreamSynchronize(c->stream));
                if (c->d_tx_flat) HIPCHK(hipFree(c->d_tx_flat));
                c->d_tx_flat = nullptr;
                c->tx_flat_cap = 0;
                c->tx_flat_have = 0;
                HIPCHK(hipMalloc(&c->d_tx_flat, cnt));
                c->tx_flat_cap = cnt;
            }
            std::vector<double> ph(2 * cnt);
            HIPCHK(hipStreamSynchronize(c->stream));                        // the expansion above
            HIPCHK(hipMemcpy(ph.data(), c->d_tx_phases + 2 * c->tx_flat_lo, sizeof(double) * 2 * cnt, hipMemcpyDeviceToHost));
            std::vector<uint8_t> bits(cnt);
            auto work = [&](size_t a, size_t b) {
                for (size_t k = a; k < b; ++k) {
                    const double p1 = ph[2 * k], p2 = ph[2 * k + 1];
                    const bool e1 = std::fabs(std::sin(p1)) == 1.0 || std::fabs(std::cos(p1)) == 1.0;
                    const bool e2 = std::fabs(std::sin(p2)) == 1.0 || std::fabs(std::cos(p2)) == 1.0;
                    bits[k] = (uint8_t)((e1 ? 1 : 0) | (e2 ? 2 : 0));
                }
            };
            unsigned nt = std::thread::hardware_concurrency();
            if (nt == 0) nt = 1;
            if (nt > 16) nt = 16;
            if (cnt < 65536) nt = 1;
            std::vector<std::thread> pool;
            const size_t per = (cnt + nt - 1) / nt;
            for (unsigned t = 1; t < nt; ++t) { const size_t a = t * per, b = a + per < cnt ? a + per : cnt; if (a < b) pool.emplace_back(work, a, b); }
            work(0, per < cnt ? per : cnt);
            for (auto& th : pool) th.join();
            HIPCHK(hipMemcpy(c->d_tx_flat, bits.data(), cnt, hipMemcpyHostToDevice));
            c->tx_flat_have = want;
        }
    }
    // ---- scratch (grow-only)
    if (c->tx_frames_cap < n_frames || !c->d_tx_cnt) {
        HIPCHK(hipStreamSynchronize(c->stream));
        void* old[] = {c->d_tx_frames, c->d_tx_codes, c->d_tx_fpar};
        for (void* p : old) if (p) HIPCHK(hipFree(p));
        c->d_tx_frames = c->d_tx_codes = c->d_tx_fpar = nullptr;
        c->tx_frames_cap = 0;
        const size_t cap = n_frames ? n_frames : 1;
        HIPCHK(hipMalloc(&c->d_tx_frames, cap * OPV_FB));
        HIPCHK(hipMalloc(&c->d_tx_codes, cap * OPV_FSYMS));
        HIPCHK(hipMalloc(&c->d_tx_fpar, cap));
        c->tx_frames_cap = cap;
        if (!c->d_tx_cnt) HIPCHK(hipMalloc(&c->d_tx_cnt, sizeof(uint32_t)));
        if (!c->d_tx_list) HIPCHK(hipMalloc(&c->d_tx_list, sizeof(uint64_t) * kAmbCap));
    }
    // ---- frames -> code bytes + frame prefixes -> samples, all on the context's stream
    if (n_frames) HIPCHK(hipMemcpyAsync(c->d_tx_frames, frames, n_frames * OPV_FB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_tx_cnt, 0, sizeof(uint32_t), c->stream));
    if (n_frames) {
        k_tx_encode<<<(unsigned)n_frames, 256, 0, c->stream>>>(c->d_tx_frames, (uint32_t)n_frames, c->d_tx_codes, c->d_tx_fpar);
        k_tx_scan_frames<<<1, 1024, 0, c->stream>>>(c->d_tx_fpar, (uint32_t)n_frames);
    }
    k_tx_modulate<<<(unsigned)((nsym_total + 63) / 64), 64, 0, c->stream>>>(c->d_tx_codes, c->d_tx_fpar, (const double2*)c->d_tx_phases, nsym,
                                                                          nsym_total, (int*)d_iq_out, c->d_tx_cnt, c->d_tx_list, kAmbCap,
                                                                          c->tx_flat_lo, c->tx_flat_hi, c->d_tx_flat);
    HIPCHK(hipGetLastError());
    uint32_t n_amb = 0;
    HIPCHK(hipMemcpyAsync(&n_amb, c->d_tx_cnt, sizeof n_amb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                                // (also: `frames` is the caller's again)
    if (n_amb > kAmbCap) return fail(OPV_ECAPACITY, "too many ambiguous samples (internal)");
    if (n_amb) {  // re-evaluate with libm exactly like the reference: tone / sign and phases of those symbols on the host
        std::vector<uint64_t> list(n_amb);
        HIPCHK(hipMemcpy(list.data(), c->d_tx_list, sizeof(uint64_t) * n_amb, hipMemcpyDeviceToHost));
        std::vector<int8_t> amp(nsym_total, 0);
        opv_tx_symbol_codes(frames, n_frames, amp.data());
        std::vector<double> ph(2 * OPV_TX_CKPT_SYMS);
        for (uint64_t n : list) {
            const size_t sym = n / OPV_SPS, ck = sym / OPV_TX_CKPT_SYMS, in = sym - ck * OPV_TX_CKPT_SYMS;
            double p[2];
            opv_tx_checkpoint_range(ck, 1, p);
            opv_tx_symbol_phases(0, in + 1, &p[0], &p[1], ph.data());
            int16_t iq[2];
            opv_tx_sample_exact(ph[2 * in], ph[2 * in + 1], amp[sym], (int)(n % OPV_SPS), &iq[0], &iq[1]);
            HIPCHK(hipMemcpy(d_iq_out + 2 * n, iq, 4, hipMemcpyHostToDevice));
        }
    }
    return (long)n_amb;
}


extern "C" long opv_tx_modulate_device_to_host(opv_ctx* c, const uint8_t* frames, size_t n_frames, int16_t* iq_out) {
    if (!c || !iq_out) return fail(OPV_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t n = (n_frames * OPV_FSYMS + 100) * (size_t)OPV_SPS;
    int16_t* d = nullptr;
    HIPCHK(hipMalloc(&d, n * 4));
    long rc = opv_tx_modulate_device(c, frames, n_frames, d);
    if (rc >= 0 && hipMemcpy(iq_out, d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(OPV_EHIP, "D2H of the modulated samples");
    (void)hipFree(d);
    return rc;
}

// ---- multi-GPU: the one collective of the path (ncclGather, /opt/rocm/include/rccl/rccl.h:745) ---------------
// RCCL is bound at first use (dlopen: a process that loaded PyTorch gets PyTorch's copy, a stand-alone C++ host the
// system's), so a single-GPU caller never needs it. Prototypes restated from rccl.h; ncclUniqueId is 128 opaque bytes
// passed by value.
namespace {
struct RcclId { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        auto sym = [&](const char* n) { return dlsym(x.lib, n); };
        x.GetUniqueId = (int (*)(RcclId*))sym("ncclGetUniqueId");
        x.CommInitRank = (int (*)(void**, int, RcclId, int))sym("ncclCommInitRank");
        x.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        x.Gather = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclGather");
        x.GroupStart = (int (*)())sym("ncclGroupStart");
        x.GroupEnd = (int (*)())sym("ncclGroupEnd");
        x.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.Gather && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}
int rccl_fail(const char* what, int rc) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
    g_err = buf;
    return OPV_EHIP;
}
constexpr int kNcclUint8 = 1, kNcclInt32 = 2;   // ncclDataType_t (rccl.h:459-461)
}  // namespace

extern "C" int opv_comm_unique_id(char out128[128]) {
    if (!out128) return fail(OPV_EINVAL, "null argument");
    if (!rccl().ok) return fail(OPV_ENODEV, "librccl.so.1 could not be loaded (multi-GPU gather needs RCCL)");
    RcclId id;
    if (int rc = rccl().GetUniqueId(&id)) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(out128, id.internal, 128);
    return OPV_OK;
}

extern "C" int opv_comm_init(void** comm, int world, int rank, const char id128[128], int device) {
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return fail(OPV_EINVAL, "opv_comm_init: bad arguments");
    if (!rccl().ok) return fail(OPV_ENODEV, "librccl.so.1 could not be loaded (multi-GPU gather needs RCCL)");
    HIPCHK(hipSetDevice(device));
    RcclId id;
    std::memcpy(id.internal, id128, 128);
    if (int rc = rccl().CommInitRank(comm, world, id, rank)) return rccl_fail("ncclCommInitRank", rc);
    return OPV_OK;
}

extern "C" int opv_comm_init_all(void** comms, int n_devices, const int* devices) {
    if (!comms || !devices || n_devices < 1) return fail(OPV_EINVAL, "opv_comm_init_all: bad arguments");
    if (!rccl().ok) return fail(OPV_ENODEV, "librccl.so.1 could not be loaded (multi-GPU gather needs RCCL)");
    RcclId id;
    if (int rc = rccl().GetUniqueId(&id)) return rccl_fail("ncclGetUniqueId", rc);
    for (int i = 0; i < n_devices; ++i) comms[i] = nullptr;
    if (int rc = rccl().GroupStart()) return rccl_fail("ncclGroupStart", rc);     // one thread, several devices: inits must be grouped
    int rc = 0;
    bool hip_ok = true;
    for (int i = 0; i < n_devices && !rc && hip_ok; ++i) {
        hip_ok = hipSetDevice(devices[i]) == hipSuccess;
        if (hip_ok) rc = rccl().CommInitRank(&comms[i], n_devices, id, i);
    }
    const int rc2 = rccl().GroupEnd();
    if (!hip_ok || rc || rc2) {                          // nothing half-made is handed back
        for (int i = 0; i < n_devices; ++i) { if (comms[i]) (void)rccl().CommDestroy(comms[i]); comms[i] = nullptr; }
        if (!hip_ok) return fail(OPV_ENODEV, "opv_comm_init_all: hipSetDevice failed (device ordinal out of range?)");
        return rccl_fail(rc ? "ncclCommInitRank" : "ncclGroupEnd", rc ? rc : rc2);
    }
    return OPV_OK;
}

extern "C" void opv_comm_destroy(void* comm) {
    if (comm && rccl().ok) (void)rccl().CommDestroy(comm);
}

// the gathers of ranks [0, n) that live in THIS thread (n = 1: one process per GPU), inside one RCCL group
static int gather_group(opv_ctx* const* ctxs, void* const* comms, int n, int root, uint8_t* d_frames_all, int32_t* d_counts_all) {
    if (!rccl().ok) return fail(OPV_ENODEV, "librccl.so.1 could not be loaded (multi-GPU gather needs RCCL)");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !comms[i]) return fail(OPV_EINVAL, "null context or communicator");
        if (ctxs[i]->n_streams != ctxs[0]->n_streams || ctxs[i]->cap_frames != ctxs[0]->cap_frames)
            return fail(OPV_EINVAL, "opv_gather_frames: every context must have the same n_streams and max_samples");
    }
    if (int rc = rccl().GroupStart()) return rccl_fail("ncclGroupStart", rc);
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) {
        opv_ctx* c = ctxs[i];
        if (hipSetDevice(c->cfg.device) != hipSuccess) { rc = -1; break; }
        const size_t nb = (size_t)OPV_FB * c->cap_frames * (size_t)c->n_streams;
        // on the context's stream: behind the kernels of its last opv_process
        rc = rccl().Gather(c->d_frames, d_frames_all, nb, kNcclUint8, root, comms[i], c->stream);
        if (!rc) rc = rccl().Gather(c->d_counts, d_counts_all, (size_t)c->n_streams, kNcclInt32, root, comms[i], c->stream);
    }
    const int rc2 = rccl().GroupEnd();
    if (rc == -1) return fail(OPV_EHIP, "hipSetDevice failed");
    if (rc) return rccl_fail("ncclGather", rc);
    if (rc2) return rccl_fail("ncclGroupEnd", rc2);
    return OPV_OK;
}

extern "C" int opv_gather_frames(opv_ctx* c, void* comm, int root, uint8_t* d_frames_all, int32_t* d_counts_all) {
    return gather_group(&c, &comm, 1, root, d_frames_all, d_counts_all);
}

extern "C" int opv_gather_frames_all(opv_ctx* const* ctxs, void* const* comms, int n, int root, uint8_t* d_frames_all,
                                     int32_t* d_counts_all) {
    if (!ctxs || !comms || n < 1) return fail(OPV_EINVAL, "opv_gather_frames_all: bad arguments");
    return gather_group(ctxs, comms, n, root, d_frames_all, d_counts_all);
}
