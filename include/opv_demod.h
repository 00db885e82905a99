/* include/opv_demod.h — C ABI of the MI355X-native OPV MSK receive chain.
 *
 * This is the drop-in boundary. The reference (OpenResearchInstitute/opv-cxx-demod) has no
 * library/plugin API: its hot path sits behind three C++ objects that only main() uses
 * (src/opv-demod.cpp:999-1001, :1164, :1182-1183) and behind the `opv-demod` process
 * contract. Each entry point below names the reference interface it replaces. All state
 * lives in device memory owned by an opv_ctx; signatures carry plain pointers and sizes
 * only. A context is not thread-safe; distinct contexts are independent.
 *
 * Return convention: 0 on success, negative OPV_E* on error (never a silent CPU fallback:
 * if no HIP device / kernel image is usable, opv_create fails with OPV_ENODEV).
 */
#ifndef OPV_DEMOD_H
#define OPV_DEMOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPV_ABI_VERSION 7   /* 7: opv_process never waits on the host again - the host-libm decision of offset-search near-ties runs as a host
                               function IN STREAM ORDER between the search and the front-end (+ opv_offset_ties_decided_on_host, opv_offset_ties_left_to_device); opv_set_frontend
                               takes 0 / 1 / 4 / 16 only (the comparison mappings are gone); opv_tap_occupancy's third entry is k_msk_frontend_x16_wg4;
                               6: offset-search near-ties are decided with the HOST's libm (opv_offset_ties_on_host), one host wait in the round
                               in which a stream's search runs; + opv_push_iq_batch_async / opv_push_wait; opv_push_iq_batch moves blocks in
                               pinned host memory with one gather kernel;
                               5: opv_set_frontend takes 16 (sixteen streams per wavefront; automatic from 8193 streams) and answers OPV_EINVAL to
                               the comparison mappings -1 / -2 unless built with them; opv_create refuses non-finite -o / -a / -p values;
                               an idle opv_process launches nothing for callers that never pop (zero-copy path);
                               4: + opv_tx_stream_* (the host modulator with its state carried from call to call), opv_tap_tx_frame;
                               3: + opv_frontend_kernel, opv_tx_bert_frames, opv_tx_modulate_device_to_host, opv_tap_tx_checkpoints,
                               opv_comm_* / opv_gather_frames(_all); opv_tx_modulate_device runs the whole chain on the device;
                               an opv_process with nothing new still resumes streams held back by back-pressure */

#define OPV_SAMPLES_PER_SYMBOL 40    /* src/opv-demod.cpp:39  */
#define OPV_FRAME_BYTES 134          /* :49  */
#define OPV_FRAME_BITS 1072          /* :50  */
#define OPV_ENCODED_BITS 2144        /* :51  */
#define OPV_FRAME_SYMBOLS 2168       /* :52  */
#define OPV_CHUNK_SAMPLES 86720      /* :1012 streaming chunk = one frame of samples */

enum {
    OPV_OK = 0,
    OPV_EINVAL = -1,   /* bad argument                                   */
    OPV_ENODEV = -2,   /* no usable HIP device / kernels failed to load  */
    OPV_ENOMEM = -3,   /* device or host allocation failed               */
    OPV_ECAPACITY = -4,/* per-stream capacity (opv_cfg.max_samples) exceeded: unprocessed samples + a push do not fit
                          (frames waiting to be popped hold their stream back; pop them and push again) */
    OPV_EHIP = -5,     /* a HIP runtime call failed (see opv_last_error) */
    OPV_ESTATE = -6    /* call not valid in this state (e.g. push after flush) */
};

/* Sync tracker states (enum class SyncState, src/opv-demod.cpp:73). */
enum { OPV_HUNTING = 0, OPV_VERIFYING = 1, OPV_LOCKED = 2 };

/* One entry per stderr line SyncTracker::process prints (src/opv-demod.cpp:651,677,695,699,705). */
enum {
    OPV_EV_HUNT_TO_VERIFY = 1,
    OPV_EV_VERIFY_TO_LOCK = 2,
    OPV_EV_SYNC_OK = 3,
    OPV_EV_SYNC_MISS = 4,
    OPV_EV_LOST_LOCK = 5
};

/* Replaces the flag parsing of main() (src/opv-demod.cpp:944-974) for the flags that reach
 * the hot path: -s, -o <hz>, -a <alpha>, and -c / -p <hz> (the batch coherent demodulator :365-572, prefix parity
 * only - its loop is chaotic); -q/-r only affect host printing. */
typedef struct opv_cfg {
    int32_t streaming;        /* 1: -s chunked semantics (:995-1125); 0: batch (:1132-1216) */
    int32_t have_init_offset; /* -o given: skip the offset search in streaming mode (:1004,:1031) */
    double init_offset_hz;    /* -o value */
    double afc_alpha;         /* -a value; 0.001 if <= 0 is NOT substituted: pass 0.001 for the default (:945) */
    int32_t device;           /* HIP device ordinal */
    int32_t coherent;         /* -c: Costas-loop demodulator instead of the energy detector; honoured in batch mode only,
                                 like the reference (:1144; with -s it merely changes the banner, :983-984,995) */
    uint64_t max_samples;     /* per-stream DEVICE BUFFER capacity in IQ samples (< 2^31). Pushed streams may be
                                 arbitrarily long: consumed samples / soft symbols are dropped when the buffer
                                 fills (>= ~3 chunks + the largest push is enough); an attached capture must fit */
    double pll_bw_hz;         /* -p value (coherent mode, set_pll_bandwidth :551-558); the reference's default is 50 */
} opv_cfg;

/* Per decoded frame: what main() knows when it prints/writes a frame
 * (src/opv-demod.cpp:1048-1062: metric, res.sync_quality, sym index of release). */
typedef struct opv_frame_meta {
    int32_t viterbi_metric;   /* ViterbiDecoder::decode return (:845); 0 == "perfect" (:910) */
    int32_t sync_ok;          /* 1: the sync word in front of this payload passed its check (HUNTING hit :642, or
                                 LOCKED corr >= 0.70 :688); 0: flywheel frame released after a sync MISS (:709-712) */
    double sync_quality;      /* SyncTracker::Result::sync_quality (:611) */
    uint64_t release_symbol;  /* global symbol index at which the frame was released (:1046) */
    uint64_t payload_symbol;  /* global symbol index of the first payload symbol */
} opv_frame_meta;

typedef struct opv_event {
    int32_t kind;             /* OPV_EV_* */
    int32_t count;            /* frame number (VERIFY_TO_LOCK) or miss number (SYNC_MISS) */
    uint64_t sym_idx;         /* the [%zu] printed by the reference */
    double corr;              /* normalised correlation */
    double raw;               /* raw correlation (HUNT_TO_VERIFY line) */
} opv_event;

/* What the status / summary lines of main() read (src/opv-demod.cpp:1080-1082,1117-1120). */
typedef struct opv_stream_state {
    double freq_offset_hz;    /* MSKDemodulatorAFC::get_freq_offset (:331) */
    double timing_freq;       /* get_timing_freq (:332) */
    double est_offset_hz;     /* estimate_offset result (:1032/:1166); NaN if it did not run */
    double mu;                /* fractional timing carry (:343) */
    uint64_t total_symbols;   /* symbols demodulated so far */
    uint64_t total_samples;   /* samples consumed by completed demodulate() calls (:1027) */
    uint64_t chunk_origin;    /* sample index at which the next chunk starts */
    int32_t sync_state;       /* OPV_HUNTING / VERIFYING / LOCKED (:738) */
    int32_t frames_released;  /* SyncTracker::get_total_frames (:739) */
    int32_t frames_decoded;   /* frames with metric >= 0 (`decoded`, :1053) */
    int32_t frames_perfect;   /* metric == 0 (`perfect`, :1054) */
    int32_t n_chunks;         /* demodulate() calls made */
    int32_t flushed;
    uint32_t events_dropped;  /* tracker events overwritten before opv_pop_events read them (the event log is lossy) */
    uint32_t edge_ties;       /* symbols whose tone choice the reference decides by the rounding of its own LO: a window
                                 with exactly one non-zero tap next to digital silence. Each may move the AFC by one
                                 step (<= 27 Hz, decaying) away from the reference; 0 on any capture without exact zeros */
    int32_t stalled;          /* back-pressure after the last opv_process: bit 0 = soft-symbol ring full, bit 1 = ring of
                                 unpopped frames full. The stream resumes at the next opv_process after opv_pop_frames */
    int32_t offset_ties;      /* offset-search candidates that were within 1e-11 (relative energy) of the winner and were
                                 therefore re-evaluated in the reference's own order of operations (estimate_offset :143-159): on
                                 the device, and once more on the host with the host's sin / cos (the reference's libm) if two of
                                 them are then still within 2e-13 of each other and opv_offset_ties_on_host() is 1. Both bands are
                                 scaled by the input's power where the correlation is weak against it (csrc/k_offset_search.hip) */
} opv_stream_state;

typedef struct opv_ctx opv_ctx;

/* ---- lifetime: replaces constructing MSKDemodulatorAFC + SyncTracker + FrameDecoder
 *      (src/opv-demod.cpp:999-1001 / :1164,:1182-1183) for n_streams independent captures */
int opv_create(opv_ctx** out, int n_streams, const opv_cfg* cfg);
/* Environment variables. The library reads FOUR, all of them TEST HOOKS - never set in production -, all of them in opv_create
 * and nowhere else (a context's behaviour is fixed when it is created; grep getenv csrc/):
 *   OPV_OFFSET_DISTRUST_LIBM  (any value) the offset search's last-place ties are decided by the device's sincos although the host's
 *                             libm reproduces the reference's (opv_offset_ties_on_host() == 0): moves a stream whose search ties
 *                             from the class "decided like the reference" to "decided by another libm". The fallback's tests use it.
 *   OPV_TX_DISTRUST_LIBM      (any value) the device transmit chain takes every symbol's flat-top bits from the host's libm instead of
 *                             the zone rule its one-time probe of sin / cos allows: same samples, slower set-up. The fallback's test uses it.
 *   OPV_PUSH_NO_GATHER        (any value) opv_push_iq_batch(_async) moves pinned blocks with one copy per block instead of one gather
 *                             kernel: same bytes, the path pageable sources take anyway. Its test runs both.
 *   OPV_PUSH_GATHER_BLOCKS    (a number > 0) workgroups of that gather kernel (default 32; a measurement knob, results unaffected). */
void opv_destroy(opv_ctx* ctx);
const char* opv_last_error(void);
int opv_abi_version(void);

/* ---- host-buffer path -----------------------------------------------------------------
 * opv_push_iq replaces the stdin reader + chunker (src/opv-demod.cpp:1021-1026 streaming,
 * :1132-1135 batch): host-endian interleaved int16 I,Q. The caller keeps ownership; the
 * samples are copied to the stream's device buffer. Nothing is computed until opv_process. */
int opv_push_iq(opv_ctx* ctx, int stream, const int16_t* iq_interleaved, size_t n_samples);
/* The same for several streams of a multi-stream server in one call (what N copies of the reader loop
 * :1021-1026 do in N reference processes): all copies are enqueued, then awaited once. Stops at the first
 * error; streams before it have been pushed.
 * Blocks that lie in PINNED host memory (hipHostMalloc / hipHostRegister; 4-byte aligned) cross PCIe together, read by one
 * gather kernel through their device-visible addresses - 55 GB/s for 1536 blocks of 347 KB where one copy per stream reaches 21
 * (scripts/microbench/h2d_many.hip) - and staging buffers that fill in the same round are compacted by one launch; pageable
 * blocks take one hipMemcpyAsync each, as opv_push_iq does. Results do not depend on the route. */
int opv_push_iq_batch(opv_ctx* ctx, int count, const int* streams, const int16_t* const* iq_interleaved,
                      const size_t* n_samples);
/* The same without the wait at its end, for a server that double-buffers its host memory: returns once the moves are ENQUEUED; the
 * blocks must stay valid and unchanged until opv_push_wait returns (every other opv_push_* call and opv_reset_stream wait first
 * by themselves). opv_process may be called in between - its kernels queue behind every move enqueued so far, on the device - so
 * that round r + 1 crosses PCIe while the kernels of round r run and the frames of round r are popped:
 *     opv_push_iq_batch_async(r = 0);  loop: opv_push_wait; opv_process; opv_push_iq_batch_async(r + 1); opv_sync; opv_pop_frames...
 * A serving round then costs max(PCIe, kernels + pops) instead of their sum (bin/opv-live-capacity --pipelined). */
int opv_push_iq_batch_async(opv_ctx* ctx, int count, const int* streams, const int16_t* const* iq_interleaved,
                            const size_t* n_samples);
int opv_push_wait(opv_ctx* ctx);
/* EOF on a stream: enables the tail processing of :1088-1113 (streaming) or the single
 * whole-capture demodulate of :1166-1173 (batch) at the next opv_process. */
int opv_flush(opv_ctx* ctx, int stream);

/* ---- device-resident path -------------------------------------------------------------
 * Zero-copy variant of push+flush for captures already in HBM (bench, multi-stream
 * servers): d_iq is a DEVICE pointer (16-byte aligned) to n_samples interleaved int16 IQ
 * that must stay valid until the context is destroyed or the stream is reset. */
int opv_attach_device_iq(opv_ctx* ctx, int stream, const int16_t* d_iq, size_t n_samples, int eof);

/* Runs the hot path on everything that is ready, for all streams, in four launches on the
 * context's HIP stream: offset search (estimate_offset :131-202), MSK front-end
 * (demodulate :206-329 incl. the chunker :1026-1076), sync tracker (:615-736) and frame
 * decode (FrameDecoder::decode :854-898). Asynchronous in every round; opv_sync waits. (A search whose candidates tie in the
 * last places of sin / cos is decided with the host's libm before the front-end starts - by a host function enqueued on the
 * context's stream between the two kernels, not by the caller: see opv_offset_ties_on_host.) */
int opv_process(opv_ctx* ctx);
int opv_sync(opv_ctx* ctx);
/* Stream-to-wavefront mapping of the front-end kernel (no counterpart in the reference, which is one thread
 * per process): 1 = one wavefront per stream (lowest per-symbol latency; right while the GPU has idle SIMDs),
 * 4 = four streams per wavefront (fewer issued instructions per symbol; right when every SIMD has work),
 * 16 = sixteen streams per wavefront, one per DPP quad (fewest issued instructions per symbol and stream: 38 against 88 and 156;
 *      right from ~8 000 streams per context, where 16 per wave still put a wave on every second SIMD),
 * 0 = automatic (4 from 2049 streams per context, 16 from 8193; measured cross-overs on MI355X).
 * Anything else is OPV_EINVAL. Results do not depend on the mapping beyond the fp64 re-association level of the soft symbols
 * (all decisions identical; the tests run every mapping and every launch shape against the oracle). */
int opv_set_frontend(opv_ctx* ctx, int streams_per_wave);
/* Name of the front-end kernel the LAST opv_process launched ("k_msk_frontend_rb", "..._rb_wg4", "k_msk_frontend_x4_wg4", "k_msk_frontend_x16_wg4", ...;
 * "" before the first round): what a profile or a bench line should be read against. No counterpart in the reference. */
const char* opv_frontend_kernel(opv_ctx* ctx);
/* Restores a stream (stream = -1: every stream) to its freshly-created state (keeps buffers). */
int opv_reset_stream(opv_ctx* ctx, int stream);

/* Measurement hook: when enabled, opv_process brackets each of its four hot-path kernels
 * with HIP events on the context's stream. opv_kernel_times (implies opv_sync) returns the
 * durations in milliseconds of the LAST opv_process: [0] offset search, [1] MSK front-end,
 * [2] sync tracker, [3] frame decode. */
int opv_enable_timing(opv_ctx* ctx, int enable);
int opv_kernel_times(opv_ctx* ctx, float ms_out[4]);

/* ---- results --------------------------------------------------------------------------
 * opv_pop_frames replaces the frame writer (src/opv-demod.cpp:1052-1062): frames whose
 * decoder returned -1 (silent frame, :859) are skipped exactly as the reference skips
 * them. Copies up to cap_frames not-yet-popped frames (134 B each, in release order) and
 * their meta; returns the number copied or a negative error. Implies opv_sync.
 * Frames are never dropped: a stream whose ring of unpopped frames is full pauses (opv_stream_state.stalled)
 * and resumes at the next opv_process after a pop.
 * opv_pop_events returns the tracker lines the reference prints to stderr (:651,677,695,699,705). Reading them
 * is OPTIONAL (opv-modem discards the child's stderr): the log is a ring of the most recent entries per stream
 * (4 per frame of capacity + 64); lines not read in time are overwritten and counted in
 * opv_stream_state.events_dropped, nothing else is affected. */
long opv_pop_frames(opv_ctx* ctx, int stream, uint8_t* out134, size_t cap_frames, opv_frame_meta* meta);
long opv_pop_events(opv_ctx* ctx, int stream, opv_event* out, size_t cap_events);
int opv_get_state(opv_ctx* ctx, int stream, opv_stream_state* out);

/* Device-side views for zero-copy consumers (RCCL gather of decoded frames): frames are
 * [n_streams][frame_capacity][134] uint8, metrics [n_streams][frame_capacity] int32 (-1 =
 * dropped, INT32_MIN = not decoded yet), counts [n_streams] int32 = frames released. */
int opv_device_frames(opv_ctx* ctx, const uint8_t** d_frames, const int32_t** d_metrics,
                      const int32_t** d_counts, size_t* frame_capacity);
void* opv_hip_stream(opv_ctx* ctx);

/* ---- multi-GPU (BASELINE configs[4]; no counterpart in the reference, whose streams are separate processes) -----------
 * Streams shard contiguously over GPUs - one context per GPU, no data-path collective - and the decoded frames return to one
 * rank with ONE gather: ncclGather (/opt/rocm/include/rccl/rccl.h:745) of every context's [n_streams][frame_capacity][134]
 * frame buffer and [n_streams] counts, on the context's HIP stream behind the kernels of the last opv_process
 * (asynchronous; opv_sync waits). d_frames_all / d_counts_all: DEVICE buffers on the root's GPU of world x that size, in
 * rank = global stream order (ignored on other ranks). All contexts of a communicator must have the same n_streams and
 * max_samples. `comm` is an ncclComm_t: the caller's own (a C++ host that links RCCL), or one made here - RCCL is bound with
 * dlopen at first use, so that callers need no RCCL headers and single-GPU callers no RCCL at all:
 *   one process per GPU:   rank 0: opv_comm_unique_id(id), hand the 128 bytes to the other ranks, everyone opv_comm_init,
 *                          then opv_gather_frames once per round;
 *   one process, N GPUs:   opv_comm_init_all(comms, N, devices) (rank i on devices[i]), then opv_gather_frames_all(ctxs, comms,
 *                          N, ...) once per round: the N ranks' gathers issued by one thread inside one RCCL group. */
int opv_comm_unique_id(char out128[128]);
int opv_comm_init(void** comm, int world, int rank, const char id128[128], int device);
int opv_comm_init_all(void** comms, int n_devices, const int* devices);
void opv_comm_destroy(void* comm);
int opv_gather_frames(opv_ctx* ctx, void* comm, int root, uint8_t* d_frames_all, int32_t* d_counts_all);
int opv_gather_frames_all(opv_ctx* const* ctxs, void* const* comms, int n, int root, uint8_t* d_frames_all,
                          int32_t* d_counts_all);

/* ---- parity taps (debug): the intermediates the 1e-5 contract is checked on ------------ */
/* soft symbols by absolute symbol index; only the retained tail of a long pushed stream is available */
long opv_tap_soft(opv_ctx* ctx, int stream, uint64_t first_symbol, double* out, size_t cap);
/* per demodulate() call, starting at call number first_chunk: {freq_offset, timing_freq, mu,
 * leftover, n_symbols}. The log is a ring of the most recent calls. */
long opv_tap_chunks(opv_ctx* ctx, int stream, uint32_t first_chunk, double* out5, size_t cap_chunks);
/* 134 candidate energies of the offset search in scan order (121 coarse, 13 fine) */
int opv_tap_offset_energies(opv_ctx* ctx, int stream, double* out134);
/* Who decides the offset search's last-place ties (estimate_offset's strict '>' between candidates whose energies, evaluated in
 * the reference's order of operations, are within 2e-13 of each other - what the last places of sin / cos can move - or equal,
 * src/opv-demod.cpp:161,195): 1 = the host, with the contenders evaluated by the reference's own loop on the
 * host's libm - opv_create found that this process's sin / cos reproduce a pinned reference energy (csrc/opv_offset_host.cpp);
 * 0 = the device's sincos (another libm on the host, a host that cannot pin the staging area or enqueue host functions, or the
 * test hook OPV_OFFSET_DISTRUST_LIBM, see opv_create): still the reference's order of
 * operations, counted in offset_ties, but an exact tie is then decided by a different libm than the reference's. */
int opv_offset_ties_on_host(opv_ctx* ctx);
/* Streams whose tie the host has decided so far in this context (they are decided in stream order behind their search kernel,
 * opv_process does not wait for them: the number is final for a round after opv_sync). Diagnostic. */
uint64_t opv_offset_ties_decided_on_host(opv_ctx* ctx);
/* Streams that were listed for the host beyond what one round stages (8 passes x min(n_streams, 512) slots = up to 4096 streams of
 * ONE context whose search ties at the last-place level in ONE round) and therefore kept the device's decision, as under
 * opv_offset_ties_on_host() == 0. Exact mirror ties (real-valued captures: the one systematic source) come out the same either
 * way; for others this is a parity class to report. 0 in every test and bench run. Diagnostic. */
uint64_t opv_offset_ties_left_to_device(opv_ctx* ctx);

/* Where and how fast the wavefront that served `stream` ran in the LAST front-end launch (with four streams per
 * wave, the four share these numbers): out[0] = HW_REG_HW_ID, out[1] = HW_REG_XCC_ID, out[2] = shader-clock cycles (s_memtime) and
 * out[3] = 100 MHz ticks (s_memrealtime) spent inside the kernel. Diagnostic: placement census and the clock the
 * chip held (cycles / ticks x 100 MHz); no counterpart in the reference. */
int opv_tap_wave_info(opv_ctx* ctx, int stream, uint64_t out[4]);

/* Workgroups of each hot-path kernel the runtime can keep resident per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor), in the
 * order k_msk_frontend_rb, _rb_wg4, k_msk_frontend_x16_wg4, k_msk_frontend_x4_wg4, k_frame_decode, k_frame_scale. Diagnostic. */
int opv_tap_occupancy(opv_ctx* ctx, int out[6]);

/* Stand-alone FrameDecoder::decode (src/opv-demod.cpp:854-898) on n_frames payloads of
 * 2144 host doubles each. Optional taps: q (quantised, :862-866), deint (:869-871),
 * bits (Viterbi hard decisions, :874-875). metrics[i] = path metric or -1. */
int opv_decode_payloads(opv_ctx* ctx, const double* soft, size_t n_frames, uint8_t* out134,
                        int32_t* metrics, int8_t* q, int8_t* deint, uint8_t* bits);

/* ---- signal source (reference src/opv-mod.cpp) ---------------------
 * Host-side, bit-identical to `opv-mod`: BERT frames (:339-361) and the whole
 * encode->interleave->MSK chain incl. 100 trailing zero symbols (:473-529). */
void opv_tx_bert_frame(const char* callsign, uint32_t token, uint32_t frame_num, uint8_t out134[OPV_FRAME_BYTES]);
/* n_frames consecutive BERT frames (frame numbers first_frame, first_frame + 1, ...) into out134[n_frames][134] */
void opv_tx_bert_frames(const char* callsign, uint32_t token, uint32_t first_frame, size_t n_frames, uint8_t* out134);
size_t opv_tx_modulated_samples(size_t n_frames);
size_t opv_tx_modulate(const uint8_t* frames134, size_t n_frames, int16_t* iq_out);
/* The same modulator as an object that carries its state from call to call (the reference's HDLModulator,
 * src/opv-mod.cpp:219-291: two free-running NCOs, the differential sign, the symbol parity), for a source that produces frames
 * as it goes - `opv-mod -R` on a live pipe (:473-498), `opv-mod -c` (:503-524: one reset per pass over the BERT frames).
 * create = a modulator after reset() (:221-226); opv_tx_stream_frames appends n_frames x 2168 x 40 samples to iq_out and
 * returns that count; opv_tx_stream_tail writes the 100 silent symbols that end a run (:528-529; 4000 samples, no state).
 * Any split of a run into calls gives the bytes of one opv_tx_modulate call. Host only, needs no device. */
/* Parity tap of the bit-level half (src/opv-mod.cpp:158-213): one frame after the randomiser (134 bytes), after the
 * convolutional encoder (2144 bits, one per byte, encoder order) and after the interleaver (on-air order). Any output may be
 * NULL. What `opv-mod -v` prints the first bytes / bits of (:171-183,198-209,326-329). */
void opv_tap_tx_frame(const uint8_t* frame134, uint8_t* randomized134, uint8_t* coded2144, uint8_t* interleaved2144);
typedef struct opv_tx_stream opv_tx_stream;
opv_tx_stream* opv_tx_stream_create(void);
void opv_tx_stream_reset(opv_tx_stream* st);
size_t opv_tx_stream_frames(opv_tx_stream* st, const uint8_t* frames134, size_t n_frames, int16_t* iq_out);
size_t opv_tx_stream_tail(int16_t* iq_out);
void opv_tx_stream_destroy(opv_tx_stream* st);
/* Device-side transmit chain: same result as opv_tx_modulate, written straight into HBM
 * (d_iq_out: device pointer, 16-byte aligned, opv_tx_modulated_samples(n_frames) samples). The whole chain of
 * src/opv-mod.cpp:97-291 runs on the device: randomiser, convolutional encoder, interleaver and sync word per frame
 * (k_tx_encode), the differential sign as a parity prefix over the run (k_tx_scan_frames), the two free-running NCOs
 * (k_tx_expand_phases, from a build-time table of their state every 128th symbol; once per context and run length, shared by
 * every stream), sample synthesis (k_tx_modulate). `frames134` is a host pointer (134 bytes per frame cross PCIe). Samples
 * whose truncation could differ between device sincos and libm are re-evaluated on the host. Returns the number of samples so
 * patched (>= 0, normally 0) or a negative error. Synchronous. */
long opv_tx_modulate_device(opv_ctx* ctx, const uint8_t* frames134, size_t n_frames, int16_t* d_iq_out);
/* The same chain for a host that wants the samples back (`opv-mod -G`): a temporary device buffer, then D2H into iq_out
 * (host, opv_tx_modulated_samples(n_frames) samples). */
long opv_tx_modulate_device_to_host(opv_ctx* ctx, const uint8_t* frames134, size_t n_frames, int16_t* iq_out);
/* Parity tap: entries [first, first + count) of the NCO checkpoint sequence the device transmit chain starts from - (ph1, ph2)
 * of src/opv-mod.cpp:274-279 at symbol 128 * entry of a run - from the build-time table, beyond it (4096 frames) from the host
 * continuation. Host only, needs no device. */
void opv_tap_tx_checkpoints(size_t first, size_t count, double* out2);
/* Device-side channel tool for synthetic multi-stream workloads:
 * d_out[n] = clip(rint(gain * d_in[n] * exp(j 2 pi f0 n / Fs) + sigma * N(0,1)+jN(0,1))),
 * noise from a counter-based generator keyed by (seed, n). d_in/d_out: device int16 IQ. */
int opv_channel_device(opv_ctx* ctx, const int16_t* d_in, int16_t* d_out, size_t n_samples,
                       double gain, double f0_hz, double sigma, uint64_t seed);
/* Sample-clock error for the same tool chain: d_out[n] = rint(linear interpolation of d_in
 * at n (1 + clock_ppm 1e-6)), i.e. the capture as an ADC running clock_ppm parts per million fast would have
 * taken it. Returns the number of samples written, floor(n_in / (1 + clock_ppm 1e-6)) <= out_capacity, or a
 * negative error. Asynchronous on the context's stream like opv_channel_device. */
long opv_resample_device(opv_ctx* ctx, const int16_t* d_in, size_t n_in, int16_t* d_out, size_t out_capacity,
                         double clock_ppm);

#ifdef __cplusplus
}
#endif
#endif /* OPV_DEMOD_H */
