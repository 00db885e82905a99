// opv_device.h — device-resident per-stream context shared by the HIP kernels and the
// C-ABI host shim. One OpvStream per independent IQ capture; it is the "checkpoint" of
// the carry of MSKDemodulatorAFC (reference src/opv-demod.cpp:337-347), of
// SyncTracker (:759-780) and the chunker of main() (:1012-1076), plus device log pointers.
#pragma once
#include <stdint.h>

#define OPV_SPS 40
#define OPV_SYNC_BITS 24
#define OPV_FB 134
#define OPV_FBITS 1072
#define OPV_CODED 2144
#define OPV_FSYMS 2168
#define OPV_CHUNK 86720
#define OPV_SYNC_WORD 0x02B8DBu

#define OPV_OFFS_TERMS 10                     // Taylor terms of the offset search's one-pass evaluation (k_offset_search.hip)

#define OPV_TILE_SAMPLES 2048                 // HBM->LDS staging unit of int16 IQ (two tiles = a power-of-two ring)
#define OPV_TILE_BYTES (OPV_TILE_SAMPLES * 4) // 8192 B = 8 wave-wide 16 B/lane loads

struct OpvFrameRec {       // written by k_sync_track, read by k_frame_decode and the host
    uint64_t payload_sym;  // index of first payload soft symbol in the soft log
    uint64_t release_sym;  // symbol index at which the reference releases the frame
    double quality;        // sync_quality_
    int32_t sync_ok;       // 1: this frame's sync word passed its check (HUNTING hit, or LOCKED corr >= 0.70); 0: flywheel
    int32_t pad;
};

struct OpvEventRec {  // mirrors opv_event
    int32_t kind;
    int32_t count;
    uint64_t sym_idx;
    double corr;
    double raw;
};

struct OpvStream {
    // ---- inputs / logs (device pointers) ----
    const int16_t* iq;   // sample 0 of the capture, interleaved I,Q, 16-byte aligned
    uint64_t n_avail;    // samples available behind iq
    int32_t eof;         // no more samples will arrive
    int32_t pad0;
    double* soft;        // soft-symbol log: a ring, symbol number i lives at soft[i & (cap_soft-1)]
    uint64_t cap_soft;   // power of two, > everything one opv_process can produce + one frame
    uint64_t iq_base;    // absolute index of iq[0] (bookkeeping; kernels index relative to iq)
    uint32_t frames_popped, events_popped;  // consumer cursors: frame/event records are rings
    OpvFrameRec* frec;   // frame records
    OpvEventRec* events;
    double* chunk_log;   // 5 doubles per demodulate() call
    uint8_t* frames;     // [cap_frames][134]
    int32_t* metrics;    // [cap_frames]
    double* fscale;      // [cap_frames] quantiser scale of each released frame (k_frame_scale -> k_frame_decode)
    uint32_t cap_frames, cap_events, cap_chunks, pad1;

    // ---- MSKDemodulatorAFC carry (ref :337-347) ----
    double freq_offset;
    double mu;
    double timing_freq;
    double afc_alpha;
    // previous on-time sums P1..P4 (S_1 = (P1+P2, P3-P4), S_2 = (P1-P2, P3+P4)) in the kernel's
    // de-rotated form, stored in p1r,p1i,p2r,p2i in that order, and X[40] =
    // exp(j 40 d) of that symbol (the LO advance the reference's prev_corr implies; see
    // k_msk_frontend)
    double p1r, p1i, p2r, p2i;
    double x40c, x40s;
    double fo_sum;            // sum of the freq_offset used by every symbol so far (absolute LO phase, see k_frontend)
    double est_offset;        // NaN until estimate_offset ran
    uint32_t est_ties;        // offset-search candidates re-evaluated in the reference's order (near-ties)
    uint32_t est_nsym;        // 40-sample windows the search used (min(N, 40 000) / 40, ref :141)
    double energies[134];     // offset-search tap
    double est_poly[2 * OPV_OFFS_TERMS - 1];   // the search's energy polynomial in theta = 2 pi o / Fs (k_offset_search.hip), for the host's tie decision
    double est_power;         // sum |x|^2 over the samples the search used: scales its two near-tie bands (k_offset_search.hip)

    // ---- chunker carry (ref :1012-1076) ----
    uint64_t origin;          // sample index where the next demodulate() call starts
    uint64_t total_samples;   // sum of chunk sizes processed (ref :1027)
    uint64_t n_soft;          // symbols produced so far
    uint32_t n_chunks;
    int32_t first_chunk_done; // offset search done or skipped
    int32_t tail_done;        // EOF tail processed
    int32_t overflow;         // configuration error (soft ring too large for 32-bit byte offsets)
    int32_t stalled;          // back-pressure, cleared every round: bit 0 = the front-end skipped a demodulate() call
                              // because unread soft symbols fill the ring, bit 1 = the tracker stopped in front of a
                              // frame release because the ring of unpopped frames is full (resumes after opv_pop_frames)
    uint32_t edge_ties;       // symbols next to digital silence whose tone choice the reference decides by the rounding of
                              // its own LO (exactly one non-zero tap in the window; see k_frontend.hip::silence_pd)

    // ---- SyncTracker carry (ref :759-780), expressed on soft-log positions ----
    int32_t trk_state;        // OPV_HUNTING / VERIFYING / LOCKED
    int32_t trk_collecting;   // a payload is pending release at anchor+2144
    uint64_t trk_anchor;      // symbol at which symbols_since_sync_ was last reset
    uint64_t trk_next;        // next symbol index the tracker has not consumed yet
    double trk_quality;
    int32_t trk_misses;
    int32_t trk_sync_ok;      // whether the pending payload's sync word passed its check
    uint32_t n_frames;        // frames released (total_frames_)
    uint32_t n_events;
    uint32_t dec_from;        // first frame record k_frame_decode must handle this round

    // ---- diagnostics of the last front-end launch (opv_tap_wave_info) ----
    uint32_t dbg_hw_id, dbg_xcc_id;   // HW_REG_HW_ID / HW_REG_XCC_ID of the wave that served the stream
    uint64_t dbg_cycles, dbg_ticks;   // s_memtime (shader clock) and s_memrealtime (100 MHz) spent in the kernel
};

// ---- offset-search ties decided with the host's libm, in stream order (k_offset_search.hip: k_tie_collect / k_tie_apply,
// opv_offset_host.cpp: opv_offset_decide_slots, opv_capi.hip: opv_process) ----
// One slot per tied stream and pass, in PINNED host memory: k_tie_collect fills `stream` .. `iq`, a host function enqueued
// behind it (hipLaunchHostFunc: no HIP call inside) fills `est` .. `energies`, k_tie_apply behind that carries them into the
// stream's context before the front-end reads the estimate. Nothing waits on the host's side.
#define OPV_TIE_SLOTS_MAX 512                 // slots per pass (82 MB of pinned memory at most; a context of S streams has min(S, 512))
#define OPV_TIE_PASSES_MAX 8                  // passes per round: 4096 streams whose search ties at the last-place level in ONE round of ONE context;
                                              // what lies beyond (counted) keeps the device's decision - see opv_process
struct OpvTieSlot {
    uint32_t stream;          // index of the stream in its context
    uint32_t nsym;            // 40-sample windows the search used (<= 1000)
    double poly[2 * OPV_OFFS_TERMS - 1];   // OpvStream.est_poly
    double power;             // OpvStream.est_power
    double est;               // host -> device: the estimate in Hz
    uint32_t ties;            // host -> device: candidates re-evaluated (0: leave the stream as the device decided it)
    uint32_t pad;
    double energies[134];     // host -> device: the search's energies in scan order
    alignas(16) int16_t iq[2 * OPV_SPS * 1000];   // the first nsym windows of the capture (160 000 B)
};
struct OpvTieStage {
    uint32_t n;               // slots filled in this pass
    uint32_t listed;          // streams on the tie list in this round (all passes)
    uint32_t beyond;          // the round's LAST pass: listed streams that no pass staged (0 otherwise)
    uint32_t pad;
    OpvTieSlot slot[1];       // [slots]
};

struct OpvGlobalCfg {
    int32_t streaming;
    int32_t have_init_offset;
    double init_offset;
};
