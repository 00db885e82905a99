/* oracle/opv_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, fp64 CPU restatement of the Opulent Voice MSK transmit and
 * receive chains of OpenResearchInstitute/opv-cxx-demod, written from the behavioural
 * spec in SURVEY.md §8a / Appendix A with the reference's arithmetic ORDER preserved so
 * that its outputs are bit-identical to the compiled reference (same libm, no FMA
 * contraction; see oracle/Makefile).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker. The product (opv-cxx-demod_amd/) never links,
 * loads or calls it.
 *
 * Parity status: PINNED. tests/test_oracle_golden.py checks it against fixtures produced
 * by the compiled reference itself (oracle/_ref, built from /root/reference/src by
 * oracle/Makefile; fixtures made by tests/golden/make_golden.py) — IQ sha256 of the
 * modulator, decoded frames, all soft symbols (bit-exact), sync events, per-chunk carry
 * state, offset-search energies, quantiser / deinterleaver / Viterbi taps — and, when
 * oracle/_ref/libopv_ref.so is present, live against the reference classes.
 */
#ifndef OPV_ORACLE_H
#define OPV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORO_SPS = 40,             /* samples per symbol            (src/opv-demod.cpp:39)  */
    ORO_SYNC_BITS = 24,       /*                               (:47)                   */
    ORO_FRAME_BYTES = 134,    /*                               (:49)                   */
    ORO_FRAME_BITS = 1072,    /*                               (:50)                   */
    ORO_CODED_BITS = 2144,    /*                               (:51)                   */
    ORO_FRAME_SYMBOLS = 2168, /*                               (:52)                   */
    ORO_CHUNK_SAMPLES = 86720 /* streaming chunk               (:1012)                 */
};

/* ------------------------------ transmit chain (src/opv-mod.cpp) ------------------- */
void oro_base40_encode(const char* callsign, uint8_t out6[6]);               /* :59-91   */
void oro_bert_frame(const char* callsign, uint32_t token, uint32_t frame_num,
                    uint8_t out[ORO_FRAME_BYTES]);                            /* :339-361 */
void oro_lfsr_table(uint8_t out[ORO_FRAME_BYTES]);                            /* :97-113  */
void oro_encode_frame(const uint8_t payload[ORO_FRAME_BYTES],
                      uint8_t coded[ORO_CODED_BITS]);                         /* :159-213 */

typedef struct {
    double ph1, ph2; /* NCO phases, free-running                      (:287-288) */
    int t;           /* d_val_xor_T in {0,+1,-1}                      (:289)     */
    int bn;          /* b_n                                           (:290)     */
} oro_mod;
void oro_mod_reset(oro_mod* m);                                               /* :221-226 */
void oro_mod_symbol(oro_mod* m, int bit, int16_t iq[2 * ORO_SPS]);            /* :228-284 */
/* Whole run of opv-mod (-B or -R): one reset, sync+payload per frame, 100 zero symbols. */
size_t oro_modulated_len(size_t nframes); /* samples */
size_t oro_modulate_frames(const uint8_t* frames, size_t nframes, int16_t* iq); /* :473-529 */

/* ------------------------------ receive chain (src/opv-demod.cpp) ------------------ */
typedef struct {
    double freq_offset;        /* :337 */
    double phase_f1, phase_f2; /* :338 */
    double prev1_re, prev1_im, prev2_re, prev2_im; /* :339 */
    double afc_alpha;          /* :340 */
    double mu;                 /* :343 */
    double timing_freq;        /* :344 */
    double alpha_timing, beta_timing; /* :345-346 */
    size_t leftover;           /* :347 */
} oro_demod;
void   oro_demod_init(oro_demod* d);                                          /* :110-119 */
/* energies[134] (optional): 121 coarse then 13 fine candidate energies, in scan order */
double oro_estimate_offset(const int16_t* iq, size_t n, double* energies);   /* :131-202 */
size_t oro_demodulate(oro_demod* d, const int16_t* iq, size_t n,
                      double* soft, size_t cap);                              /* :206-329 */

/* Coherent (Costas-loop) demodulator, `opv-demod -c`, batch mode only (ref:365-572). */
typedef struct {
    double freq_offset;        /* :562 */
    double carrier_phase;      /* :563 */
    double phase_f1, phase_f2; /* :564 */
    double loop_freq;          /* :565  rad/sample */
    double prev_re, prev_im;   /* :566  prev_dominant_ */
    double afc_alpha;          /* :567 */
    double pll_alpha, pll_beta;/* :568-569 */
} oro_coh;
void   oro_coh_init(oro_coh* d);                                              /* :367-376 */
void   oro_coh_set_pll_bandwidth(oro_coh* d, double bw_hz);                   /* :551-558 */
/* extra[3*k..] (optional, cap_extra symbols): carrier_phase, loop_freq, freq_offset AFTER symbol k */
size_t oro_coh_demodulate(oro_coh* d, const int16_t* iq, size_t n, double* soft, size_t cap,
                          double* extra, size_t cap_extra);                   /* :455-543 */

enum { ORO_HUNTING = 0, ORO_VERIFYING = 1, ORO_LOCKED = 2 };                  /* :73      */
enum { /* event kinds, one per stderr line of SyncTracker::process */
    ORO_EV_HUNT_TO_VERIFY = 1, /* :651 */
    ORO_EV_VERIFY_TO_LOCK = 2, /* :677 */
    ORO_EV_SYNC_OK = 3,        /* :695 */
    ORO_EV_SYNC_MISS = 4,      /* :699 */
    ORO_EV_LOST_LOCK = 5       /* :705 */
};
typedef struct {
    int32_t kind;
    int32_t count;    /* frame number (VERIFY_TO_LOCK) or miss number (SYNC_MISS) */
    uint64_t sym_idx;
    double corr, raw;
} oro_event;

typedef struct {
    int state;
    double ring[ORO_SYNC_BITS];
    size_t ring_idx;
    size_t total_symbols;
    double pattern[ORO_SYNC_BITS];
    int collecting;
    double pending[ORO_CODED_BITS];
    size_t pending_n;
    size_t since_sync;
    double quality;
    int misses;
    int total_frames;
} oro_tracker;
void oro_tracker_init(oro_tracker* t);                                        /* :591-607 */
/* returns 1 if a frame was released (payload[2144], *quality filled). Events appended. */
int  oro_tracker_process(oro_tracker* t, double soft, size_t sym_idx, double* payload,
                         double* quality, oro_event* ev, size_t* n_ev, size_t cap_ev); /* :615-736 */

size_t oro_deinterleave_addr(size_t i);                                       /* :792-795 */
int    oro_viterbi(const int* in2144, uint8_t* bits1072);                     /* :800-847 */
/* optional taps: q[2144] quantised, deint[2144], bits[1072]. returns metric or -1 */
int    oro_frame_decode(const double* soft2144, uint8_t out[ORO_FRAME_BYTES],
                        int* q, int* deint, uint8_t* bits);                   /* :854-898 */

/* Whole receiver as main() drives it (streaming :995-1125, batch :1132-1216). */
typedef struct {
    int streaming;        /* -s */
    int have_init_offset; /* -o */
    double init_offset;
    double afc_alpha;     /* -a, default 0.001 */
    int coherent;         /* -c: honoured in batch mode only (ref:1144; the -s path never looks at it) */
    double pll_bw;        /* -p, default 50.0 (ref:946) */
} oro_rx_cfg;

typedef struct {
    /* caller-provided capacities / buffers (any may be NULL with cap 0) */
    uint8_t* frames; int32_t* metrics; double* quality; uint64_t* frame_sym; size_t cap_frames;
    double* soft; size_t cap_soft;
    oro_event* events; size_t cap_events;
    double* chunk_state; size_t cap_chunks; /* per demodulate call: afc, timing_freq, mu, leftover, nsoft */
    /* results */
    size_t n_frames;   /* frames with metric >= 0 (what -r writes) */
    size_t n_perfect;
    size_t n_soft;     /* total symbols */
    size_t n_events;
    size_t n_chunks;
    double est_offset; /* NaN if estimate_offset was not run */
    double final_freq_offset;
    double final_timing_freq;
    int final_state;
} oro_rx_out;

int oro_receive(const int16_t* iq, size_t n_samples, const oro_rx_cfg* cfg, oro_rx_out* out);

#ifdef __cplusplus
}
#endif
#endif
