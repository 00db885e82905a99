// opv_tx_internal.h — pieces of the host transmit chain (opv_tx.cpp) reused by the device
// modulator in opv_capi.hip / k_tx_modulate.hip. Not part of the public ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

// per-symbol tone/sign code of a whole opv-mod run: +/-1 tone 1, +/-2 tone 2, 0 silent
void opv_tx_symbol_codes(const uint8_t* frames134, size_t n_frames, int8_t* amp);
// NCO phases (ph1, ph2) at the start of n_symbols consecutive symbols, continuing from *ph1/*ph2
void opv_tx_symbol_phases(size_t first_symbol, size_t n_symbols, double* ph1_io, double* ph2_io, double* out2);
// one sample exactly as the reference computes it (libm), i samples into a symbol
void opv_tx_sample_exact(double ph1_sym, double ph2_sym, int a, int i, int16_t* I, int16_t* Q);

// ---- device transmit chain (k_tx_modulate.hip) ------------------------------------------------------------------
#define OPV_TX_CKPT_SYMS 128      // an NCO checkpoint every 128 symbols (5120 samples)
#define OPV_TX_CKPT_FRAMES 4096   // frames the build-time table covers (longer runs continue on the host, once per process)
// the embedded table (opv_tx_ckpt.cpp): n_entries pairs (ph1, ph2), entry j = state at symbol j * OPV_TX_CKPT_SYMS
const double* opv_tx_checkpoints(size_t* n_entries);
// entries [first, first + count) of the checkpoint sequence into out2 (2 doubles each): from the embedded table, beyond it
// from a process-wide extension that continues the recurrence on the host (thread-safe, computed once)
void opv_tx_checkpoint_range(size_t first, size_t count, double* out2);
