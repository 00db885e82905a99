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
