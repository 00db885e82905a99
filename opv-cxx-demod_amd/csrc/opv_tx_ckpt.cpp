// opv_tx_ckpt.cpp — embeds build/opv_tx_checkpoints.bin (made by tools/gen_tx_checkpoints at build time): the state
// (ph1, ph2) of the modulator's free-running NCOs at every OPV_TX_CKPT_SYMS-th symbol of a run, for the first
// OPV_TX_CKPT_FRAMES frames. See tools/gen_tx_checkpoints.cpp for why this is a table.
#include "opv_tx_internal.h"

__asm__(".section .rodata\n"
        ".balign 16\n"
        ".global opv_tx_ckpt_begin\n"
        "opv_tx_ckpt_begin:\n"
        ".incbin \"build/opv_tx_checkpoints.bin\"\n"
        ".global opv_tx_ckpt_end\n"
        "opv_tx_ckpt_end:\n"
        ".previous\n");

extern "C" const double opv_tx_ckpt_begin[], opv_tx_ckpt_end[];

const double* opv_tx_checkpoints(size_t* n_entries) {
    *n_entries = (size_t)(opv_tx_ckpt_end - opv_tx_ckpt_begin) / 2;
    return opv_tx_ckpt_begin;
}
