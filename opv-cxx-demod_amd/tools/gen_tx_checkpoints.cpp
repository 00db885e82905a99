// gen_tx_checkpoints.cpp — build-time tool: the state of the modulator's two free-running NCOs at every
// OPV_TX_CKPT_SYMS-th symbol of an `opv-mod` run, for the first <n_frames> frames.
//
// The NCO phases (reference src/opv-mod.cpp:274-279: ph += inc, wrapped into (-pi, pi] by while-loops, every sample, for
// BOTH tones whatever is keyed) do not depend on the data, but each value is the ROUNDED sum of its predecessor: the
// sequence can only be produced in order. The device modulator therefore starts from tabulated states: this tool replays
// the reference's additions once, at build time (0.1 s per 1000 frames), and writes (ph1, ph2) as raw doubles; the
// library embeds the file (csrc/opv_tx_ckpt.cpp) and the device replays the 128 x 40 additions between two entries in
// parallel (k_tx_expand_phases). Same arithmetic as opv_tx.cpp::opv_tx_symbol_phases (built with -ffp-contract=off).
#include <cstdio>
#include <cstdlib>

#include "../csrc/opv_tx_internal.h"

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s <n_frames> <out.bin>\n", argv[0]); return 2; }
    const size_t n_frames = std::strtoull(argv[1], nullptr, 10);
    const size_t nsym = n_frames * 2168u;
    const size_t n_ck = (nsym + OPV_TX_CKPT_SYMS - 1) / OPV_TX_CKPT_SYMS;
    std::FILE* f = std::fopen(argv[2], "wb");
    if (!f) { std::perror(argv[2]); return 1; }
    double ph1 = 0.0, ph2 = 0.0, tmp[2 * OPV_TX_CKPT_SYMS];
    for (size_t j = 0; j <= n_ck; ++j) {             // n_ck + 1 entries: the last one is where a longer run continues
        const double ck[2] = {ph1, ph2};
        if (std::fwrite(ck, sizeof ck, 1, f) != 1) { std::perror("fwrite"); return 1; }
        opv_tx_symbol_phases(j * OPV_TX_CKPT_SYMS, OPV_TX_CKPT_SYMS, &ph1, &ph2, tmp);
    }
    std::fclose(f);
    return 0;
}
