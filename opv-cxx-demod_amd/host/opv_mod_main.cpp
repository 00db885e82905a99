// opv_mod_main.cpp — host signal source with the CLI of the reference `opv-mod`
// (reference src/opv-mod.cpp:393-533): -S CALLSIGN -B N (BERT), -R (134-byte frames on
// stdin), -t TOKEN; int16 I/Q on stdout. Thin wrapper over opv_tx_* (csrc/opv_tx.cpp),
// whose output is sha256-identical to the reference modulator. -G <n>: the whole chain on GPU n
// (csrc/k_tx_modulate.hip, opv_tx_modulate_device_to_host) - the same bytes.
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/opv_demod.h"

static int usage(const char* prog) {  // ref opv-mod.cpp:393-412
    fprintf(stderr, "Usage: %s [OPTIONS]\n\n  -B FRAMES     BERT mode: generate N test frames\n"
                    "  -R            Raw mode: read 134-byte frames from stdin\n"
                    "  -S CALLSIGN   Station callsign (required for BERT mode)\n"
                    "  -t TOKEN      24-bit token (default: 0xBBAADD)\n"
                    "  -G DEVICE     run the transmit chain on that GPU (same output)\n\n"
                    "Output: 16-bit I/Q samples (little-endian, interleaved) to stdout\n", prog);
    return 1;
}

int main(int argc, char** argv) {
    std::string call;
    int bert = 0;
    bool raw = false;
    uint32_t token = 0xBBAADD;
    int opt, gpu = -1;
    while ((opt = getopt(argc, argv, "S:B:t:G:Rcvh")) != -1) {
        switch (opt) {
            case 'S': call = optarg; break;
            case 'B': bert = atoi(optarg); break;
            case 't': token = (uint32_t)strtoul(optarg, nullptr, 0); break;
            case 'R': raw = true; break;
            case 'G': gpu = atoi(optarg); break;
            case 'c': case 'v': break;  // continuous/verbose: not needed by any caller of the hot path
            default: return usage(argv[0]);
        }
    }
    if ((raw && bert > 0) || (!raw && bert <= 0) || (!raw && call.empty())) return usage(argv[0]);

    std::vector<uint8_t> frames;
    if (raw) {
        uint8_t buf[OPV_FRAME_BYTES];
        for (;;) {
            size_t got = 0;
            while (got < OPV_FRAME_BYTES) {
                ssize_t r = read(STDIN_FILENO, buf + got, OPV_FRAME_BYTES - got);
                if (r <= 0) break;
                got += (size_t)r;
            }
            if (got < OPV_FRAME_BYTES) break;  // clean EOF or partial frame (ref :365-387)
            frames.insert(frames.end(), buf, buf + OPV_FRAME_BYTES);
        }
    } else {
        frames.resize((size_t)bert * OPV_FRAME_BYTES);
        for (int f = 0; f < bert; ++f) opv_tx_bert_frame(call.c_str(), token, (uint32_t)f, &frames[(size_t)f * OPV_FRAME_BYTES]);
    }
    const size_t nf = frames.size() / OPV_FRAME_BYTES;
    std::vector<int16_t> iq(2 * opv_tx_modulated_samples(nf));
    if (gpu >= 0) {
        opv_cfg cfg{};
        cfg.streaming = 1;
        cfg.afc_alpha = 0.001;
        cfg.device = gpu;
        cfg.max_samples = 1 << 16;                 // (the context is only the device handle of the transmit chain here)
        opv_ctx* ctx = nullptr;
        if (opv_create(&ctx, 1, &cfg) < 0 || opv_tx_modulate_device_to_host(ctx, frames.data(), nf, iq.data()) < 0) {
            fprintf(stderr, "opv-mod: %s\n", opv_last_error());
            return 2;
        }
        opv_destroy(ctx);
    } else {
        opv_tx_modulate(frames.data(), nf, iq.data());
    }
    const char* p = reinterpret_cast<const char*>(iq.data());
    size_t left = iq.size() * sizeof(int16_t);
    while (left) {
        ssize_t w = write(STDOUT_FILENO, p, left);
        if (w <= 0) return 1;
        p += w;
        left -= (size_t)w;
    }
    return 0;
}
