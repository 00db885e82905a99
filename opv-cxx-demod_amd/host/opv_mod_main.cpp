// opv_mod_main.cpp — host signal source with the CLI of the reference `opv-mod`
// (reference src/opv-mod.cpp:393-533): -S CALLSIGN -B N (BERT), -R (134-byte frames on stdin), -t TOKEN, -c (loop the BERT
// pass forever), -v (progress on stderr); int16 I/Q on stdout. Thin wrapper over opv_tx_* (csrc/opv_tx.cpp), whose output is
// sha256-identical to the reference modulator.
// Like the reference it WRITES AS IT GOES: raw mode modulates the frames that have arrived (one at a time from a live
// source, up to 64 at once from a file) and BERT mode works in blocks of 64 frames, so memory is bounded, `-c` can run
// forever and a reader sees the first samples after the first frame. Not the reference's structure: frames are modulated
// in blocks by opv_tx_stream_frames (frame-parallel on host threads), not sample by sample through an ostream.
// -G <n>: the whole chain on GPU n (csrc/k_tx_modulate.hip, opv_tx_modulate_device_to_host) - the same bytes; one finite
// run at a time (all frames first, then one device pass), so not with -c.
#include <poll.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/opv_demod.h"

namespace {

constexpr size_t kBlock = 64;                                      // frames per modulator call (22 MB of samples)
constexpr size_t kFrameSamples = (size_t)OPV_FRAME_SYMBOLS * OPV_SAMPLES_PER_SYMBOL;

[[noreturn]] void usage(const char* prog) {  // ref opv-mod.cpp:393-412 (+ the one flag that is ours)
    fprintf(stderr, "Usage: %s [OPTIONS]\n\n", prog);
    fprintf(stderr, "Modes (mutually exclusive):\n");
    fprintf(stderr, "  -B FRAMES     BERT mode: generate N test frames\n");
    fprintf(stderr, "  -R            Raw mode: read 134-byte frames from stdin\n");
    fprintf(stderr, "\n");
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "  -S CALLSIGN   Station callsign (required for BERT mode)\n");
    fprintf(stderr, "  -t TOKEN      24-bit token (default: 0xBBAADD)\n");
    fprintf(stderr, "  -c            Continuous mode (loop BERT forever)\n");
    fprintf(stderr, "  -v            Verbose output to stderr\n");
    fprintf(stderr, "  -G DEVICE     run the transmit chain on that GPU (same output; one finite run)\n");
    fprintf(stderr, "\n");
    fprintf(stderr, "Output: 16-bit I/Q samples (little-endian, interleaved) to stdout\n");
    fprintf(stderr, "\n");
    fprintf(stderr, "Examples:\n");
    fprintf(stderr, "  %s -S W5NYV -B 10              # 10 BERT frames\n", prog);
    fprintf(stderr, "  %s -R < frames.bin             # Modulate pre-built frames\n", prog);
    fprintf(stderr, "  cat frames.bin | %s -R         # Same, via pipe\n", prog);
    exit(1);
}

bool write_all(const void* data, size_t bytes) {
    const char* p = static_cast<const char*>(data);
    while (bytes) {
        const ssize_t w = write(STDOUT_FILENO, p, bytes);
        if (w <= 0) return false;
        p += w;
        bytes -= (size_t)w;
    }
    return true;
}

// one whole frame from stdin; false on EOF, on a partial frame and on a read error, with the reference's messages (ref :365-387)
bool read_frame(uint8_t* out) {
    size_t got = 0;
    while (got < OPV_FRAME_BYTES) {
        const ssize_t r = read(STDIN_FILENO, out + got, OPV_FRAME_BYTES - got);
        if (r <= 0) {
            if (r == 0 && got != 0) fprintf(stderr, "Warning: EOF after partial frame (%zu bytes)\n", got);
            else if (r < 0) fprintf(stderr, "Error reading from stdin\n");
            return false;
        }
        got += (size_t)r;
    }
    return true;
}

// what the reference's encode_frame / send_encoded_frame print per frame under -v (ref :171-183, :198-209, :326-329)
void print_frame_debug(const uint8_t* frame) {
    uint8_t rnd[OPV_FRAME_BYTES], lin[OPV_ENCODED_BITS], il[OPV_ENCODED_BITS];
    opv_tap_tx_frame(frame, rnd, lin, il);
    char line[160];
    int n = snprintf(line, sizeof line, "Payload[0:11]: ");
    for (int i = 0; i < 12; ++i) n += snprintf(line + n, sizeof line - n, "%02x ", frame[i]);
    fprintf(stderr, "%s\n", line);
    n = snprintf(line, sizeof line, "Randomized[0:5]: ");
    for (int i = 0; i < 6; ++i) n += snprintf(line + n, sizeof line - n, "%02x ", rnd[i]);
    fprintf(stderr, "%s\n", line);
    auto bits = [&](const char* head, const uint8_t* b) {
        int m = snprintf(line, sizeof line, "%s", head);
        for (int i = 0; i < 32; ++i) line[m++] = (char)('0' + b[i]);
        line[m] = 0;
        fprintf(stderr, "%s\n", line);
    };
    bits("Before interleave [0:31]: ", lin);
    bits("After interleave [0:31]:  ", il);
    bits("Encoded bits [0:31]: ", il);
}

bool stdin_has_data() {
    pollfd p{STDIN_FILENO, POLLIN, 0};
    return poll(&p, 1, 0) > 0 && (p.revents & POLLIN);
}

}  // namespace

int main(int argc, char** argv) {
    std::string call;
    int bert = 0;
    bool raw = false, continuous = false, verbose = false;
    uint32_t token = 0xBBAADD;
    int opt, gpu = -1;
    while ((opt = getopt(argc, argv, "S:B:t:G:Rcvh")) != -1) {
        switch (opt) {
            case 'S': call = optarg; break;
            case 'B': bert = atoi(optarg); break;
            case 't': token = (uint32_t)strtoul(optarg, nullptr, 0); break;
            case 'R': raw = true; break;
            case 'c': continuous = true; break;
            case 'v': verbose = true; break;
            case 'G': gpu = atoi(optarg); break;
            default: usage(argv[0]);
        }
    }
    if (raw && bert > 0) { fprintf(stderr, "Error: -R and -B are mutually exclusive\n"); usage(argv[0]); }              // ref :432-435
    if (!raw && bert <= 0) { fprintf(stderr, "Error: Must specify either -R (raw mode) or -B N (BERT mode)\n"); usage(argv[0]); }
    if (!raw && call.empty()) { fprintf(stderr, "Error: BERT mode requires -S CALLSIGN\n"); usage(argv[0]); }
    if (!call.empty() && call.length() > 9) {                                                                             // ref :451-454
        fprintf(stderr, "Warning: Callsign truncated to 9 characters for Base-40 encoding\n");
        call = call.substr(0, 9);
    }
    if (gpu >= 0 && continuous && !raw) { fprintf(stderr, "Error: -G modulates one finite run; not with -c\n"); usage(argv[0]); }
    if (verbose) {                                                                                                        // ref :456-469
        fprintf(stderr, "OPV Modulator\n");
        if (raw) {
            fprintf(stderr, "  Mode: Raw (reading 134-byte frames from stdin)\n");
        } else {
            fprintf(stderr, "  Mode: BERT\n");
            fprintf(stderr, "  Callsign: %s\n", call.c_str());
            fprintf(stderr, "  Token:    0x%x\n", token);
            fprintf(stderr, "  Frames:   %d\n", bert);
        }
        fprintf(stderr, "  Conv encoder: G1=0x4F, G2=0x6D\n");
        fprintf(stderr, "\n");
    }

    std::vector<uint8_t> frames;
    std::vector<int16_t> iq;

    if (gpu >= 0) {  // one finite run through the device chain
        if (raw) {
            uint8_t f[OPV_FRAME_BYTES];
            while (read_frame(f)) frames.insert(frames.end(), f, f + OPV_FRAME_BYTES);
        } else {
            frames.resize((size_t)bert * OPV_FRAME_BYTES);
            opv_tx_bert_frames(call.c_str(), token, 0, (size_t)bert, frames.data());
        }
        const size_t nf = frames.size() / OPV_FRAME_BYTES;
        iq.resize(2 * opv_tx_modulated_samples(nf));
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
        if (verbose) {                                // (the reference's progress lines, after the fact: the device pass is one step)
            for (size_t f = 0; f < nf; ++f) {
                print_frame_debug(&frames[f * OPV_FRAME_BYTES]);
                if (raw) { if ((f + 1) % 100 == 0) fprintf(stderr, "Sent %zu frames\n", f + 1); }
                else if ((f + 1) % 10 == 0 || f + 1 == nf) fprintf(stderr, "Sent frame %zu/%zu\n", f + 1, nf);
            }
            if (raw) fprintf(stderr, "End of input. Total frames: %zu\n", nf);
        }
        if (!write_all(iq.data(), iq.size() * sizeof(int16_t))) return 1;
        if (verbose) fprintf(stderr, "Done.\n");
        return 0;
    }

    opv_tx_stream* mod = opv_tx_stream_create();   // = g_mod after reset() (ref :476 / :506)
    if (!mod) { fprintf(stderr, "opv-mod: out of memory\n"); return 2; }
    frames.resize(kBlock * OPV_FRAME_BYTES);
    iq.resize(2 * kBlock * kFrameSamples);

    if (raw) {                                                                                                            // ref :473-498
        unsigned long long count = 0;
        for (;;) {
            size_t n = 0;
            // the frames that are there: at least one (blocking), more only while stdin has them ready
            while (n < kBlock && (n == 0 || stdin_has_data())) {
                if (!read_frame(&frames[n * OPV_FRAME_BYTES])) { if (n == 0) goto raw_done; break; }
                ++n;
            }
            const size_t ns = opv_tx_stream_frames(mod, frames.data(), n, iq.data());
            if (!write_all(iq.data(), ns * 2 * sizeof(int16_t))) return 1;
            for (size_t k = 0; k < n; ++k) {
                if (verbose) print_frame_debug(&frames[k * OPV_FRAME_BYTES]);
                if (++count % 100 == 0 && verbose) fprintf(stderr, "Sent %llu frames\n", count);
            }
        }
    raw_done:
        if (verbose) fprintf(stderr, "End of input. Total frames: %llu\n", count);
    } else {                                                                                                              // ref :503-524
        uint32_t frame_num = 0;
        do {
            opv_tx_stream_reset(mod);
            for (int f = 0; f < bert;) {
                const size_t n = (size_t)(bert - f) < kBlock ? (size_t)(bert - f) : kBlock;
                opv_tx_bert_frames(call.c_str(), token, frame_num, n, frames.data());
                frame_num += (uint32_t)n;
                const size_t ns = opv_tx_stream_frames(mod, frames.data(), n, iq.data());
                if (!write_all(iq.data(), ns * 2 * sizeof(int16_t))) return 1;
                for (size_t k = 0; k < n; ++k, ++f) {
                    if (verbose) print_frame_debug(&frames[k * OPV_FRAME_BYTES]);
                    if (verbose && ((f + 1) % 10 == 0 || f == bert - 1)) fprintf(stderr, "Sent frame %d/%d\n", f + 1, bert);
                }
            }
        } while (continuous);
    }
    const size_t nt = opv_tx_stream_tail(iq.data());                                                                     // ref :527-529
    if (!write_all(iq.data(), nt * 2 * sizeof(int16_t))) return 1;
    opv_tx_stream_destroy(mod);
    if (verbose) fprintf(stderr, "Done.\n");
    return 0;
}
