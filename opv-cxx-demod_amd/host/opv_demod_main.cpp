// opv_demod_main.cpp — host program with the process contract of the reference `opv-demod`
// (reference src/opv-demod.cpp:943-1217): int16 I/Q on stdin, 134-byte frames on stdout
// (-r), human text on stderr, flags -q -r -s -c -a -o -p -h (-c/-p: the batch Costas-loop demodulator,
// csrc/k_coherent.hip - prefix parity only), exit status 0 iff at least one frame decoded. All arithmetic runs on the MI355X through
// the C ABI in include/opv_demod.h; this file only moves bytes and prints.
#include <poll.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/opv_demod.h"

namespace {

constexpr double kFs = 2168000.0;

// Base-40 station id (ref :87-103): first character least significant
std::string base40(const uint8_t* b) {
    uint64_t v = 0;
    for (int i = 0; i < 6; ++i) v = (v << 8) | b[i];
    if (v == 0) return "(empty)";
    std::string s;
    while (v > 0) {
        const int d = (int)(v % 40);
        v /= 40;
        char c = 0;
        if (d >= 1 && d <= 26) c = (char)('A' + d - 1);
        else if (d >= 27 && d <= 36) c = (char)('0' + d - 27);
        else if (d == 37) c = '-';
        else if (d == 38) c = '/';
        else if (d == 39) c = '.';
        if (c) s += c;
    }
    return s.empty() ? "(empty)" : s;
}

const char* state_name(int s) {  // ref :75-82
    return s == OPV_HUNTING ? "HUNTING" : s == OPV_VERIFYING ? "VERIFYING" : s == OPV_LOCKED ? "LOCKED" : "?";
}

void print_event(FILE* err, const opv_event& e) {  // ref :651,677,695,699,705
    const unsigned long long i = e.sym_idx;
    switch (e.kind) {
        case OPV_EV_HUNT_TO_VERIFY: fprintf(err, "[%llu] HUNTING→VERIFYING (corr=%.3f, raw=%.0f)\n", i, e.corr, e.raw); break;
        case OPV_EV_VERIFY_TO_LOCK: fprintf(err, "[%llu] VERIFYING→LOCKED (frame %d)\n", i, e.count); break;
        case OPV_EV_SYNC_OK: fprintf(err, "[%llu] LOCKED: sync OK (corr=%.3f)\n", i, e.corr); break;
        case OPV_EV_SYNC_MISS: fprintf(err, "[%llu] LOCKED: sync MISS #%d (corr=%.3f)\n", i, e.count, e.corr); break;
        case OPV_EV_LOST_LOCK: fprintf(err, "[%llu] LOCKED→HUNTING (lost lock)\n", i); break;
        default: break;
    }
}

// The per-frame box of the reference (ref :907-938) is part of the process contract (stderr text
// byte for byte), so its literals are fixed; it is assembled here in one buffer and handed to stderr with a
// single write, a row at a time from a small table of field formatters.
namespace box {
constexpr const char* kTop = "┌─────────────────────────────────────────────────────────────────┐\n";
constexpr const char* kSep = "├─────────────────────────────────────────────────────────────────┤\n";
constexpr const char* kBot = "└─────────────────────────────────────────────────────────────────┘\n\n";

inline uint32_t be24(const uint8_t* p) { return ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2]; }

struct Text {
    std::string s;
    void add(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
        char tmp[160];
        va_list ap;
        va_start(ap, fmt);
        const int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
        va_end(ap);
        if (n > 0) s.append(tmp, (size_t)std::min<int>(n, (int)sizeof tmp - 1));
    }
};

// one 16-byte row of the dump: offset, hex column padded to 16 places, printable column
void dump_row(Text& t, const uint8_t* f, size_t at) {
    const size_t n = std::min<size_t>(16, OPV_FRAME_BYTES - at);
    std::string hex, asc;
    char h[4];
    for (size_t k = 0; k < 16; ++k) {
        if (k < n) { snprintf(h, sizeof h, "%02X ", f[at + k]); hex += h; asc += (f[at + k] >= 0x20 && f[at + k] < 0x7F) ? (char)f[at + k] : '.'; }
        else hex += "   ";
    }
    t.add("│ %02zx: %s │%s│\n", at, hex.c_str(), asc.c_str());
}
}  // namespace box

void print_frame(FILE* err, int num, const uint8_t* f, int metric, double sync) {
    box::Text t;
    t.s.reserve(2048);
    t.s += box::kTop;
    t.add("│ FRAME %4d  │  Sync: %.3f  │  Metric: %5d%s\n", num, sync, metric, metric == 0 ? " (perfect)" : "");
    t.s += box::kSep;
    t.add("│ Station ID:  %-12s (Base-40)\n", base40(f).c_str());
    const uint32_t token = box::be24(f + 6);
    t.add("│ Token:       0x%06X%s\n", token, token == 0xBBAADD ? " (default)" : "");
    t.add("│ Reserved:    0x%06X\n", box::be24(f + 9));
    t.s += box::kSep;
    t.s += "│ Hex Dump:                                                       │\n";
    for (size_t at = 0; at < OPV_FRAME_BYTES; at += 16) box::dump_row(t, f, at);
    t.s += box::kBot;
    fwrite(t.s.data(), 1, t.s.size(), err);
}

struct Options {
    bool quiet = false, raw = false, coherent = false, streaming = false, have_off = false;
    double afc = 0.001, off = 0.0, capacity_sec = 2.0, pll_bw = 50.0;  // ref :946
    int device = 0;
};

struct Sink {
    opv_ctx* ctx;
    Options o;
    int decoded = 0, perfect = 0;
    size_t chunks_seen = 0;
    uint64_t total_samples = 0, total_symbols = 0;
    uint64_t full_samples = 0, full_symbols = 0;  // what the reference's Total: line counts (full chunks only, ref :1027,:1067)
    bool est_printed = false;
    FILE* err = stderr;  // where tracker lines and frame boxes go (batch mode holds them back while a stalled stream finishes)

    void write_frame(const uint8_t* f) {
        // one write(2) per frame, like the 134-byte fully-buffered stdout of the reference (ref :978-979)
        size_t off = 0;
        while (off < OPV_FRAME_BYTES) {
            ssize_t w = ::write(STDOUT_FILENO, f + off, OPV_FRAME_BYTES - off);
            if (w <= 0) return;
            off += (size_t)w;
        }
    }

    // Print everything the device produced since the last call, in the order the reference
    // would have printed it: per chunk, tracker lines and frames by symbol index, then the
    // 5-second status line (ref :1045-1083).
    int drain() {
        opv_stream_state st;
        if (opv_get_state(ctx, 0, &st) < 0) return -1;
        if (!o.quiet && o.streaming && !est_printed && !std::isnan(st.est_offset_hz)) {
            fprintf(stderr, "Estimated carrier offset: %.1f Hz\n\n", st.est_offset_hz);  // ref :1035
            est_printed = true;
        }
        std::vector<opv_event> ev(4096);
        std::vector<uint8_t> fr(256 * OPV_FRAME_BYTES);
        std::vector<opv_frame_meta> meta(256);
        std::vector<opv_event> all_ev;
        std::vector<uint8_t> all_fr;
        std::vector<opv_frame_meta> all_meta;
        for (;;) {
            long n = opv_pop_events(ctx, 0, ev.data(), ev.size());
            if (n < 0) return -1;
            all_ev.insert(all_ev.end(), ev.begin(), ev.begin() + n);
            if ((size_t)n < ev.size()) break;
        }
        for (;;) {
            long n = opv_pop_frames(ctx, 0, fr.data(), 256, meta.data());
            if (n < 0) return -1;
            all_fr.insert(all_fr.end(), fr.begin(), fr.begin() + n * OPV_FRAME_BYTES);
            all_meta.insert(all_meta.end(), meta.begin(), meta.begin() + n);
            if (n < 256) break;
        }
        const size_t n_new = (size_t)st.n_chunks - chunks_seen;  // chunk log entries not printed yet
        std::vector<double> cl(n_new * 5 + 5);
        if (n_new && opv_tap_chunks(ctx, 0, (uint32_t)chunks_seen, cl.data(), n_new) < 0) return -1;
        const size_t chunk0 = chunks_seen;

        size_t ie = 0, ifr = 0;
        auto emit_until = [&](uint64_t sym_end) {  // everything with symbol index < sym_end
            for (;;) {
                const bool he = ie < all_ev.size() && all_ev[ie].sym_idx < sym_end;
                const bool hf = ifr < all_meta.size() && all_meta[ifr].release_symbol < sym_end;
                if (!he && !hf) break;
                if (he && (!hf || all_ev[ie].sym_idx <= all_meta[ifr].release_symbol)) {
                    print_event(err, all_ev[ie++]);  // tracker lines are printed even under -q (ref :651)
                } else {
                    const opv_frame_meta& m = all_meta[ifr];
                    const uint8_t* f = all_fr.data() + ifr * OPV_FRAME_BYTES;
                    ++decoded;
                    if (m.viterbi_metric == 0) ++perfect;
                    if (!o.quiet) print_frame(err, decoded, f, m.viterbi_metric, m.sync_quality);
                    if (o.raw) write_frame(f);
                    ++ifr;
                }
            }
        };
        for (; chunks_seen < (size_t)st.n_chunks; ++chunks_seen) {
            const double* c = &cl[(chunks_seen - chunk0) * 5];
            const uint64_t nsym = (uint64_t)c[4];
            // chunk size = what demodulate() was given: full chunks are 86720, the tail is the rest
            emit_until(total_symbols + nsym);
            total_symbols += nsym;
            if (o.streaming) {
                const bool tail = st.flushed && chunks_seen + 1 == (size_t)st.n_chunks && (uint64_t)st.total_samples - total_samples < OPV_CHUNK_SAMPLES;
                const uint64_t csz = tail ? (uint64_t)st.total_samples - total_samples : OPV_CHUNK_SAMPLES;
                total_samples += csz;
                if (!tail) { full_samples += csz; full_symbols += nsym; }
                if (!tail && !o.quiet && (total_samples % (uint64_t)(kFs * 5) < OPV_CHUNK_SAMPLES))  // ref :1079-1083
                    fprintf(stderr, "[%.1fs] %llu symbols, %d frames (%d perfect), AFC: %.1f Hz, TFreq: %.4f\n",
                            total_samples / kFs, (unsigned long long)total_symbols, decoded, perfect, c[0], c[1]);
            }
        }
        emit_until(~0ull);
        return 0;
    }
};

int die(const char* what) {
    fprintf(stderr, "opv-demod: %s: %s\n", what, opv_last_error());
    return 2;
}

// After the last opv_process: a stream held back by back-pressure (opv_stream_state.stalled - unpopped frames or
// unread soft symbols fill a ring) has not finished. Keep draining and processing until it has; a round that
// neither pops anything nor advances the stream is an error, never a silent truncation.
int finish(opv_ctx* ctx, Sink& sink) {
    for (;;) {
        opv_stream_state st;
        if (opv_get_state(ctx, 0, &st) < 0) return -1;
        if (!st.stalled) return 0;
        const uint64_t sym0 = st.total_symbols;
        const int rel0 = st.frames_released, dec0 = sink.decoded;
        if (opv_process(ctx) < 0 || sink.drain() < 0) return -1;
        if (opv_get_state(ctx, 0, &st) < 0) return -1;
        if (st.stalled && st.total_symbols == sym0 && st.frames_released == rel0 && sink.decoded == dec0) {
            fprintf(stderr, "opv-demod: stream stalled (0x%x) without progress: device rings too small for this input\n", st.stalled);
            return -2;
        }
    }
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {  // same hand-rolled loop as the reference (ref :950-974): unknown flags ignored
        if (!strcmp(argv[i], "-q")) o.quiet = true;
        else if (!strcmp(argv[i], "-r")) o.raw = true;
        else if (!strcmp(argv[i], "-c")) o.coherent = true;
        else if (!strcmp(argv[i], "-s")) o.streaming = true;
        else if (!strcmp(argv[i], "-a") && i + 1 < argc) o.afc = atof(argv[++i]);
        else if (!strcmp(argv[i], "-p") && i + 1 < argc) o.pll_bw = atof(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) { o.off = atof(argv[++i]); o.have_off = true; }
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) o.device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--capacity-sec") && i + 1 < argc) o.capacity_sec = atof(argv[++i]);
        else if (!strcmp(argv[i], "-h")) {
            fprintf(stderr, "Usage: %s [options] < input.iq\n\n", argv[0]);
            fprintf(stderr, "Options:\n");
            fprintf(stderr, "  -q          Quiet mode\n");
            fprintf(stderr, "  -r          Raw output to stdout\n");
            fprintf(stderr, "  -s          Streaming mode (for live PlutoSDR input)\n");
            fprintf(stderr, "  -c          Coherent mode (Costas loop, ~3dB better)\n");
            fprintf(stderr, "  -a <bw>     AFC bandwidth (default: 0.001)\n");
            fprintf(stderr, "  -o <hz>     Initial frequency offset (streaming mode)\n");
            fprintf(stderr, "  -p <hz>     PLL bandwidth in Hz (default: 50, coherent only)\n");
            fprintf(stderr, "  -h          Help\n");
            // the reference's text ends here (ref :962-971, byte for byte above); two flags it does not have follow
            fprintf(stderr, "  --device <n>        HIP device ordinal (default 0)\n");
            fprintf(stderr, "  --capacity-sec <s>  device staging buffer in seconds of IQ (default 2; streams may be any length)\n");
            return 0;
        }
    }
    // -c: the reference honours it in batch mode only (:1144); with -s it merely changes the banner
    // (:983-984) and the non-coherent streaming path runs (:995-1125). The batch Costas loop runs on the GPU too
    // (csrc/k_coherent.hip); its trajectory is chaotic in the reference itself, so the output agrees with the
    // reference's for the first ~2000 symbols only.
    if (!o.quiet) {  // ref :981-990
        fprintf(stderr, "╔═══════════════════════════════════════════════════════════════════╗\n");
        if (o.coherent) fprintf(stderr, "║       OPV MSK Demodulator with Costas Loop v1.0 (coherent)       ║\n");
        else if (o.streaming) fprintf(stderr, "║       OPV MSK Demodulator with AFC v1.0 (streaming)              ║\n");
        else fprintf(stderr, "║           OPV MSK Demodulator with AFC v1.0                       ║\n");
        fprintf(stderr, "╚═══════════════════════════════════════════════════════════════════╝\n\n");
    }

    std::vector<int16_t> all;  // batch mode slurps (ref :1132-1135)
    opv_cfg cfg{};
    cfg.streaming = o.streaming;
    cfg.have_init_offset = o.have_off;
    cfg.init_offset_hz = o.off;
    cfg.afc_alpha = o.afc;
    cfg.device = o.device;
    cfg.coherent = o.coherent;
    cfg.pll_bw_hz = o.pll_bw;

    std::vector<char> buf(1 << 18);
    size_t carry = 0;  // bytes of a partial sample at the end of a read
    auto read_block = [&](std::vector<int16_t>& dst) -> long {  // returns samples appended, 0 on EOF
        for (;;) {
            ssize_t r = ::read(STDIN_FILENO, buf.data() + carry, buf.size() - carry);
            if (r < 0) return -1;
            if (r == 0) return 0;  // trailing partial sample is ignored (ref :1022)
            const size_t have = carry + (size_t)r, ns = have / 4;
            if (ns) {
                const int16_t* p = reinterpret_cast<const int16_t*>(buf.data());
                dst.insert(dst.end(), p, p + 2 * ns);
            }
            carry = have - ns * 4;
            if (carry) memmove(buf.data(), buf.data() + ns * 4, carry);
            if (ns) return (long)ns;
        }
    };

    opv_ctx* ctx = nullptr;
    if (o.streaming) {
        if (!o.quiet) fprintf(stderr, "Streaming mode: processing data as it arrives...\n\n");
        if (o.have_off && !o.quiet) fprintf(stderr, "Initial frequency offset: %.1f Hz\n", o.off);
        cfg.max_samples = (uint64_t)std::min(o.capacity_sec * kFs, 2147483000.0);
        if (opv_create(&ctx, 1, &cfg) < 0) return die("opv_create");
        Sink sink{ctx, o};
        std::vector<int16_t> blk;
        uint64_t pushed = 0, since = 0;
        // A round's results are delivered at once when the source is live (nothing waiting on stdin: latency counts), and
        // one read later when it is not (a file, a fast pipe): the next block is then read and copied to the device while the
        // kernels of this round run, instead of the host waiting for them first.
        bool pending = false;
        // what a round may take in before it is processed: eight chunks, or what the staging buffer holds besides the carry of
        // the last chunk, a chunk of slack and one more read (a small --capacity-sec: chunk by chunk as before)
        const uint64_t hold = 2ull * OPV_CHUNK_SAMPLES + buf.size() / 4;
        const uint64_t round_cap = cfg.max_samples > hold ? std::min<uint64_t>(8ull * OPV_CHUNK_SAMPLES, cfg.max_samples - hold) : 0;
        auto stdin_has_data = [] { pollfd p{STDIN_FILENO, POLLIN, 0}; return poll(&p, 1, 0) > 0 && (p.revents & (POLLIN | POLLHUP)); };
        for (;;) {
            blk.clear();
            long ns = read_block(blk);
            if (ns <= 0) break;
            if (opv_push_iq(ctx, 0, blk.data(), (size_t)ns) < 0) return die("opv_push_iq");
            if (pending) { if (sink.drain() < 0) return die("drain"); pending = false; }
            pushed += (uint64_t)ns;
            since += (uint64_t)ns;
            if (since >= OPV_CHUNK_SAMPLES - 64) {  // a chunk boundary may have been crossed
                // a backlog on stdin (a file, a fast pipe): up to eight chunks go into one round - the launches and the
                // hand-shake of a round are paid once; a live source never has a backlog and is served chunk by chunk
                if (since < round_cap && stdin_has_data()) continue;
                if (opv_process(ctx) < 0) return die("opv_process");
                if (stdin_has_data()) pending = true;
                else if (sink.drain() < 0) return die("drain");
                since = 0;
            }
        }
        if (pending && sink.drain() < 0) return die("drain");
        if (opv_flush(ctx, 0) < 0 || opv_process(ctx) < 0) return die("opv_flush");
        if (sink.drain() < 0) return die("drain");
        if (finish(ctx, sink) < 0) return die("finish");
        opv_stream_state st;
        opv_get_state(ctx, 0, &st);
        if (!o.quiet) {  // ref :1115-1122
            fprintf(stderr, "\n════════════════════════════════════════════════════════════════════\n");
            fprintf(stderr, "Summary: %d frames (%d perfect, %d errors)\n", sink.decoded, sink.perfect, sink.decoded - sink.perfect);
            fprintf(stderr, "Total: %.3f sec, %llu symbols\n", sink.full_samples / kFs, (unsigned long long)sink.full_symbols);
            fprintf(stderr, "Final state: %s, AFC: %.1f Hz\n", state_name(st.sync_state), st.freq_offset_hz);
            fprintf(stderr, "════════════════════════════════════════════════════════════════════\n");
        }
        opv_destroy(ctx);
        return sink.decoded > 0 ? 0 : 1;
    }

    // ---- batch mode (ref :1132-1216) ----
    for (;;) {
        long ns = read_block(all);
        if (ns <= 0) break;
    }
    const size_t n = all.size() / 2;
    if (!o.quiet) fprintf(stderr, "Loaded %zu samples (%.3f sec)\n", n, n / kFs);
    cfg.max_samples = n + 64;
    if (opv_create(&ctx, 1, &cfg) < 0) return die("opv_create");
    if (n && opv_push_iq(ctx, 0, all.data(), n) < 0) return die("opv_push_iq");
    if (opv_flush(ctx, 0) < 0 || opv_process(ctx) < 0) return die("opv_process");
    opv_stream_state st;
    if (opv_get_state(ctx, 0, &st) < 0) return die("opv_get_state");
    Sink sink{ctx, o};
    // The reference prints its symbol count and final AFC value BEFORE the frames (:1170-1175). The device rings are sized for
    // the whole capture (max_samples = n + 64), so the one opv_process above has normally finished the stream; should
    // back-pressure have held it back all the same, the stream is finished first - tracker lines and frame boxes collected
    // in memory - so that the line reports the final figures, then they follow it.
    char* held = nullptr;
    size_t held_n = 0;
    if (st.stalled) {
        sink.err = open_memstream(&held, &held_n);
        if (!sink.err) return die("open_memstream");
        const bool ok = sink.drain() >= 0 && finish(ctx, sink) >= 0;
        fclose(sink.err);                  // (completes `held`)
        sink.err = stderr;
        if (!ok) {                         // what was decoded before the failure is not lost with it
            if (held) { fwrite(held, 1, held_n, stderr); free(held); }
            return die("drain / finish");
        }
        if (opv_get_state(ctx, 0, &st) < 0) return die("opv_get_state");
    }
    if (!o.quiet) {
        fprintf(stderr, "Estimated carrier offset: %.1f Hz\n", st.est_offset_hz);  // ref :1151 / :1170
        if (o.coherent) fprintf(stderr, "PLL bandwidth: %.1f Hz\n", o.pll_bw);       // ref :1157
        fprintf(stderr, "Demodulated %llu symbols, final AFC offset: %.1f Hz\n\n", (unsigned long long)st.total_symbols, st.freq_offset_hz);
    }
    if (held) { fwrite(held, 1, held_n, stderr); free(held); }
    if (sink.drain() < 0) return die("drain");
    if (finish(ctx, sink) < 0) return die("finish");
    if (opv_get_state(ctx, 0, &st) < 0) return die("opv_get_state");
    if (!o.quiet) {  // ref :1208-1214
        fprintf(stderr, "════════════════════════════════════════════════════════════════════\n");
        fprintf(stderr, "Summary: %d frames (%d perfect, %d errors)\n", sink.decoded, sink.perfect, sink.decoded - sink.perfect);
        fprintf(stderr, "Final state: %s, AFC: %.1f Hz\n", state_name(st.sync_state), st.freq_offset_hz);
        fprintf(stderr, "════════════════════════════════════════════════════════════════════\n");
    }
    opv_destroy(ctx);
    return sink.decoded > 0 ? 0 : 1;
}
