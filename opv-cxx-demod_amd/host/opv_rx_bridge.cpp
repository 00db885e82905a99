// opv_rx_bridge.cpp — multi-stream receive bridge: N IQ byte streams -> ONE GPU context -> UDP.
//
// The MI355X-side counterpart of the reference modem's RX mode (`opv-modem -R`, reference
// src/opv-modem.cpp:673-838: stdin IQ -> `opv-demod -s -r` child -> 134-byte UDP datagrams to
// port 57373). Instead of one child process per stream it drives the C ABI
// directly: every input is a stream of one opv_ctx, fed in the reference's 16 KB reads
// (opv-modem.cpp:734,753), and each decoded frame leaves as one datagram, exactly 134 bytes
// (opv-modem.cpp:782-783). Stream k sends to UDP port base+k. The demodulator's inherent
// one-frame latency (a frame is released once 50 samples of the next one have arrived,
// src/opv-demod.cpp:221) is unchanged.
//
//   opv-rx-bridge [-H host] [-P base_port] [-o hz] [-a alpha] [--device n | --devices a,b,...] [--gather] [-q] [input ...]
//     --devices a,b,...  one context per listed GPU, the inputs sharded contiguously over them (stream k -> entry
//                        k / ceil(S / N) of the list; BASELINE configs[4]'s layout in one C++ process); UDP output unchanged
//     --gather           at the end, the decoded-frame buffers and counts of every context are gathered to the first
//                        listed GPU with ONE RCCL gather (opv_gather_frames_all, ncclGather) and compared with what was sent
//     inputs, one stream each ("N stdin/UDP sources"):
//       PATH      file or FIFO with int16 I/Q
//       -         stdin (also the only stream when no input is named, like `opv-modem -R`)
//       udp:PORT  IQ bytes arriving as UDP datagrams on 127.0.0.1:PORT (what a network SDR front-end sends);
//                 a zero-length datagram ends the stream
#include <arpa/inet.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>   // the gather's receive buffers are this host's own device memory

#include "../../include/opv_demod.h"

namespace {
struct Input {
    int fd = -1;
    bool eof = false, udp = false;
    unsigned char carry[4];
    size_t ncarry = 0;
    long frames = 0, perfect = 0;
    uint64_t samples = 0;
};
}  // namespace

int main(int argc, char** argv) {
    std::string host = "127.0.0.1";
    int base_port = 57373;  // OPV network port (opv-modem.cpp:567-568)
    bool quiet = false, have_off = false;
    double off = 0.0, afc = 0.001;
    int device = 0;
    std::vector<int> devices;
    bool gather = false;
    std::vector<std::string> paths;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-H") && i + 1 < argc) host = argv[++i];
        else if (!strcmp(argv[i], "-P") && i + 1 < argc) base_port = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) { off = atof(argv[++i]); have_off = true; }
        else if (!strcmp(argv[i], "-a") && i + 1 < argc) afc = atof(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (const char* p = argv[++i]; *p;) {
                char* end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p || v < 0 || v > 1023 || (*end != ',' && *end != 0) || (*end == ',' && end[1] == 0)) {     // ("x,y", "0,,1", "0,1,", "-1": not a list of ordinals)
                    fprintf(stderr, "opv-rx-bridge: --devices takes a comma-separated list of HIP device ordinals, got '%s'\n", argv[i]);
                    return 2;
                }
                devices.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
            }
            if (devices.empty()) { fprintf(stderr, "opv-rx-bridge: --devices takes a comma-separated list of HIP device ordinals\n"); return 2; }
        } else if (!strcmp(argv[i], "--gather")) gather = true;
        else if (!strcmp(argv[i], "-q")) quiet = true;
        else if (!strcmp(argv[i], "-h")) {
            fprintf(stderr, "Usage: %s [-H host] [-P base_port] [-o hz] [-a alpha] [--device n | --devices a,b,...] [--gather] [-q] [input ...]\n", argv[0]);
            return 0;
        } else paths.push_back(argv[i]);
    }
    const int S = paths.empty() ? 1 : (int)paths.size();
    if (base_port < 1 || base_port + S - 1 > 65535) {
        fprintf(stderr, "opv-rx-bridge: -P %d: ports %d..%d are not valid UDP ports\n", base_port, base_port, base_port + S - 1);
        return 2;
    }
    std::vector<Input> in(S);
    for (int k = 0; k < S; ++k) {
        const std::string p = paths.empty() ? "-" : paths[k];
        if (p == "-") in[k].fd = STDIN_FILENO;
        else if (p.rfind("udp:", 0) == 0) {
            in[k].udp = true;
            in[k].fd = socket(AF_INET, SOCK_DGRAM, 0);
            sockaddr_in a{};
            a.sin_family = AF_INET;
            a.sin_port = htons((uint16_t)atoi(p.c_str() + 4));
            a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
            const int big = 8 << 20;
            if (in[k].fd >= 0) setsockopt(in[k].fd, SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
            if (in[k].fd < 0 || bind(in[k].fd, (sockaddr*)&a, sizeof a) < 0) { perror(p.c_str()); return 2; }
        } else in[k].fd = open(p.c_str(), O_RDONLY | O_NONBLOCK);
        if (in[k].fd < 0) { perror(p.c_str()); return 2; }
        fcntl(in[k].fd, F_SETFL, fcntl(in[k].fd, F_GETFL) | O_NONBLOCK);
    }
    const int sock = socket(AF_INET, SOCK_DGRAM, 0);
    if (sock < 0) { perror("socket"); return 2; }
    std::vector<sockaddr_in> dst(S);
    for (int k = 0; k < S; ++k) {
        memset(&dst[k], 0, sizeof dst[k]);
        dst[k].sin_family = AF_INET;
        dst[k].sin_port = htons((uint16_t)(base_port + k));
        if (inet_pton(AF_INET, host.c_str(), &dst[k].sin_addr) != 1) { fprintf(stderr, "bad host %s\n", host.c_str()); return 2; }
    }

    // streams shard contiguously over the listed GPUs: global stream k is stream k % per of context k / per
    if (devices.empty()) devices.push_back(device);
    const int D = (int)devices.size(), per = (S + D - 1) / D;
    opv_cfg cfg{};
    cfg.streaming = 1;
    cfg.have_init_offset = have_off;
    cfg.init_offset_hz = off;
    cfg.afc_alpha = afc;
    cfg.pll_bw_hz = 50.0;
    cfg.max_samples = 8 * OPV_CHUNK_SAMPLES;  // staging buffer per stream; streams themselves are unbounded
    std::vector<opv_ctx*> ctxs(D, nullptr);
    for (int d = 0; d < D; ++d) {
        cfg.device = devices[d];
        if (opv_create(&ctxs[d], per, &cfg) < 0) { fprintf(stderr, "opv-rx-bridge: device %d: %s\n", devices[d], opv_last_error()); return 2; }
    }
    if (!quiet) {
        fprintf(stderr, "opv-rx-bridge: %d stream(s) -> udp://%s:%d..%d", S, host.c_str(), base_port, base_port + S - 1);
        if (D > 1) fprintf(stderr, ", %d per context on %d contexts (GPUs", per, D);
        for (int d = 0; D > 1 && d < D; ++d) fprintf(stderr, "%s%d", d ? "," : " ", devices[d]);
        fprintf(stderr, D > 1 ? ")\n" : "\n");
    }

    std::vector<pollfd> pfd(S);
    constexpr size_t kRead = 16384;                       // opv-modem's read size (src/opv-modem.cpp:734,753)
    constexpr size_t kDgram = 65536;                      // a UDP datagram is taken whole (<= 65507 bytes of payload)
    constexpr size_t kRound = 1u << 20;                   // at most this much per stream and poll round (0.12 s of IQ)
    // one buffer per stream, 16-byte aligned, in PINNED host memory: a poll round is ONE batched push, and opv_push_iq_batch
    // moves blocks that lie in pinned memory with one gather kernel at the link's rate instead of one copy per stream
    constexpr size_t kStride = kRound + kDgram + 16;
    unsigned char* bufs = nullptr;
    // (one context only: with a context per GPU the buffers would have to be pinned per device; pageable then, as before)
    bool bufs_pinned = D == 1 && hipSetDevice(devices[0]) == hipSuccess && hipHostMalloc((void**)&bufs, (size_t)S * kStride, hipHostMallocDefault) == hipSuccess;
    if (!bufs_pinned) bufs = (unsigned char*)malloc((size_t)S * kStride);     // (pageable works too: per-stream copies)
    if (!bufs) { fprintf(stderr, "opv-rx-bridge: out of host memory\n"); return 2; }
    std::vector<int> pending_flush;
    std::vector<std::vector<int>> ids(D);
    std::vector<std::vector<const int16_t*>> ptrs(D);
    std::vector<std::vector<size_t>> lens(D);
    uint8_t frames[64 * OPV_FRAME_BYTES];
    opv_frame_meta meta[64];
    int open_streams = S;
    auto drain = [&]() -> int {
        for (int d = 0; d < D; ++d)
            if (opv_process(ctxs[d]) < 0) return -1;          // asynchronous: every GPU works while the first one is popped
        for (int k = 0; k < S; ++k) {
            for (;;) {
                const long n = opv_pop_frames(ctxs[k / per], k % per, frames, 64, meta);
                if (n < 0) return -1;
                for (long f = 0; f < n; ++f) {
                    sendto(sock, frames + f * OPV_FRAME_BYTES, OPV_FRAME_BYTES, 0, (sockaddr*)&dst[k], sizeof dst[k]);
                    in[k].frames++;
                    if (meta[f].viterbi_metric == 0) in[k].perfect++;
                }
                if (n < 64) break;
            }
        }
        return 0;
    };
    while (open_streams > 0) {
        for (int k = 0; k < S; ++k) { pfd[k].fd = in[k].eof ? -1 : in[k].fd; pfd[k].events = POLLIN; pfd[k].revents = 0; }
        if (poll(pfd.data(), S, 10) < 0) break;  // 10 ms like the reference's select timeout
        bool any = false;
        for (int d = 0; d < D; ++d) { ids[d].clear(); ptrs[d].clear(); lens[d].clear(); }
        for (int k = 0; k < S; ++k) {
            if (in[k].eof) continue;
            if (!(pfd[k].revents & (POLLIN | POLLHUP))) continue;
            // everything the source has ready goes into this round (reads of 16 KB like the reference's loop, a
            // datagram at a time for UDP): one push + one opv_process per poll round, however fast the data arrives
            unsigned char* buf = bufs + (size_t)k * kStride;
            memcpy(buf, in[k].carry, in[k].ncarry);
            size_t have = in[k].ncarry;
            bool ended = false;
            while (have < kRound) {
                const ssize_t r = read(in[k].fd, buf + have, in[k].udp ? kDgram : kRead);
                if (r < 0) break;                         // EAGAIN: drained for now
                if (r == 0) { ended = true; break; }      // EOF / the empty datagram that ends a UDP stream
                have += (size_t)r;
            }
            const size_t ns = have / 4;
            if (ns) { ids[k / per].push_back(k % per); ptrs[k / per].push_back(reinterpret_cast<const int16_t*>(buf)); lens[k / per].push_back(ns); any = true; }
            in[k].samples += ns;
            in[k].ncarry = have - ns * 4;
            memcpy(in[k].carry, buf + ns * 4, in[k].ncarry);
            if (ended) {
                in[k].eof = true;
                --open_streams;
                pending_flush.push_back(k);
                any = true;
            }
        }
        for (int d = 0; d < D; ++d)
            if (!ids[d].empty() && opv_push_iq_batch(ctxs[d], (int)ids[d].size(), ids[d].data(), ptrs[d].data(), lens[d].data()) < 0) {
                fprintf(stderr, "opv-rx-bridge: %s\n", opv_last_error());
                return 2;
            }
        for (int k : pending_flush)
            if (opv_flush(ctxs[k / per], k % per) < 0) { fprintf(stderr, "opv-rx-bridge: %s\n", opv_last_error()); return 2; }
        pending_flush.clear();
        if (any && drain() < 0) { fprintf(stderr, "opv-rx-bridge: %s\n", opv_last_error()); return 2; }
    }
    if (drain() < 0) { fprintf(stderr, "opv-rx-bridge: %s\n", opv_last_error()); return 2; }
    // streams held back by back-pressure (opv_stream_state.stalled) have not finished: keep going until none is, and
    // fail loudly if a round moves nothing
    auto progress = [&](bool* stalled) -> uint64_t {
        uint64_t p = 0;
        *stalled = false;
        for (int k = 0; k < S; ++k) {
            opv_stream_state st;
            if (opv_get_state(ctxs[k / per], k % per, &st) < 0) continue;
            *stalled |= st.stalled != 0;
            p += st.total_symbols + (uint64_t)st.frames_released + (uint64_t)in[k].frames;
        }
        return p;
    };
    for (;;) {
        bool stalled = false;
        const uint64_t p0 = progress(&stalled);
        if (!stalled) break;
        if (drain() < 0) { fprintf(stderr, "opv-rx-bridge: %s\n", opv_last_error()); return 2; }
        if (progress(&stalled) == p0 && stalled) { fprintf(stderr, "opv-rx-bridge: a stream stalled without progress\n"); return 2; }
    }
    long total = 0;
    for (int k = 0; k < S; ++k) {
        total += in[k].frames;
        if (!quiet)
            fprintf(stderr, "stream %d: %.3f s of IQ, %ld frames (%ld perfect) -> port %d\n", k, in[k].samples / 2168000.0,
                    in[k].frames, in[k].perfect, base_port + k);
    }
    int rc = total > 0 ? 0 : 1;
    if (gather) {
        // BASELINE configs[4]'s collective from a C++ host: every context's [per][cap][134] frame buffer + [per] counts to the
        // first listed GPU with one RCCL gather (grouped over the contexts this thread owns), then a look at what arrived
        size_t cap = 0;
        const int32_t* d_counts0 = nullptr;
        opv_device_frames(ctxs[0], nullptr, nullptr, &d_counts0, &cap);
        std::vector<void*> comms(D, nullptr);
        uint8_t* d_all = nullptr;
        int32_t* d_cnt = nullptr;
        const size_t nb = (size_t)per * cap * OPV_FRAME_BYTES;
        bool ok = opv_comm_init_all(comms.data(), D, devices.data()) == 0;
        if (!ok) fprintf(stderr, "opv-rx-bridge: gather: %s\n", opv_last_error());
        ok = ok && hipSetDevice(devices[0]) == hipSuccess && hipMalloc((void**)&d_all, nb * D) == hipSuccess &&
             hipMalloc((void**)&d_cnt, sizeof(int32_t) * per * D) == hipSuccess;
        if (ok && opv_gather_frames_all(ctxs.data(), comms.data(), D, 0, d_all, d_cnt) < 0) { fprintf(stderr, "opv-rx-bridge: gather: %s\n", opv_last_error()); ok = false; }
        for (int d = 0; ok && d < D; ++d) ok = opv_sync(ctxs[d]) == 0;
        std::vector<int32_t> cnt((size_t)per * D);
        ok = ok && hipMemcpy(cnt.data(), d_cnt, sizeof(int32_t) * cnt.size(), hipMemcpyDeviceToHost) == hipSuccess;
        long gathered = 0, mismatch = 0;
        for (int k = 0; ok && k < S; ++k) {
            opv_stream_state st;
            opv_get_state(ctxs[k / per], k % per, &st);
            gathered += cnt[k];                                // rank-major = global stream order
            mismatch += cnt[k] != st.frames_released;
        }
        if (ok) fprintf(stderr, "gather: %d rank(s) x %d stream(s) -> GPU %d over RCCL: %ld frames released in all, %ld stream(s) differ from the local counts\n",
                        D, per, devices[0], gathered, mismatch);
        if (!ok || mismatch) rc = 2;
        if (d_all) (void)hipFree(d_all);
        if (d_cnt) (void)hipFree(d_cnt);
        for (void* c : comms) opv_comm_destroy(c);
    }
    for (opv_ctx* c : ctxs) opv_destroy(c);
    if (bufs_pinned) (void)hipHostFree(bufs);
    else free(bufs);
    close(sock);
    return rc;
}
