// opv_live_capacity.cpp — how many LIVE streams one GPU context serves in real time, measured the way the boundary's real caller
// uses it (reference `opv-modem -R`, src/opv-modem.cpp:673-838: IQ arrives at 2.168 MSPS, one frame = 86 720 samples = 40 ms).
//
// A serving round is what opv-rx-bridge does for every 40 ms of signal, for N streams at once: one 86 720-sample chunk per
// stream pushed from (pinned) host memory across PCIe (opv_push_iq_batch), one opv_process, every stream's frames popped
// (opv_pop_frames). The context keeps up with real time while a round takes less than the 40 ms of signal it consumes. This
// tool runs R rounds for a given N and prints the distribution of the round time as one JSON line; bench.py searches for the
// largest N whose p99 stays under 40 ms (extras.live_capacity). No kernel is specific to this tool: it is a caller of
// include/opv_demod.h like opv-rx-bridge, without sockets so that the number is the library's.
//
//   opv-live-capacity <n_streams> [rounds (120)] [warmup (6)] [device (0)] [--pipelined]
//     --pipelined   the double-buffered server: opv_push_iq_batch_async of round r + 1 is enqueued right behind opv_process of
//                   round r, so the chunks of the next round cross PCIe while this round's kernels run and its frames are
//                   popped; a round then costs max(PCIe, kernels + pops). Reported per round: the time from one opv_push_wait
//                   to the next (steady-state period); a chunk's frames surface one round later than in the serial loop.
//
// The signal is ONE clean BERT run of N + rounds + warmup + 1 frames (the device transmit chain, bit-identical to `opv-mod -S W5NYV
// -B ...`, brought back into pinned host memory); stream k listens to it from frame k on. So in every round every stream's
// chunk lies at its own host addresses - N x 347 KB of distinct pinned memory cross PCIe per round, nothing a device cache
// could serve twice (with one shared chunk the gather kernel "moved" 1.8 TB/s: L2 hits) - and every stream decodes its own
// frame sequence. Checked: after the first round every round releases exactly one frame per stream, equal to the transmitted one.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>   // pinned host memory for the chunks (what a DMA-fed SDR server would hold them in)

#include "../../include/opv_demod.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "Usage: %s <n_streams> [rounds] [warmup] [device]\n", argv[0]);
        return 2;
    }
    bool pipelined = false;
    if (argc > 2 && !strcmp(argv[argc - 1], "--pipelined")) { pipelined = true; --argc; }
    const int N = atoi(argv[1]);
    const int rounds = argc > 2 ? atoi(argv[2]) : 120, warm = argc > 3 ? atoi(argv[3]) : 6, device = argc > 4 ? atoi(argv[4]) : 0;
    if (N < 1 || rounds < 1 || warm < 1) { fprintf(stderr, "opv-live-capacity: bad arguments\n"); return 2; }
    const int total = rounds + warm;
    const size_t chunk = OPV_CHUNK_SAMPLES;

    // the signal: N + total + 1 frames (+ the modulator's 100 silent symbols), in pinned memory
    const size_t n_frames = (size_t)N + (size_t)total + 1;
    std::vector<uint8_t> frames(n_frames * OPV_FRAME_BYTES);
    opv_tx_bert_frames("W5NYV", 0xBBAADD, 0, n_frames, frames.data());
    const size_t n_all = opv_tx_modulated_samples(n_frames);
    int16_t* iq = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipHostMalloc((void**)&iq, n_all * 4, hipHostMallocDefault) != hipSuccess) {
        fprintf(stderr, "opv-live-capacity: no pinned host memory / no HIP device\n");
        return 2;
    }

    opv_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.streaming = 1;
    cfg.afc_alpha = 0.001;
    cfg.device = device;
    cfg.pll_bw_hz = 50.0;
    cfg.max_samples = 4 * chunk + 65536;               // a live server's staging buffer: a few chunks per stream
    opv_ctx* ctx = nullptr;
    if (opv_create(&ctx, N, &cfg) < 0) { fprintf(stderr, "opv-live-capacity: %s\n", opv_last_error()); return 2; }
    if (opv_tx_modulate_device_to_host(ctx, frames.data(), n_frames, iq) < 0) { fprintf(stderr, "opv-live-capacity: %s\n", opv_last_error()); return 2; }

    std::vector<int> ids(N);
    std::vector<const int16_t*> ptrs(N);
    std::vector<size_t> lens(N, chunk);
    for (int k = 0; k < N; ++k) ids[k] = k;
    std::vector<double> t_round, t_push, t_proc, t_pop, t_launch, t_enq;
    uint8_t out[4 * OPV_FRAME_BYTES];
    opv_frame_meta meta[4];
    long released = 0, wrong = 0, imperfect = 0, uneven = 0;
    std::vector<size_t> next(N, 0);                     // per stream: the number of the next frame it owes
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto set_ptrs = [&](int r) { for (int k = 0; k < N; ++k) ptrs[k] = iq + 2 * ((size_t)r + (size_t)k) * chunk; };   // stream k is k frames into the run
    if (pipelined) {
        set_ptrs(0);
        if (opv_push_iq_batch_async(ctx, N, ids.data(), ptrs.data(), lens.data()) < 0) { fprintf(stderr, "push: %s\n", opv_last_error()); return 2; }
    }
    for (int r = 0; r < total; ++r) {
        const auto t0 = clk::now();
        if (pipelined) {
            if (opv_push_wait(ctx) < 0) { fprintf(stderr, "push wait: %s\n", opv_last_error()); return 2; }   // round r's chunks have landed
        } else {
            set_ptrs(r);
            if (opv_push_iq_batch(ctx, N, ids.data(), ptrs.data(), lens.data()) < 0) { fprintf(stderr, "push: %s\n", opv_last_error()); return 2; }
        }
        const auto t1 = clk::now();
        if (opv_process(ctx) < 0) { fprintf(stderr, "process: %s\n", opv_last_error()); return 2; }
        const auto t1a = clk::now();
        if (pipelined && r + 1 < total) {                 // round r + 1 starts crossing PCIe behind round r's launches
            set_ptrs(r + 1);
            if (opv_push_iq_batch_async(ctx, N, ids.data(), ptrs.data(), lens.data()) < 0) { fprintf(stderr, "push: %s\n", opv_last_error()); return 2; }
        }
        const auto t1b = clk::now();
        if (opv_sync(ctx) < 0) { fprintf(stderr, "process: %s\n", opv_last_error()); return 2; }
        const auto t2 = clk::now();
        long got_round = 0;
        for (int k = 0; k < N; ++k) {
            const long g = opv_pop_frames(ctx, k, out, 4, meta);
            if (g < 0) { fprintf(stderr, "pop: %s\n", opv_last_error()); return 2; }
            for (long f = 0; f < g; ++f) {
                const size_t idx = (size_t)k + next[k]++;
                if (idx >= n_frames || memcmp(out + f * OPV_FRAME_BYTES, frames.data() + idx * OPV_FRAME_BYTES, OPV_FRAME_BYTES) != 0) ++wrong;
                if (meta[f].viterbi_metric != 0) ++imperfect;
            }
            got_round += g;
        }
        const auto t3 = clk::now();
        if (r >= 1 && got_round != N) ++uneven;         // (one frame per stream and round once the first chunk is in: the steady state)
        released += got_round;
        if (r >= warm) {
            t_round.push_back(ms(t0, t3));
            t_push.push_back(ms(t0, t1));
            t_proc.push_back(ms(t1, t2));
            t_pop.push_back(ms(t2, t3));
            t_launch.push_back(ms(t1, t1a));
            t_enq.push_back(ms(t1a, t1b));
        }
    }
    auto pct = [](std::vector<double> v, double p) {
        std::sort(v.begin(), v.end());
        const size_t i = (size_t)(p * (double)(v.size() - 1) + 0.5);
        return v[i < v.size() ? i : v.size() - 1];
    };
    printf("{\"streams\": %d, \"pipelined\": %s, \"rounds\": %d, \"signal_ms_per_round\": 40.0, \"round_ms_p50\": %.3f, \"round_ms_p99\": %.3f, \"round_ms_max\": %.3f, "
           "\"push_ms_p50\": %.3f, \"process_ms_p50\": %.3f, \"pop_ms_p50\": %.3f, \"pcie_GBps_p50\": %.2f, \"frames_released\": %ld, "
           "\"frames_wrong\": %ld, \"frames_imperfect\": %ld, \"rounds_not_one_frame_per_stream\": %ld, \"process_call_ms_p50\": %.3f, \"async_enqueue_ms_p50\": %.3f}\n",
           N, pipelined ? "true" : "false", rounds, pct(t_round, 0.5), pct(t_round, 0.99), pct(t_round, 1.0), pct(t_push, 0.5), pct(t_proc, 0.5), pct(t_pop, 0.5),
           // serial: the moves alone; pipelined: push_ms is only the wait for moves that ran beside the previous round - the rate over the period
           (double)N * chunk * 4 / (pct(pipelined ? t_round : t_push, 0.5) * 1e-3) / 1e9, released, wrong, imperfect, uneven, pct(t_launch, 0.5), pct(t_enq, 0.5));
    opv_destroy(ctx);
    (void)hipHostFree(iq);
    return wrong ? 1 : 0;
}
