// h2d_many.hip — how fast do N separate 346 880-byte blocks (one 40 ms chunk of int16 IQ per live stream) cross PCIe from
// pinned host memory into N separate device buffers? The serving path (opv_push_iq_batch) is bound by this, not by its
// kernels (bench.py extras.live_capacity: 1536 streams, 23.8 of 26.5 ms per round are the copies, 22 GB/s).
//   a) N hipMemcpyAsync on ONE stream (what opv_push_iq_batch did)        b) the same spread over 2 / 4 / 8 streams
//   c) ONE kernel that reads the host blocks through their device-visible addresses (16 B per lane, grid-stride over a
//      table of {src, dst} pairs) - pinned memory from hipHostMalloc is mapped into the device's address space
//   d) one hipMemcpyAsync of the same total size (the link's rate for a large transfer)
// Build: hipcc -O3 --offload-arch=gfx950 -o h2d_many h2d_many.hip      Run: ./h2d_many [n_blocks (1536)] [reps (10)]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Pair { const int4* src; int4* dst; };

// one block of threads per (pair, slice): slices of `per` int4 each
__global__ __launch_bounds__(256) void k_gather(const Pair* tab, unsigned n_pairs, unsigned quads, unsigned slices) {
    const unsigned per = (quads + slices - 1) / slices;
    for (unsigned w = blockIdx.x; w < n_pairs * slices; w += gridDim.x) {
        const Pair p = tab[w / slices];
        const unsigned lo = (w % slices) * per, hi = lo + per < quads ? lo + per : quads;
        for (unsigned i = lo + threadIdx.x; i < hi; i += 256) p.dst[i] = p.src[i];
    }
}

int main(int argc, char** argv) {
    const unsigned N = argc > 1 ? atoi(argv[1]) : 1536, reps = argc > 2 ? atoi(argv[2]) : 10;
    const size_t blk = 346880;
    char* h = nullptr;
    CK(hipHostMalloc((void**)&h, blk * N, hipHostMallocDefault));
    for (size_t i = 0; i < blk * N; i += 4096) h[i] = (char)i;
    std::vector<char*> d(N);
    for (unsigned k = 0; k < N; ++k) CK(hipMalloc((void**)&d[k], blk + 16384));
    char* dbig = nullptr;
    CK(hipMalloc((void**)&dbig, blk * N));
    hipStream_t st[8];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    using clk = std::chrono::steady_clock;
    auto gbs = [&](clk::time_point a, clk::time_point b) { return (double)blk * N * reps / std::chrono::duration<double>(b - a).count() / 1e9; };

    for (int ns : {1, 2, 4, 8}) {
        for (int warm = 0; warm < 2; ++warm) {
            const auto t0 = clk::now();
            for (unsigned r = 0; r < (warm ? reps : 1); ++r) {
                for (unsigned k = 0; k < N; ++k) CK(hipMemcpyAsync(d[k], h + blk * k, blk, hipMemcpyHostToDevice, st[k % ns]));
                for (int s = 0; s < ns; ++s) CK(hipStreamSynchronize(st[s]));
            }
            if (warm) printf("%u x %zu B hipMemcpyAsync over %d stream(s): %.1f GB/s\n", N, blk, ns, gbs(t0, clk::now()));
        }
    }
    {   // the kernel: table of pairs, host blocks by their device-visible addresses
        std::vector<Pair> tab(N);
        for (unsigned k = 0; k < N; ++k) {
            void* dv = nullptr;
            CK(hipHostGetDevicePointer(&dv, h + blk * k, 0));
            tab[k] = {(const int4*)dv, (int4*)d[k]};
        }
        Pair* dtab = nullptr;
        CK(hipMalloc((void**)&dtab, sizeof(Pair) * N));
        CK(hipMemcpy(dtab, tab.data(), sizeof(Pair) * N, hipMemcpyHostToDevice));
        for (unsigned slices : {1u, 4u, 16u})
            for (unsigned grid : {256u, 1024u, 4096u}) {
                for (int warm = 0; warm < 2; ++warm) {
                    const auto t0 = clk::now();
                    for (unsigned r = 0; r < (warm ? reps : 1); ++r) {
                        k_gather<<<grid, 256, 0, st[0]>>>(dtab, N, (unsigned)(blk / 16), slices);
                        CK(hipStreamSynchronize(st[0]));
                    }
                    if (warm) printf("one gather kernel, %u blocks of 256 threads, %u slice(s) per block of IQ: %.1f GB/s\n", grid, slices, gbs(t0, clk::now()));
                }
            }
        // did it copy? compare one block through a D2H
        std::vector<char> back(blk);
        CK(hipMemcpy(back.data(), d[N - 1], blk, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < blk; ++i) bad += back[i] != h[blk * (N - 1) + i];
        printf("gather check: %zu differing bytes in the last block\n", bad);
    }
    for (int warm = 0; warm < 2; ++warm) {
        const auto t0 = clk::now();
        for (unsigned r = 0; r < (warm ? reps : 1); ++r) {
            CK(hipMemcpyAsync(dbig, h, blk * N, hipMemcpyHostToDevice, st[0]));
            CK(hipStreamSynchronize(st[0]));
        }
        if (warm) printf("one hipMemcpyAsync of %.0f MB: %.1f GB/s\n", blk * N / 1e6, gbs(t0, clk::now()));
    }
    return 0;
}
