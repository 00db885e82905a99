// Cost of handing a value from one wave to another wave of the same workgroup through LDS (dev tool).
// Two waves on one CU play ping-pong on two LDS words by polling; variant B inserts an s_barrier pair.
// Reported: cycles per ROUND TRIP (two hand-overs) at the measured kernel time and an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool kBarrier>
__global__ __launch_bounds__(128) void k_pingpong(int n, int* out) {
    __shared__ volatile int box[64];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) box[threadIdx.x] = 0;
    __syncthreads();
    int seen = 0;
    for (int i = 1; i <= n; ++i) {
        if constexpr (kBarrier) {
            if (wave == 0) box[0] = i;
            __syncthreads();
            seen += box[0];
            if (wave == 1) box[16] = i;
            __syncthreads();
            seen += box[16];
        } else {
            if (wave == 0) {
                box[0] = i;                       // hand over
                while (box[16] != i) {}           // wait for the answer
            } else {
                while (box[0] != i) {}
                box[16] = i;
            }
        }
    }
    if (threadIdx.x == 0) out[0] = seen + box[16];
}

int main() {
    int* d;
    hipMalloc(&d, 4);
    const int n = 2000000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (variant == 0) k_pingpong<false><<<1, 128>>>(n, d); else k_pingpong<true><<<1, 128>>>(n, d);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.1f ns per round trip = %.0f cycles @2.4 GHz (one hand-over ~%.0f cycles)\n",
               variant == 0 ? "polling ping-pong" : "LDS write + s_barrier + read, twice", ms * 1e6 / n, ms * 1e6 / n * 2.4,
               ms * 1e6 / n * 2.4 / 2);
    }
    return 0;
}
