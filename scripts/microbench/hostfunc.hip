// hostfunc.hip — is hipLaunchHostFunc usable for a stream-ordered host decision between two kernels (round 6: the offset
// search's tie decision, opv_capi.hip), and what does it cost?
//   a) correctness: kernel A writes pinned host memory, a host function reads it and writes a result into pinned memory,
//      kernel B reads that result and stores it in device memory - the value must have made the round trip, 1000 times
//   b) host cost of the three enqueues (kernel, host function, kernel): what opv_process pays per pass
//   c) stream cost of one pass: hipEvent time of N back-to-back passes whose host function does nothing, against the
//      same 2 N kernels without host functions
//   d) the copy engine / a second stream keep running while a host function sleeps 2 ms on the first stream
// Build: hipcc -O3 --offload-arch=gfx950 -o hostfunc hostfunc.hip -lpthread      Run: ./hostfunc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Stage { volatile uint32_t from_dev; volatile uint32_t from_host; volatile uint32_t calls; uint32_t sleep_us; };

__global__ void k_a(Stage* st, uint32_t v) { if (threadIdx.x == 0) st->from_dev = v; }
__global__ void k_b(const Stage* st, uint32_t* out) { if (threadIdx.x == 0) *out = st->from_host; }
__global__ void k_spin(uint64_t ticks, uint32_t* out) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0) *out = 1;
}

static void host_fn(void* p) {
    Stage* st = (Stage*)p;
    st->from_host = st->from_dev * 3u + 1u;
    st->calls = st->calls + 1;
    if (st->sleep_us) std::this_thread::sleep_for(std::chrono::microseconds(st->sleep_us));
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    Stage* st = nullptr;
    CK(hipHostMalloc((void**)&st, sizeof(Stage), hipHostMallocDefault));
    st->from_dev = st->from_host = st->calls = 0; st->sleep_us = 0;
    uint32_t* d_out = nullptr;
    CK(hipMalloc((void**)&d_out, 4));
    uint32_t h_out = 0;

    // a) round trips
    int bad = 0;
    for (uint32_t i = 1; i <= 1000; ++i) {
        k_a<<<1, 64, 0, s>>>(st, i);
        CK(hipLaunchHostFunc(s, host_fn, st));
        k_b<<<1, 64, 0, s>>>(st, d_out);
        CK(hipMemcpyAsync(&h_out, d_out, 4, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        bad += h_out != i * 3u + 1u;
    }
    printf("a) 1000 kernel -> host function -> kernel round trips: %d wrong, host function called %u times\n", bad, st->calls);

    // b) host cost of enqueueing one pass, queue kept shallow
    {
        double t_pass = 0, t_k = 0;
        for (int r = 0; r < 200; ++r) {
            double t0 = now_us();
            k_a<<<1, 64, 0, s>>>(st, 7);
            CK(hipLaunchHostFunc(s, host_fn, st));
            k_b<<<1, 64, 0, s>>>(st, d_out);
            t_pass += now_us() - t0;
            CK(hipStreamSynchronize(s));
            t0 = now_us();
            k_a<<<1, 64, 0, s>>>(st, 7);
            k_b<<<1, 64, 0, s>>>(st, d_out);
            t_k += now_us() - t0;
            CK(hipStreamSynchronize(s));
        }
        printf("b) host time to enqueue kernel + host function + kernel: %.1f us (two kernels alone: %.1f us)\n", t_pass / 200, t_k / 200);
    }
    // b2) eight passes enqueued back to back (a 2048-stream search round with 256 slots)
    {
        double t8 = 0;
        for (int r = 0; r < 50; ++r) {
            const double t0 = now_us();
            for (int p = 0; p < 8; ++p) {
                k_a<<<1, 64, 0, s>>>(st, 7);
                CK(hipLaunchHostFunc(s, host_fn, st));
                k_b<<<1, 64, 0, s>>>(st, d_out);
            }
            t8 += now_us() - t0;
            CK(hipStreamSynchronize(s));
        }
        printf("b2) host time to enqueue 8 passes: %.1f us\n", t8 / 50);
    }
    // c) stream cost of a pass
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int n : {1, 8, 64}) {
            float with = 0, without = 0;
            for (int r = 0; r < 20; ++r) {
                float ms;
                CK(hipEventRecord(e0, s));
                for (int p = 0; p < n; ++p) { k_a<<<1, 64, 0, s>>>(st, 7); CK(hipLaunchHostFunc(s, host_fn, st)); k_b<<<1, 64, 0, s>>>(st, d_out); }
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); with += ms;
                CK(hipEventRecord(e0, s));
                for (int p = 0; p < n; ++p) { k_a<<<1, 64, 0, s>>>(st, 7); k_b<<<1, 64, 0, s>>>(st, d_out); }
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); without += ms;
            }
            printf("c) %2d passes on the stream: %.1f us with host functions, %.1f us without -> %.1f us per host function\n", n, with * 50, without * 50,
                   (with - without) * 50 / n);
        }
    }
    // d) another stream keeps going while the host function sleeps
    {
        st->sleep_us = 2000;
        uint32_t* d2 = nullptr;
        CK(hipMalloc((void**)&d2, 4));
        const double t0 = now_us();
        k_a<<<1, 64, 0, s>>>(st, 9);
        CK(hipLaunchHostFunc(s, host_fn, st));
        k_b<<<1, 64, 0, s>>>(st, d_out);
        k_spin<<<1, 64, 0, s2>>>(10000 /* 100 us */, d2);
        CK(hipStreamSynchronize(s2));
        const double t_other = now_us() - t0;
        CK(hipStreamSynchronize(s));
        const double t_all = now_us() - t0;
        st->sleep_us = 0;
        printf("d) host function sleeping 2000 us on stream 1: a 100 us kernel on stream 2 done after %.0f us, stream 1 after %.0f us\n", t_other, t_all);
    }
    return bad ? 1 : 0;
}
