// loop_size.hip — does a lone wave's cost per instruction depend on the SIZE of the loop body (instruction fetch)?
// N dependent-free v_fma_f64 (8 bytes each) per trip, 8-byte aligned, N = 64 ... 2560 (0.5 KB ... 20 KB of code).
// Build: hipcc -O3 --offload-arch=gfx950 -o loop_size loop_size.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define U4(x) x x x x
#define U16(x) U4(x) U4(x) U4(x) U4(x)
#define U64(x) U16(x) U16(x) U16(x) U16(x)
#define U640(x) U64(x) U64(x) U64(x) U64(x) U64(x) U64(x) U64(x) U64(x) U64(x) U64(x)
#define BODY "v_fma_f64 v[20:21], v[24:25], v[28:29], v[32:33]\n"
#define CLOB "v20","v21"

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, int rep) {
    double a = 1.0, b = 0.5, c = 0.25;
    asm volatile("v_mov_b64 v[24:25], %0\n v_mov_b64 v[28:29], %1\n v_mov_b64 v[32:33], %2" : : "v"(a), "v"(b), "v"(c) : "v24","v25","v28","v29","v32","v33");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        if (MODE == 0) asm volatile(".p2align 3\n" U64(BODY) ::: CLOB);
        if (MODE == 1) asm volatile(".p2align 3\n" U64(BODY) U64(BODY) U64(BODY) U64(BODY) ::: CLOB);
        if (MODE == 2) asm volatile(".p2align 3\n" U640(BODY) ::: CLOB);
        if (MODE == 3) asm volatile(".p2align 3\n" U640(BODY) U640(BODY) ::: CLOB);
        if (MODE == 4) asm volatile(".p2align 3\n" U640(BODY) U640(BODY) U640(BODY) U640(BODY) ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double v; asm volatile("v_mov_b64 %0, v[20:21]" : "=v"(v));
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(int n, double* d, unsigned long long* c) {
    const int rep = 200000 / n + 10;
    for (int w = 0; w < 3; ++w) k<MODE><<<1, 64>>>(d, c, rep);
    (void)hipDeviceSynchronize();
    unsigned long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%5d instructions per trip (%5.1f KB): %.3f cycles per instruction (%.3f without the 36-cycle back-edge)\n", n, n * 8 / 1024.0,
           (double)cy / rep / n, ((double)cy / rep - 36.0) / n);
}

int main() {
    double* d; unsigned long long* c;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 8);
    run<0>(64, d, c); run<1>(256, d, c); run<2>(640, d, c); run<3>(1280, d, c); run<4>(2560, d, c);
    return 0;
}
