// loop_align.hip — does the cost of a lone wave's instruction depend on where its loop sits in memory?
// One kernel per PAD: PAD s_nop instructions (4 bytes each) in front of a loop of 64 dependent v_fma_f64 (8 bytes each).
// Build: hipcc -O3 --offload-arch=gfx950 -o loop_align loop_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define U4(x) x x x x
#define U16(x) U4(x) U4(x) U4(x) U4(x)
#define U64(x) U16(x) U16(x) U16(x) U16(x)

template <int PAD>
__global__ void k(double* out, unsigned long long* cyc, int rep) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
#pragma unroll
    for (int i = 0; i < PAD; ++i) asm volatile("s_nop 0");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) asm volatile(U64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PAD>
static void run(double* d, unsigned long long* c) {
    for (int w = 0; w < 3; ++w) k<PAD><<<1, 64>>>(d, c, 2000);
    hipDeviceSynchronize();
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf(" %.2f", (double)cy / 2000 / 64);
}

int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    hipMemset(d, 0, 64 * 8);
    printf("cycles per v_fma_f64 (64 dependent per pass) with 0..17 s_nop in front of the loop:\n");
    run<0>(d, c); run<1>(d, c); run<2>(d, c); run<3>(d, c); run<4>(d, c); run<5>(d, c); run<6>(d, c); run<7>(d, c); run<8>(d, c);
    run<9>(d, c); run<10>(d, c); run<11>(d, c); run<12>(d, c); run<13>(d, c); run<14>(d, c); run<15>(d, c); run<16>(d, c); run<17>(d, c);
    printf("\nthe same kernel (PAD 0) six times:");
    for (int i = 0; i < 6; ++i) run<0>(d, c);
    printf("\n");
    return 0;
}
