// exec_mask.hip — does a lone wave's VALU instruction get cheaper when only one row of 16 lanes is enabled?
// (the front-end's loop filters are wave-uniform: if a 16-lane EXEC skipped three of the four passes, its scalar tail
// could run on one row). Prints cycles per pass of 64 dependent v_fma_f64 / v_fma_f32 with EXEC = all, 16, 1 lanes.
// Build: hipcc -O3 --offload-arch=gfx950 -o exec_mask exec_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define U4(x) x x x x
#define U16(x) U4(x) U4(x) U4(x) U4(x)
#define U64(x) U16(x) U16(x) U16(x) U16(x)

template <int KIND>
__global__ void k(double* out, unsigned long long* cyc, int rep, unsigned long long mask) {
    double* po = out + blockIdx.x * 64 + threadIdx.x;                       // every per-lane address exists BEFORE the mask changes: hipcc does not
    double a = *po, b = 1.0000001, c = 1e-9;              // know about the s_mov to exec and may move VALU code across it
    float af = (float)a, bf = 1.0000001f, cf = 1e-9f;
    asm volatile("" : "+v"(po), "+v"(a), "+v"(af), "+v"(b), "+v"(c), "+v"(bf), "+v"(cf));
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(keep) : "s"(mask));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        if (KIND == 0) asm volatile(U64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        else asm volatile(U64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(af) : "v"(bf), "v"(cf));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 exec, %0" : : "s"(keep));
    asm volatile("" : "+v"(po), "+v"(a), "+v"(af));
    *po = a + af;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 64 * 64 * 8); hipMalloc(&c, 8);
    std::vector<double> h(64, 1.0);
    const unsigned long long masks[3] = {~0ull, 0xFFFFull, 1ull};
    const char* names[3] = {"64 lanes", "16 lanes (one row)", "1 lane"};
    for (int kind = 0; kind < 2; ++kind)
        for (int m = 0; m < 3; ++m) {
            hipMemcpy(d, h.data(), 64 * 8, hipMemcpyHostToDevice);
            for (int w = 0; w < 2; ++w) {
                if (kind == 0) k<0><<<1, 64>>>(d, c, 2000, masks[m]); else k<1><<<1, 64>>>(d, c, 2000, masks[m]);
            }
            hipDeviceSynchronize();
            unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            printf("%-8s EXEC = %-20s %6.2f cycles per instruction (64 dependent per pass, loop included)\n", kind ? "fma_f32" : "fma_f64", names[m], (double)cy / 2000 / 64);
        }
    // the same launch twelve times over: does the figure depend on anything but the code?
    for (int grid : {1, 64}) {
        printf("fma_f64, 64 lanes, %d workgroup(s), twelve launches:", grid);
        for (int i = 0; i < 12; ++i) {
            k<0><<<grid, 64>>>(d, c, 2000, ~0ull);
            hipDeviceSynchronize();
            unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            printf(" %.2f", (double)cy / 2000 / 64);
        }
        printf("\n");
    }
    return 0;
}
