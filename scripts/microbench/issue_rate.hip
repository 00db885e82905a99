// Which instruction kinds slow down when more SIMDs of a CU are busy? (dev tool)
// Single-wave workgroups (the front-end's LDS footprint, so at most 8 per CU) run a dependent chain of one
// instruction kind; cycles per instruction per wave for 256 / 512 / 1024 / 2048 groups = 1, 2, 4 busy SIMDs per
// CU with one wave each, then two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(64) void k_chain(int n, double* out) {
    __shared__ double pad[2380];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-9, a2 = a + 1, a3 = a + 2, a4 = a + 3;
    int lo = threadIdx.x, hi = 7;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) a = fma(a, b, c);                                                     // fp64 FMA
            if (KIND == 1) lo = __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, true) + 1;          // DPP row_ror + add
            if (KIND == 2) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)hi, false, false); lo = (int)r[0]; hi = (int)r[1]; }
            if (KIND == 3) { auto r = __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)hi, false, false); lo = (int)r[0]; hi = (int)r[1]; }
            if (KIND == 4) lo = __builtin_amdgcn_readlane(lo, 17) + (int)threadIdx.x;            // readlane -> SGPR -> VALU
            if (KIND == 5) a = pad[(__double2loint(a) & 1023) + 64];                              // dependent LDS read (b64)
            if (KIND == 6) a = __builtin_amdgcn_rcp(a) + c;                                       // v_rcp_f64 + add
            if (KIND == 7) { a = fma(a, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }  // 4 independent chains
            if (KIND == 8 || KIND == 9) { a = fma(a, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }
        }
        // one store per 64 FMAs: KIND 8 all lanes to the SAME address (the front-end's soft-log store), KIND 9 lane 0 only
        if (KIND == 8) out[8192 + blockIdx.x * 4096 + (i & 4095)] = a;
        if (KIND == 9 && threadIdx.x == 0) out[8192 + blockIdx.x * 4096 + (i & 4095)] = a;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = a + lo + hi + a2 + a3 + a4;
}

template <int KIND>
void run(const char* name, double* d) {
    const int n = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    printf("%-34s", name);
    for (int g : {256, 512, 1024, 2048}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            k_chain<KIND><<<g, 64>>>(n, d);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("  %4d groups: %6.2f", g, ms * 1e-3 * 2.4e9 / (16.0 * n));
    }
    printf("   (cycles per step per wave)\n");
}

int main() {
    double* d;
    hipMalloc(&d, (8192 + 2048 * 4096) * 8);
    run<0>("v_fma_f64 (dependent)", d);
    run<1>("v_mov_dpp row_ror + v_add", d);
    run<2>("v_permlane32_swap", d);
    run<3>("v_permlane16_swap", d);
    run<4>("v_readlane + v_add", d);
    run<5>("ds_read_b64 (dependent address)", d);
    run<6>("v_rcp_f64 + v_add_f64", d);
    run<7>("4 independent v_fma_f64 (per 4)", d);
    run<8>("same + 64-lane same-address store", d);
    run<9>("same + one-lane store", d);
    return 0;
}
