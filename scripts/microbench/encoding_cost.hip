// encoding_cost.hip — what the ENCODING of a lone wave's instruction costs: the same dependent fp64 FMAC chain as 4-byte
// VOP2 (_e32), as 8-byte VOP3 (_e64) on 8-byte addresses, as 8-byte VOP3 at 4 mod 8, and mixed. Companion of
// loop_align.hip; input to tools/align_vop3.py's choice (widen as few instructions as possible, or freely?).
// Build: hipcc -O3 --offload-arch=gfx950 -o encoding_cost encoding_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define U4(x) x x x x
#define U16(x) U4(x) U4(x) U4(x) U4(x)
#define U32(x) U16(x) U16(x)
#define U64(x) U32(x) U32(x)

template <int KIND>
__global__ void k(double* out, unsigned long long* cyc, int rep) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    int i0 = (int)a, i1 = 0x7fffffff, i2 = 3;
    float f0 = (float)a, f1 = 1.0000001f, f2 = 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        if (KIND == 0) asm volatile(".p2align 3\n" U64("v_fmac_f64_e32 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile(".p2align 3\n" U64("v_fmac_f64_e64 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (KIND == 2) asm volatile(".p2align 3\ns_nop 0\n" U64("v_fmac_f64_e64 %0, %1, %2\n") "s_nop 0\n" : "+v"(a) : "v"(b), "v"(c));
        if (KIND == 3) asm volatile(".p2align 3\n" U32("v_fmac_f64_e32 %0, %1, %2\nv_fmac_f64_e32 %0, %1, %2\nv_fmac_f64_e64 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (KIND == 4) asm volatile(".p2align 3\n" U32("v_fmac_f64_e32 %0, %1, %2\nv_fmac_f64_e64 %0, %1, %2\nv_fmac_f64_e32 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (KIND == 5) asm volatile(".p2align 3\n" U64("v_fma_f64 %0, %1, %2, %0\n") : "+v"(a) : "v"(b), "v"(c));
        if (KIND == 6) asm volatile(".p2align 3\n" U64("v_add_u32_e32 %0, 1, %0\n") : "+v"(i0));
        if (KIND == 7) asm volatile(".p2align 3\n" U64("v_add_u32_e64 %0, 1, %0\n") : "+v"(i0));
        if (KIND == 8) asm volatile(".p2align 3\n" U64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(f0) : "v"(f1), "v"(f2));
        if (KIND == 9) asm volatile(".p2align 3\n" U64("v_bfi_b32 %0, %1, %0, %2\n") : "+v"(i0) : "v"(i1), "v"(i2));
        if (KIND == 10) asm volatile(".p2align 3\n" U64("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\ns_nop 1\n") : "+v"(i0));
        if (KIND == 11) asm volatile(".p2align 3\n" U64("v_max_f64 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (KIND == 12) asm volatile(".p2align 3\n" U64("v_cndmask_b32_e64 %0, %0, %1, vcc\n") : "+v"(i0) : "v"(i1));
        if (KIND == 13) asm volatile(".p2align 3\n" U64("v_readlane_b32 s20, %0, 3\n") : : "v"(i0) : "s20");
        if (KIND == 14) asm volatile(".p2align 3\n" U64("v_cvt_f64_i32_e64 %0, %1\n") : "=v"(a) : "v"(i0));
        if (KIND == 15) asm volatile(".p2align 3\n" U32("v_permlane32_swap_b32_e64 %0, %1\ns_nop 0\n") : "+v"(i0), "+v"(i1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + i0 + f0 + i1;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_pass, double* d, unsigned long long* c) {
    for (int w = 0; w < 3; ++w) k<KIND><<<1, 64>>>(d, c, 2000);
    hipDeviceSynchronize();
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.1f cycles per pass = %.2f per instruction\n", name, (double)cy / 2000, (double)cy / 2000 / per_pass);
}

int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 64 * 8); hipMalloc(&c, 8); hipMemset(d, 0, 64 * 8);
    run<0>("64 x v_fmac_f64_e32 (4 bytes)", 64, d, c);
    run<1>("64 x v_fmac_f64_e64 (8 bytes, aligned)", 64, d, c);
    run<2>("s_nop + 64 x v_fmac_f64_e64 (at 4 mod 8) + s_nop", 66, d, c);
    run<3>("32 x (e32, e32, e64): every e64 aligned", 96, d, c);
    run<4>("32 x (e32, e64, e32): every e64 at 4 mod 8", 96, d, c);
    run<5>("64 x v_fma_f64 (VOP3 only, aligned)", 64, d, c);
    run<6>("64 x v_add_u32_e32 dependent", 64, d, c);
    run<7>("64 x v_add_u32_e64 dependent (aligned)", 64, d, c);
    run<8>("64 x v_fma_f32 dependent (VOP3, aligned)", 64, d, c);
    run<9>("64 x v_bfi_b32 dependent (aligned)", 64, d, c);
    run<10>("64 x (v_mov_b32_dpp row_ror + s_nop 1)", 128, d, c);
    run<11>("64 x v_max_f64 dependent (aligned)", 64, d, c);
    run<12>("64 x v_cndmask_b32_e64 dependent (aligned)", 64, d, c);
    run<13>("64 x v_readlane_b32 (aligned)", 64, d, c);
    run<14>("64 x v_cvt_f64_i32_e64 (aligned)", 64, d, c);
    run<15>("32 x (v_permlane32_swap_b32_e64 + s_nop 0)", 64, d, c);
    return 0;
}
