// dpp64.hip — what the DP-ALU DPP forms of gfx950 cost for ONE wave on a SIMD (the front-end's regime):
//   v_fmac_f64_dpp dst, src0 row_newbcast:n, src1     dst[l] += src0[row(l) * 16 + n] * src1[l]
//   v_mov_b64_dpp  dst, src0 row_newbcast:n           dst[l]  = src0[row(l) * 16 + n]
// (the only DPP controls 64-bit operations take: llvm-mc, "DP ALU dpp only supports row_newbcast").
// With them a row of 16 lanes can form 16 differently weighted sums of its 16 values in 16 instructions - a
// candidate replacement for the front-end's products + permlane / DPP-rotation reductions + v_readlane hand-outs.
// Prints cycles per pass (s_memtime; a pass of the empty loop costs 36, overlapping partly with the work) and checks the
// semantics on the way. A plain dependent fp64 FMA costs a lone wave 4.6 cycles (lone_wave.hip, 64 per pass).
// Build: hipcc -O3 --offload-arch=gfx950 -o dpp64 dpp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define B16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

// semantics: out[l] = sum_n x[row n] * w_n[l]  with w_n[l] = 1 + n + 100 l
__global__ void k_check(double* out) {
    const int l = threadIdx.x;
    double x = 1.0 + 0.5 * l, acc = 0.0;
#define STEP(N) { double w = 1.0 + N + 100.0 * l; asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w)); }
    B16(STEP)
#undef STEP
    double m;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(acc));
    double m1;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(m1) : "v"(acc));
    out[l] = acc;
    out[64 + l] = m;
    out[128 + l] = m1;
}

// 16 bcast-FMACs into ONE accumulator (dependent chain), REP times
__global__ void k_fmac_dpp_dep(double* out, unsigned long long* cyc, int rep) {
    double x = out[threadIdx.x], w = 1.0000001, acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#define STEP(N) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile(B16(STEP) : "+v"(acc) : "v"(x), "v"(w));
#undef STEP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the same into two alternating accumulators
__global__ void k_fmac_dpp_2acc(double* out, unsigned long long* cyc, int rep) {
    double x = out[threadIdx.x], w = 1.0000001, a0 = 0.0, a1 = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#define STEP(N) "v_fmac_f64_dpp %0, %2, %3 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %3 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile(STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) : "+v"(a0), "+v"(a1) : "v"(x), "v"(w));
#undef STEP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// plain dependent v_fmac_f64 (same encoding family, no DPP) for comparison
__global__ void k_fmac_plain(double* out, unsigned long long* cyc, int rep) {
    double x = out[threadIdx.x], w = 1.0000001, acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#define STEP(N) "v_fmac_f64 %0, %1, %2\n\t"
        asm volatile(B16(STEP) : "+v"(acc) : "v"(x), "v"(w));
#undef STEP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// 16 bcast moves (independent)
__global__ void k_mov_dpp(double* out, unsigned long long* cyc, int rep) {
    double x = out[threadIdx.x], m = 0.0, s = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#define STEP(N) "v_mov_b64_dpp %0, %1 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile(B16(STEP) : "=&v"(m) : "v"(x));
#undef STEP
        s += m;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// producer -> DPP consumer: the source is written by a VALU instruction right before each bcast-FMAC (hazard cost)
__global__ void k_fmac_dpp_fresh(double* out, unsigned long long* cyc, int rep) {
    double x = out[threadIdx.x], w = 1.0000001, acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#define STEP(N) "v_add_f64 %1, %1, %2\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile(B16(STEP) : "+v"(acc), "+v"(x) : "v"(w));
#undef STEP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the all-reduce across rows of one 64-bit value: permlane32 swap of (v, v) + add, permlane16 swap of (v, v) + add
__device__ inline double rows_step32(double v) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ inline double rows_step16(double v) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__global__ void k_rows_allreduce(double* out, unsigned long long* cyc, int rep) {
    double v = out[threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v = rows_step16(rows_step32(v)) * 0.25;
            asm volatile("" : "+v"(v));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the loop itself (counter, compare, branch), subtracted from every figure below
__global__ void k_empty(double* out, unsigned long long* cyc, int rep) {
    double x = out[threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) asm volatile("" : "+v"(x));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static double g_loop = 0.0, g_plain = 0.0;   // cycles per pass of the empty loop / of 16 plain dependent FMACs
static bool k_is_plain = false;

template <typename K>
static void run(const char* name, K k, int per_rep, int rep = 4000) {
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, 128 * 8)); CK(hipMalloc(&c, 8));
    std::vector<double> h(128, 1.0);
    CK(hipMemcpy(d, h.data(), 128 * 8, hipMemcpyHostToDevice));
    k<<<1, 64>>>(d, c, rep);
    k<<<1, 64>>>(d, c, rep);
    CK(hipDeviceSynchronize());
    unsigned long long cy; CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
    if (per_rep == 0) { g_loop = (double)cy / rep; printf("%-28s %7.2f cycles per pass\n", name, g_loop); }
    else printf("%-28s %7.1f cycles per pass of %d instructions (+%.2f per instruction over the same number of plain dependent v_fmac_f64)\n",
                name, (double)cy / rep, per_rep, g_plain > 0 ? ((double)cy / rep - g_plain * per_rep / 16.0) / per_rep : 0.0);
    if (k_is_plain) g_plain = (double)cy / rep;
    CK(hipFree(d)); CK(hipFree(c));
}

int main() {
    double* d; CK(hipMalloc(&d, 192 * 8));
    k_check<<<1, 64>>>(d);
    std::vector<double> h(192);
    CK(hipMemcpy(h.data(), d, 192 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        double ref = 0; const int row = l / 16;
        for (int n = 0; n < 16; ++n) ref += (1.0 + 0.5 * (row * 16 + n)) * (1.0 + n + 100.0 * l);
        if (std::fabs(ref - h[l]) > 1e-9 * std::fabs(ref)) ++bad;
    }
    for (int l = 0; l < 64; ++l) if (h[64 + l] != h[(l / 16) * 16 + 5]) ++bad;
    int bc = 0;
    for (int l = 0; l < 64; ++l) if (h[128 + l] != h[(l / 16) * 16 + 5]) ++bc;
    printf("v_mov_b64_dpp row_newbcast with bound_ctrl:1: %s (%d lanes differ from the broadcast)\n", bc ? "NOT a broadcast" : "same as without", bc);
    printf("semantics: %s\n", bad ? "MISMATCH" : "ok (fmac: dst += src0[row lane n] * src1[own lane]; mov: dst = src0[row lane n])");
    run("empty loop", k_empty, 0);
    k_is_plain = true; run("v_fmac_f64 plain, dependent", k_fmac_plain, 16); k_is_plain = false;
    run("v_fmac_f64_dpp, dependent", k_fmac_dpp_dep, 16);
    run("v_fmac_f64_dpp, 2 accum.", k_fmac_dpp_2acc, 16);
    run("v_mov_b64_dpp", k_mov_dpp, 16);
    run("add + s_nop 1 + fmac_dpp", k_fmac_dpp_fresh, 48);
    run("4 x (rows all-reduce + ldexp)", k_rows_allreduce, 52);   // 2 x (2 movs, s_nop, 2 swaps, add) + v_ldexp each
    return bad;
}
