// lone_wave.hip — issue/latency figures of ONE wavefront on a gfx950 SIMD, the regime the MSK
// front-end lives in (one wave per stream, serial over symbols). Dev tool: build with
//   hipcc -O3 --offload-arch=gfx950 -o lone_wave lone_wave.hip && ./lone_wave
// Each test runs REP x UNROLL copies of a small instruction pattern in a single wave and reports
// nanoseconds and shader cycles (at the measured clock) per pattern instance.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REP = 2000;

#define U4(x) x x x x
#define U16(x) U4(x) U4(x) U4(x) U4(x)
#define U64(x) U16(x) U16(x) U16(x) U16(x)

__global__ void k_empty(double* out, int rep) {
    double a = out[threadIdx.x];
    for (int r = 0; r < rep; ++r) asm volatile("" : "+v"(a));
    out[threadIdx.x] = a;
}
// 64 dependent fp64 FMAs
__global__ void k_fma_dep(double* out, int rep) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); }
    out[threadIdx.x] = a;
}
// 64 FMAs, 4 independent chains
__global__ void k_fma_ind4(double* out, int rep) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    out[threadIdx.x] = a0 + a1 + a2 + a3;
}
// 64 FMAs, 2 independent chains
__global__ void k_fma_ind2(double* out, int rep) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, b = 1.0000001, c = 1e-9;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("v_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\nv_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\n")
                     : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
    }
    out[threadIdx.x] = a0 + a1;
}
__global__ void k_add_dep(double* out, int rep) {
    double a = out[threadIdx.x], c = 1e-9;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(c)); }
    out[threadIdx.x] = a;
}
__global__ void k_mul_dep(double* out, int rep) {
    double a = out[threadIdx.x], c = 1.0000001;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(c)); }
    out[threadIdx.x] = a;
}
// fp32 dependent chain for comparison
__global__ void k_fma32_dep(double* out, int rep) {
    float a = (float)out[threadIdx.x], b = 1.0000001f, c = 1e-9f;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); }
    out[threadIdx.x] = a;
}
// integer dependent chain
__global__ void k_iadd_dep(double* out, int rep) {
    int a = (int)out[threadIdx.x];
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_add_u32 %0, %0, 1\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
// rcp_f64 dependent
__global__ void k_rcp_dep(double* out, int rep) {
    double a = out[threadIdx.x] + 1.5;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_rcp_f64 %0, %0\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
// rcp_f64 independent pairs
__global__ void k_rcp_ind(double* out, int rep) {
    double a = out[threadIdx.x] + 1.5, b, c, d, e;
    for (int r = 0; r < rep; ++r) { asm volatile(U4("v_rcp_f64 %1, %0\nv_rcp_f64 %2, %0\nv_rcp_f64 %3, %0\nv_rcp_f64 %4, %0\n") : "+v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e)); }
    out[threadIdx.x] = a + b + c + d + e;
}
__device__ inline int dlo(double v) { return __double2loint(v); }
__device__ inline int dhi(double v) { return __double2hiint(v); }
__device__ inline double mkd(int hi, int lo) { return __hiloint2double(hi, lo); }
template <int CTRL>
__device__ inline double dpp_add(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(dlo(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(dhi(v), CTRL, 0xF, 0xF, true);
    return v + mkd(hi, lo);
}
__device__ inline double swap32_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)dlo(a), (unsigned)dlo(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)dhi(a), (unsigned)dhi(b), false, false);
    return mkd((int)hi[0], (int)lo[0]) + mkd((int)hi[1], (int)lo[1]);
}
// DPP step of the row all-sum: 2 mov_dpp + add, dependent (compiler-scheduled, 16 per rep)
__global__ void k_dpp_step(double* out, int rep) {
    double a = out[threadIdx.x];
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a = dpp_add<0x128>(a); a = dpp_add<0x124>(a); a = dpp_add<0x122>(a); a = dpp_add<0x121>(a); }
        asm volatile("" : "+v"(a));
    }
    out[threadIdx.x] = a;
}
// three interleaved chains as in the kernel
__global__ void k_dpp_step3(double* out, int rep) {
    double a = out[threadIdx.x], b = a + 1, c = a + 2;
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a = dpp_add<0x128>(a); b = dpp_add<0x128>(b); c = dpp_add<0x128>(c);
            a = dpp_add<0x124>(a); b = dpp_add<0x124>(b); c = dpp_add<0x124>(c);
            a = dpp_add<0x122>(a); b = dpp_add<0x122>(b); c = dpp_add<0x122>(c);
            a = dpp_add<0x121>(a); b = dpp_add<0x121>(b); c = dpp_add<0x121>(c);
        }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    }
    out[threadIdx.x] = a + b + c;
}
// permlane32 swap pair + add (dependent)
__global__ void k_swap32(double* out, int rep) {
    double a = out[threadIdx.x], b = a + 1;
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { a = swap32_add(a, b); }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    out[threadIdx.x] = a + b;
}
// LDS: lane0-of-row write, all read 16 B, use (the broadcast round trip)
__global__ void k_lds_bcast(double* out, int rep) {
    __shared__ double red[16];
    double a = out[threadIdx.x];
    const unsigned waddr = (threadIdx.x >> 4) * 8;
    for (int r = 0; r < rep; ++r) {
        double x, y;
        asm volatile(U16("ds_write_b64 %3, %0\nds_read_b64 %1, %4\nds_read_b64 %2, %4 offset:8\ns_waitcnt lgkmcnt(0)\nv_add_f64 %0, %0, %2\n")
                     : "+v"(a), "=&v"(x), "=&v"(y) : "v"(waddr), "v"(0u) : "memory");
        (void)x;
    }
    out[threadIdx.x] = a + red[0];
}
// LDS pointer chase: ds_read_b32 dependent
__global__ void k_lds_chase(double* out, int rep) {
    __shared__ unsigned tab[64];
    tab[threadIdx.x] = 0;
    __syncthreads();
    unsigned p = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)tab;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\nv_add_u32 %0, %0, %1\n") : "+v"(p) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)tab) : "memory");
    }
    out[threadIdx.x] = p;
}
// ds_read2_b32 chase (the tap fetch)
__global__ void k_lds_read2(double* out, int rep) {
    __shared__ unsigned tab[128];
    tab[threadIdx.x] = 0; tab[threadIdx.x + 64] = 0;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)tab;
    unsigned p = base;
    for (int r = 0; r < rep; ++r) {
        unsigned long long w;
        asm volatile(U16("ds_read2_b32 %1, %0 offset0:1 offset1:2\ns_waitcnt lgkmcnt(0)\nv_add_u32 %0, %0, %2\nv_and_b32 %0, 0xffff, %0\n") : "+v"(p), "=&v"(w) : "v"(0u) : "memory");
    }
    out[threadIdx.x] = p;
}
// readlane -> SGPR -> VALU consumer, dependent
__global__ void k_readlane(double* out, int rep) {
    int a = (int)out[threadIdx.x];
    for (int r = 0; r < rep; ++r) {
        int s;
        asm volatile(U16("v_readlane_b32 %1, %0, 50\ns_nop 3\nv_add_u32 %0, %0, %1\n") : "+v"(a), "=&s"(s));
    }
    out[threadIdx.x] = a;
}
// 24 independent readlanes then one consumer (broadcast alternative)
__global__ void k_readlane24(double* out, int rep) {
    int a = (int)out[threadIdx.x];
    for (int r = 0; r < rep; ++r) {
        int s0, s1, s2, s3;
        asm volatile(U4("v_readlane_b32 %1, %0, 0\nv_readlane_b32 %2, %0, 16\nv_readlane_b32 %3, %0, 32\nv_readlane_b32 %4, %0, 48\n"
                        "v_readlane_b32 %1, %0, 1\nv_readlane_b32 %2, %0, 17\n")
                     "s_nop 3\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %2\nv_add_u32 %0, %0, %3\nv_add_u32 %0, %0, %4\n"
                     : "+v"(a), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3));
    }
    out[threadIdx.x] = a;
}
// readfirstlane -> salu -> valu
__global__ void k_rfl_salu(double* out, int rep) {
    int a = (int)out[threadIdx.x];
    for (int r = 0; r < rep; ++r) {
        int s;
        asm volatile(U16("v_readfirstlane_b32 %1, %0\ns_add_i32 %1, %1, 1\ns_nop 0\nv_add_u32 %0, %0, %1\n") : "+v"(a), "=&s"(s) : : "scc");
    }
    out[threadIdx.x] = a;
}
// v_cmp_f64 -> vcc -> cndmask -> (dependent) cvt back
__global__ void k_cmp_sel(double* out, int rep) {
    double a = out[threadIdx.x];
    const double b = 0.5, c = 0.25;
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { a = (a < b) ? a + c : a - c; asm volatile("" : "+v"(a)); }
    }
    out[threadIdx.x] = a;
}
// salu chain
__global__ void k_salu(double* out, int rep) {
    int s = rep;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("s_add_i32 %0, %0, 1\n") : "+s"(s) : : "scc"); }
    out[threadIdx.x] = s;
}
// cvt chain: v_cvt_f64_i32 / v_cvt_i32_f64 (quarter rate?)
__global__ void k_cvt(double* out, int rep) {
    double a = out[threadIdx.x];
    int i;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_cvt_i32_f64 %1, %0\nv_cvt_f64_i32 %0, %1\n") : "+v"(a), "=&v"(i)); }
    out[threadIdx.x] = a;
}
__global__ void k_floor(double* out, int rep) {
    double a = out[threadIdx.x];
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_floor_f64 %0, %0\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
__global__ void k_minmax(double* out, int rep) {
    double a = out[threadIdx.x], b = 2.0;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_max_f64 %0, %0, %1\n") : "+v"(a) : "v"(b)); }
    out[threadIdx.x] = a;
}
// global store of one double per iteration (flat vs global), followed by an LDS read + wait
__global__ void k_flat_store_lds(double* out, double* sink, int rep) {
    __shared__ unsigned tab[64];
    tab[threadIdx.x] = 0;
    __syncthreads();
    unsigned p = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)tab;
    const unsigned base = p;
    double v = 1.0;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("flat_store_dwordx2 %2, %3\nds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\nv_add_u32 %0, %0, %1\n") : "+v"(p) : "v"(base), "v"(sink), "v"(v) : "memory");
    }
    out[threadIdx.x] = p;
}
__global__ void k_global_store_lds(double* out, double* sink, int rep) {
    __shared__ unsigned tab[64];
    tab[threadIdx.x] = 0;
    __syncthreads();
    unsigned p = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)tab;
    const unsigned base = p;
    double v = 1.0;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("global_store_dwordx2 %2, %3, off\nds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\nv_add_u32 %0, %0, %1\n") : "+v"(p) : "v"(base), "v"(sink), "v"(v) : "memory");
    }
    out[threadIdx.x] = p;
}


// ---- second batch: the instruction kinds of the front-end loop, one by one ----------------------
__global__ void k_swap32_ind(double* out, int rep) {
    int a = (int)out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
    out[threadIdx.x] = a + b + c + d;
}
__global__ void k_swap16_ind(double* out, int rep) {
    int a = (int)out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
    out[threadIdx.x] = a + b + c + d;
}
__global__ void k_dppmov_ind(double* out, int rep) {
    int a = (int)out[threadIdx.x], b, c, d, e;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("v_mov_b32_dpp %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %2, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %3, %0 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %4, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                     : "+v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e));
    }
    out[threadIdx.x] = a + b + c + d + e;
}
__global__ void k_mul_lo(double* out, int rep) {
    int a = (int)out[threadIdx.x] + 3;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_mul_lo_u32 %0, %0, %0\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
__global__ void k_mul_u24(double* out, int rep) {
    int a = (int)out[threadIdx.x] + 3;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_mul_u32_u24 %0, %0, %0\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
__global__ void k_ldexp(double* out, int rep) {
    double a = out[threadIdx.x];
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_ldexp_f64 %0, %0, 1\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
__global__ void k_fract(double* out, int rep) {
    double a = out[threadIdx.x];
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_fract_f64 %0, %0\n") : "+v"(a)); }
    out[threadIdx.x] = a;
}
__global__ void k_cvt_i32_f64(double* out, int rep) {
    double a = out[threadIdx.x];
    int i0, i1, i2, i3;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_cvt_i32_f64 %1, %0\nv_cvt_i32_f64 %2, %0\nv_cvt_i32_f64 %3, %0\nv_cvt_i32_f64 %4, %0\n") : "+v"(a), "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3)); }
    out[threadIdx.x] = a + i0 + i1 + i2 + i3;
}
__global__ void k_cvt_f64_i32(double* out, int rep) {
    int a = (int)out[threadIdx.x];
    double d0, d1, d2, d3;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_cvt_f64_i32 %1, %0\nv_cvt_f64_i32 %2, %0\nv_cvt_f64_i32 %3, %0\nv_cvt_f64_i32 %4, %0\n") : "+v"(a), "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)); }
    out[threadIdx.x] = a + d0 + d1 + d2 + d3;
}
__global__ void k_mov_b64(double* out, int rep) {
    double a = out[threadIdx.x], b;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("v_mov_b64 %1, %0\nv_mov_b64 %0, %1\nv_mov_b64 %1, %0\nv_mov_b64 %0, %1\n") : "+v"(a), "=&v"(b)); }
    out[threadIdx.x] = a + b;
}
__global__ void k_cmp_cnd2(double* out, int rep) {
    double a = out[threadIdx.x], b = 0.5;
    int x = 1, y = 2;
    unsigned long long m;
    for (int r = 0; r < rep; ++r) {
        asm volatile(U16("v_cmp_lt_f64 %3, %0, %1\nv_cndmask_b32 %2, %2, %4, %3\nv_cndmask_b32 %4, %4, %2, %3\n") : "+v"(a), "+v"(b), "+v"(x), "=&s"(m), "+v"(y));
    }
    out[threadIdx.x] = a + x + y;
}
// the broadcast round trip of the kernel: 2 writes, 6 x 16-byte reads, wait, dependent add
__global__ void k_red_trip(double* out, int rep) {
    __shared__ __attribute__((aligned(16))) double red[16];
    double a = out[threadIdx.x];
    const unsigned waddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)red + (threadIdx.x >> 4) * 24;
    const unsigned raddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)red;
    for (int r = 0; r < rep; ++r) {
        double x0, x1, x2, x3, x4, x5, x6;
        asm volatile(U16("ds_write2_b64 %7, %0, %0 offset1:1\nds_write_b64 %7, %0 offset:16\n"
                         "ds_read_b64 %1, %8\nds_read_b64 %2, %8 offset:16\nds_read_b64 %3, %8 offset:32\nds_read_b64 %4, %8 offset:48\n"
                         "ds_read_b64 %5, %8 offset:64\nds_read_b64 %6, %8 offset:80\ns_waitcnt lgkmcnt(0)\nv_add_f64 %0, %0, %6\n")
                     : "+v"(a), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6) : "v"(waddr), "v"(raddr) : "memory");
        (void)x0;
    }
    out[threadIdx.x] = a;
}
// 2 readlanes into SGPRs, then VALU consuming them (the X40 hand-over)
__global__ void k_readlane_use(double* out, int rep) {
    int a = (int)out[threadIdx.x], b = a + 1;
    for (int r = 0; r < rep; ++r) {
        int s0, s1;
        asm volatile(U16("v_readlane_b32 %2, %0, 50\nv_readlane_b32 %3, %1, 50\ns_nop 0\nv_add_u32 %0, %0, %2\nv_add_u32 %1, %1, %3\n") : "+v"(a), "+v"(b), "=&s"(s0), "=&s"(s1));
    }
    out[threadIdx.x] = a + b;
}
// global store of one double per iteration with nothing else
__global__ void k_gstore(double* out, double* sink, int rep) {
    double v = 1.0;
    unsigned off = 0;
    for (int r = 0; r < rep; ++r) { asm volatile(U16("global_store_dwordx2 %0, %1, %2\n") : : "v"(off), "v"(v), "s"(sink) : "memory"); }
    out[threadIdx.x] = v;
}
// s_waitcnt with nothing outstanding
__global__ void k_waitcnt(double* out, int rep) {
    for (int r = 0; r < rep; ++r) { asm volatile(U64("s_waitcnt lgkmcnt(0)\n") ::: "memory"); }
    out[threadIdx.x] = rep;
}
__global__ void k_snop(double* out, int rep) {
    for (int r = 0; r < rep; ++r) { asm volatile(U64("s_nop 0\n") ::: "memory"); }
    out[threadIdx.x] = rep;
}
// alternating VALU / SALU (does the scalar instruction ride along for free?)
__global__ void k_valu_salu(double* out, int rep) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    int s = rep;
    for (int r = 0; r < rep; ++r) { asm volatile(U64("v_fma_f64 %0, %0, %2, %3\ns_add_i32 %1, %1, 1\n") : "+v"(a), "+s"(s) : "v"(b), "v"(c) : "scc"); }
    out[threadIdx.x] = a + s;
}
struct Test { const char* name; int per_rep; float ms; };

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    double* d; double* sink;
    CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(d, 0, 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    auto run = [&](const char* name, auto launch, int per_rep) {
        launch(10);  // warm
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int t = 0; t < 3; ++t) {
            CK(hipEventRecord(e0)); launch(REP); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double ns = best * 1e6 / ((double)REP * per_rep);
        printf("%-22s %8.3f ns  %7.2f cyc@%.0fMHz per pattern (%d/rep, %.3f ms)\n", name, ns, ns * clk_khz * 1e-6, clk_khz * 1e-3, per_rep, best);
    };
#define L1(k) [&](int rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rep); }
#define L2(k) [&](int rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sink, rep); }
    run("empty loop", L1(k_empty), 1);
    run("fma_f64 dep", L1(k_fma_dep), 64);
    run("fma_f64 2 chains", L1(k_fma_ind2), 64);
    run("fma_f64 4 chains", L1(k_fma_ind4), 64);
    run("add_f64 dep", L1(k_add_dep), 64);
    run("mul_f64 dep", L1(k_mul_dep), 64);
    run("max_f64 dep", L1(k_minmax), 64);
    run("floor_f64 dep", L1(k_floor), 64);
    run("fma_f32 dep", L1(k_fma32_dep), 64);
    run("add_u32 dep", L1(k_iadd_dep), 64);
    run("rcp_f64 dep", L1(k_rcp_dep), 16);
    run("rcp_f64 ind", L1(k_rcp_ind), 16);
    run("cvt i32<->f64 pair", L1(k_cvt), 16);
    run("dpp step (2mov+add)", L1(k_dpp_step), 16);
    run("dpp step x3 interl.", L1(k_dpp_step3), 16);
    run("swap32 x2 + add", L1(k_swap32), 16);
    run("lds write->read->use", L1(k_lds_bcast), 16);
    run("lds read chase", L1(k_lds_chase), 16);
    run("lds read2 chase", L1(k_lds_read2), 16);
    run("readlane->valu", L1(k_readlane), 16);
    run("24 readlane + use", L1(k_readlane24), 1);
    run("rfl->salu->valu", L1(k_rfl_salu), 16);
    run("cmp_f64->cndmask", L1(k_cmp_sel), 16);
    run("s_add dep", L1(k_salu), 64);
    run("flat st + lds chase", L2(k_flat_store_lds), 16);
    run("global st + lds chase", L2(k_global_store_lds), 16);
    run("swap32 x2 indep", L1(k_swap32_ind), 16);
    run("swap16 x2 indep", L1(k_swap16_ind), 16);
    run("mov_dpp x4 indep", L1(k_dppmov_ind), 16);
    run("mul_lo_u32 dep", L1(k_mul_lo), 64);
    run("mul_u32_u24 dep", L1(k_mul_u24), 64);
    run("ldexp_f64 dep", L1(k_ldexp), 64);
    run("fract_f64 dep", L1(k_fract), 64);
    run("cvt_i32_f64 x4", L1(k_cvt_i32_f64), 16);
    run("cvt_f64_i32 x4", L1(k_cvt_f64_i32), 16);
    run("mov_b64 x4 dep", L1(k_mov_b64), 16);
    run("cmp_f64+2 cndmask", L1(k_cmp_cnd2), 16);
    run("red trip 2w+6r+use", L1(k_red_trip), 16);
    run("2 readlane+nop+2add", L1(k_readlane_use), 16);
    run("global store", L2(k_gstore), 16);
    run("s_waitcnt idle", L1(k_waitcnt), 64);
    run("s_nop 0", L1(k_snop), 64);
    run("fma + s_add pair", L1(k_valu_salu), 64);
    return 0;
}
