// Can the fp64 matrix pipe sum a DPP row (16 lanes) faster than four DPP rotations? (dev tool)
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products, one per block of 16 lanes; one f64 per lane for
// A, B and C/D. Part 1 prints the lane maps found with integer data; part 2 times a dependent chain of
// (two MFMAs = one 16-lane all-reduce) against the four-step DPP all-reduce, one wave alone on a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_map(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

__device__ inline double dpp_row_sum(double v) {
    auto step = [&](auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
        v += __hiloint2double(hi, lo);
    };
    step(std::integral_constant<int, 0x128>{});
    step(std::integral_constant<int, 0x124>{});
    step(std::integral_constant<int, 0x122>{});
    step(std::integral_constant<int, 0x121>{});
    return v;
}
__device__ inline double mfma_row_sum(double v) {
    const double p = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);   // sums over k
    return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, p, 0.0, 0, 0, 0);             // sums over the rest
}

template <int MODE>
__global__ __launch_bounds__(64) void k_time(int n, double* out, long long* cyc) {
    double v = 1.0 + threadIdx.x * 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) v = dpp_row_sum(v) * 0.0625;
        else v = mfma_row_sum(v) * 0.0625;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    double *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    std::vector<double> ha(64), hb(64), hd(64);
    // A = one-hot at lane la, B = all ones  ->  D lanes that light up tell which (i) row lane la feeds
    printf("A lane -> D lanes that receive it (B = ones):\n");
    for (int la = 0; la < 16; ++la) {
        for (int l = 0; l < 64; ++l) { ha[l] = (l == la) ? 1.0 : 0.0; hb[l] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        k_map<<<1, 64>>>(a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("  A lane %2d ->", la);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    printf("B lane -> D lanes that receive it (A = ones):\n");
    for (int lb = 0; lb < 16; ++lb) {
        for (int l = 0; l < 64; ++l) { hb[l] = (l == lb) ? 1.0 : 0.0; ha[l] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        k_map<<<1, 64>>>(a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("  B lane %2d ->", lb);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    // all-reduce check: lane values 1..64 -> every lane of a 16-block must hold its block's sum
    double* out; long long* cyc;
    hipMalloc(&out, 512); hipMalloc(&cyc, 16);
    const int n = 200000;
    k_time<0><<<1, 64>>>(n, out, cyc);
    k_time<1><<<1, 64>>>(n, out, cyc);
    long long hc[2];
    hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), out, 512, hipMemcpyDeviceToHost);
    printf("dependent all-reduce + 1 mul: DPP %.1f, MFMA pair %.1f clock64 ticks per iteration (100 MHz ticks x24 = cycles @2.4GHz: %.0f vs %.0f)\n",
           (double)hc[0] / n, (double)hc[1] / n, (double)hc[0] / n * 24, (double)hc[1] / n * 24);
    printf("mfma result lanes 0,5,17,63: %.12g %.12g %.12g %.12g\n", hd[0], hd[5], hd[17], hd[63]);
    return 0;
}
