// Feasibility probe for a speculative helper-wave front-end ("what would move the 64-stream number"):
// one 256-thread workgroup per stream = four waves on the four SIMDs of a CU.
//   wave 0 "main"   : per symbol, waits for the helpers' window sums (LDS flag), reads 4 x 16 B per lane, combines them
//                     (8 FMA), broadcasts 16 values by v_readlane, then runs a serial fp64 chain standing in for the loop
//                     filters / atan2 (TAIL dependent FMAs), and publishes the new position.
//   wave 1 "taps"   : two symbols ahead: reads the published position, forms 2 x 60 rotated samples (~30 fp64 ops),
//                     writes them to LDS, raises a flag.
//   waves 2, 3 "sum": each lane owns one output (type x order x alignment): 40 taps x (2 LDS reads + 1 FMA), writes its
//                     sum, raises a flag.
// All hand-overs are LDS words polled by the consumer (no s_barrier: the main wave must never wait for a barrier).
// Output: cycles per symbol of the main wave (s_memtime), i.e. what such a kernel could reach, against 1052 today.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef TAIL
#define TAIL 96
#endif
#define DEPTH 4

__global__ __launch_bounds__(256) void k_pipe(double* out, unsigned long long* cyc, int nsym) {
    __shared__ double zbuf[DEPTH][128][2];        // rotated samples of a symbol (two alignments)
    __shared__ double ubuf[DEPTH][128];           // window sums
    __shared__ double coef[40][64];
    __shared__ volatile int f_pos, f_taps, f_sum[2];
    __shared__ double s_pos;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < 40 * 64; i += 256) (&coef[0][0])[i] = 1.0 / (1.0 + i);
    if (threadIdx.x == 0) { f_pos = 1; f_taps = -1; f_sum[0] = f_sum[1] = -1; s_pos = 0.25; }
    __syncthreads();
    if (wave == 0) {
        double acc = 1.0, pos = 0.25;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < nsym; ++k) {
            while (f_sum[0] < k || f_sum[1] < k) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const double* u = &ubuf[k % DEPTH][0];
            double a0 = u[lane], a1 = u[64 + lane], b0 = u[(lane + 1) & 63], b1 = u[64 + ((lane + 1) & 63)];
            const double d = acc * 1e-9, f = pos - floor(pos);
            double v0 = fma(fma(fma(a1, d, a0), d, a1), d, a0), v1 = fma(fma(fma(b1, d, b0), d, b1), d, b0);
            double v = fma(f, v1 - v0, v0);
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), r * 4), __builtin_amdgcn_readlane(__double2loint(v), r * 4));
                s += x;
            }
            double t = s * 1e-3 + acc;
#pragma unroll
            for (int r = 0; r < TAIL; ++r) t = fma(t, 0.999999, 1e-7);   // serial chain: the loop filters' share
            acc = t;
            pos += 40.0 + 1e-4 * (t - floor(t));
            if (lane == 0) { s_pos = pos; f_pos = k + 2; }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[blockIdx.x] = t1 - t0; out[blockIdx.x] = acc; }
    } else if (wave == 1) {
        for (int k = 0; k < nsym; ++k) {
            while (f_pos < k) __builtin_amdgcn_s_sleep(1);       // position published two symbols back is enough
            const double p = s_pos;
            double x = p * (lane + 1), c = 1.0, sn = x;
#pragma unroll
            for (int r = 0; r < 13; ++r) { c = fma(c, x * 1e-3, 0.5); sn = fma(sn, x * 1e-3, c); }   // the exp(j m d) polynomial's share
            const double zr = c * x - sn, zi = sn * x + c, yr = c * (x + 1) - sn, yi = sn * (x + 1) + c;
            zbuf[k % DEPTH][lane][0] = zr; zbuf[k % DEPTH][lane][1] = zi;
            zbuf[k % DEPTH][64 + lane][0] = yr; zbuf[k % DEPTH][64 + lane][1] = yi;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0) f_taps = k;
        }
    } else {
        const int h = wave - 2;
        for (int k = 0; k < nsym; ++k) {
            while (f_taps < k) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const double* z = &zbuf[k % DEPTH][0][0];
            double a = 0.0;
            const int base = (lane >> 2) + 64 * h;                // which of the 128 rotated samples the output starts at
#pragma unroll
            for (int j = 0; j < 40; ++j) a = fma(z[2 * ((base + j) & 127) + (lane & 1)], coef[j][lane], a);
            ubuf[k % DEPTH][64 * h + lane] = a;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0) f_sum[h] = k;
        }
    }
}

int main(int argc, char** argv) {
    const int nsym = argc > 1 ? atoi(argv[1]) : 20000;
    for (int nwg : {1, 64, 256}) {
        double* d_out; unsigned long long* d_cyc;
        hipMalloc(&d_out, nwg * 8); hipMalloc(&d_cyc, nwg * 8);
        k_pipe<<<nwg, 256>>>(d_out, d_cyc, nsym);
        k_pipe<<<nwg, 256>>>(d_out, d_cyc, nsym);
        std::vector<unsigned long long> c(nwg);
        hipMemcpy(c.data(), d_cyc, nwg * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0, mn = ~0ull;
        for (auto v : c) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        printf("TAIL=%d workgroups=%d: main wave %.1f .. %.1f cycles per symbol\n", TAIL, nwg, (double)mn / nsym, (double)mx / nsym);
        hipFree(d_out); hipFree(d_cyc);
    }
    return 0;
}
