// k_channel.hip — device-side channel tool for synthetic multi-stream workloads
// (the reference has no channel model). Element-wise, HBM-bound:
// 4 B in + 4 B out per sample, 16-byte accesses (4 samples per lane).
//   out[n] = clip(rint(gain * in[n] * exp(j 2 pi f0 n / Fs) + sigma * (N(0,1) + j N(0,1))))
// Noise is counter-based (keyed by seed and the sample index) so a stream is reproducible
// on any device and independent of the launch geometry.
// k_resample_clock adds the sample-clock error of SURVEY.md §8f-2: the capture as an ADC running
// `ppm` parts per million fast would have taken it (linear interpolation at n (1 + ppm 1e-6), round to
// nearest) - what makes the timing loop drift through sample boundaries and move the chunk grid.
#include <hip/hip_runtime.h>
#include <math.h>

#include "opv_device.h"

namespace {

__device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline int pack_iq(double re, double im) {
    re = fmin(fmax(rint(re), -32768.0), 32767.0);
    im = fmin(fmax(rint(im), -32768.0), 32767.0);
    return ((int)re & 0xFFFF) | ((int)im << 16);
}

}  // namespace

extern "C" __global__ __launch_bounds__(256) void k_channel(const int4* __restrict__ in, int4* __restrict__ out,
                                                             uint64_t n_quads, double gain, double f0_over_fs,
                                                             double sigma, uint64_t seed) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads;
         q += (uint64_t)gridDim.x * blockDim.x) {
        const int4 v = in[q];
        const int w[4] = {v.x, v.y, v.z, v.w};
        int r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t n = 4 * q + k;
            double t = f0_over_fs * (double)n;
            t -= rint(t);  // cycles, |t| <= 0.5
            double sn, cs;
            sincospi(2.0 * t, &sn, &cs);
            const double xr = gain * (double)(int)(short)(w[k] & 0xFFFF);
            const double xi = gain * (double)(w[k] >> 16);
            double yr = xr * cs - xi * sn, yi = xr * sn + xi * cs;
            if (sigma > 0.0) {
                const uint64_t h = mix64(seed ^ (n * 0xD1342543DE82EF95ull));
                const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
                const float u2 = ((float)(uint32_t)((h >> 8) & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);
                const float rad = sqrtf(-2.0f * logf(u1));
                float s2, c2;
                sincospif(2.0f * u2, &s2, &c2);
                yr += sigma * (double)(rad * c2);
                yi += sigma * (double)(rad * s2);
            }
            r[k] = pack_iq(yr, yi);
        }
        out[q] = make_int4(r[0], r[1], r[2], r[3]);
    }
}

// out[n] = rint(lerp(in, n * rate)), rate = 1 + ppm 1e-6, index clamped to n_in - 2 (the last output samples).
// Same operations in the same order as the numpy model the parity tests use (tests/oracle_lib.py
// resample_clock; contraction is off for this TU), so the two are bit-identical.
extern "C" __global__ __launch_bounds__(256) void k_resample_clock(const int* __restrict__ in, uint64_t n_in,
                                                                    int* __restrict__ out, uint64_t n_out, double rate) {
    for (uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += (uint64_t)gridDim.x * blockDim.x) {
        const double t = (double)n * rate;
        uint64_t i = (uint64_t)t;
        if (i > n_in - 2) i = n_in - 2;
        const double f = t - (double)i, g = 1.0 - f;
        const int w0 = in[i], w1 = in[i + 1];
        const double re = (double)(int)(short)(w0 & 0xFFFF) * g + (double)(int)(short)(w1 & 0xFFFF) * f;
        const double im = (double)(w0 >> 16) * g + (double)(w1 >> 16) * f;
        out[n] = pack_iq(re, im);
    }
}
