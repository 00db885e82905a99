// rcp_accuracy.hip — how good is v_rcp_f64 on gfx950, and what do one / two Newton steps leave?
// (the front-end shares ONE reciprocal between its two divides and polishes the quotients with a residual step;
// whether the second Newton step on the reciprocal is needed depends on this)
// Build: hipcc -O3 --offload-arch=gfx950 -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ void k(const double* x, double* y0, double* y1, double* y2, double* q1, double* q2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t = x[i], num = x[(i * 7 + 3) % n];
    double y = __builtin_amdgcn_rcp(t);
    y0[i] = y;
    double a = fma(fma(-t, y, 1.0), y, y);
    y1[i] = a;
    double b = fma(fma(-t, a, 1.0), a, a);
    y2[i] = b;
    double r1 = num * a; r1 = fma(fma(-t, r1, num), a, r1); q1[i] = r1;     // quotient with ONE Newton step + residual step
    double r2 = num * b; r2 = fma(fma(-t, r2, num), b, r2); q2[i] = r2;     // with two
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    srand48(11);
    for (int i = 0; i < n; ++i) x[i] = (drand48() + 0.5) * exp((drand48() * 2 - 1) * 40.0) * (i & 1 ? -1 : 1);
    double *dx, *d[5];
    hipMalloc(&dx, n * 8); for (auto& p : d) hipMalloc(&p, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d[0], d[1], d[2], d[3], d[4], n);
    std::vector<double> h[5];
    for (int j = 0; j < 5; ++j) { h[j].resize(n); hipMemcpy(h[j].data(), d[j], n * 8, hipMemcpyDeviceToHost); }
    const char* names[5] = {"v_rcp_f64", "+1 Newton step", "+2 Newton steps", "quotient, 1 step + residual", "quotient, 2 steps + residual"};
    for (int j = 0; j < 5; ++j) {
        long double worst = 0; long inexact = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = j < 3 ? 1.0L / (long double)x[i] : (long double)x[(i * 7 + 3) % n] / (long double)x[i];
            const long double e = fabsl(((long double)h[j][i] - ref) / ref);
            if (e > worst) worst = e;
            if (j >= 3 && h[j][i] != (double)ref) ++inexact;
        }
        printf("%-30s max relative error %.3Le (2^%.1Lf)%s", names[j], worst, log2l(worst), j >= 3 ? "" : "\n");
        if (j >= 3) printf(", not correctly rounded: %ld of %d\n", inexact, n);
    }
    return 0;
}
