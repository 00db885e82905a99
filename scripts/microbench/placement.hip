// Where do single-wave workgroups land? (dev tool) For grid sizes 512 / 1024 / 2048 with the front-end's
// LDS footprint, every workgroup records XCC / SE / CU / SIMD from the hardware id registers while all of
// them are resident; the host prints the histogram of waves per CU and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(64) void k_where(unsigned* out, long long spin) {
    __shared__ unsigned char lds[19040];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + lds[5] * 0; }
}

int main(int argc, char** argv) {
    int pad = argc > 1 ? atoi(argv[1]) : 0;
    for (int n : {256, 512, 1024, 2048}) {
        unsigned* d;
        hipMalloc(&d, n * 8);
        k_where<<<n, 64, pad>>>(d, 20000000 / 100);  // 100 MHz clock: ~2 ms
        std::vector<unsigned> h(2 * n);
        hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_cu, per_simd;
        for (int i = 0; i < n; ++i) {
            unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
            unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey]++;
            per_simd[(cukey << 2) | simd]++;
        }
        std::map<int, int> hc, hs;
        for (auto& kv : per_cu) hc[kv.second]++;
        for (auto& kv : per_simd) hs[kv.second]++;
        printf("grid %d pad %d: CUs used %zu, SIMDs used %zu | waves/CU histogram:", n, pad, per_cu.size(), per_simd.size());
        for (auto& kv : hc) printf(" %dx%d", kv.second, kv.first);
        printf(" | waves/SIMD histogram:");
        for (auto& kv : hs) printf(" %dx%d", kv.second, kv.first);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
