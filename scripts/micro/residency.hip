// Diagnostics: how many workgroups (T threads, L bytes of LDS) are really co-resident, and where the dispatcher puts them.
// usage: residency [grid] [ldsBytes] [threads]
// Each workgroup records its start time (100 MHz wall clock), HW_ID and XCC_ID, then spins for ~20 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
extern __shared__ unsigned char lds[];
__global__ void k(unsigned long long* out, int spinTicks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (unsigned char)hw;
    while (wall_clock64() - t0 < (unsigned long long)spinTicks) { __builtin_amdgcn_s_sleep(10); }
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t0; out[blockIdx.x * 3 + 1] = hw; out[blockIdx.x * 3 + 2] = (xcc & 0xF) | ((unsigned long long)lds[0] << 32); }
}
int main(int argc, char** argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 2560, ldsBytes = argc > 2 ? atoi(argv[2]) : 16384, threads = argc > 3 ? atoi(argv[3]) : 64;
    unsigned long long* d; hipMalloc(&d, grid * 24);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, threads, ldsBytes);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), ldsBytes, 0, d, 2000000);
    if (hipGetLastError() != hipSuccess) { printf("launch failed (threads %d, lds %d)\n", threads, ldsBytes); return 1; }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 3); hipMemcpy(h.data(), d, grid * 24, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull; for (int i = 0; i < grid; i++) tmin = std::min(tmin, h[i * 3]);
    int early = 0; std::map<unsigned, int> perCu;
    for (int i = 0; i < grid; i++) {
        const bool e = h[i * 3] - tmin < 1000000;       // started within 10 ms of the first
        early += e;
        const unsigned hw = (unsigned)h[i * 3 + 1], xcc = (unsigned)(h[i * 3 + 2] & 0xF);
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (e) perCu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist; for (auto& kv : perCu) hist[kv.second]++;
    printf("grid %d lds %d threads %d: occupancy API %d/CU; started in the first 10 ms: %d on %zu CUs; per-CU histogram:", grid, ldsBytes, threads, occ, early, perCu.size());
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\n");
    return 0;
}
