// micro: on which SIMD do the two waves of a 128-thread workgroup land when nine such workgroups (17.5 KiB of LDS each) fill a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
template <int W> __global__ __launch_bounds__(64 * W) void k(uint32_t* out, int spin)
{
    __shared__ uint32_t lds[17520 / 4];
    lds[threadIdx.x] = threadIdx.x;
    uint32_t hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1664525u + 1013904223u;           // keep every workgroup resident while the others arrive
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * W + (threadIdx.x >> 6); out[2 * w] = hw; out[2 * w + 1] = xcc + (a == 7 ? 1 : 0) + lds[5] * 0; }
}
template <int W> void run(uint32_t* d, int wgs)
{
    static uint32_t h[2 * 256 * 32 * 2];
    k<W><<<wgs, 64 * W>>>(d, 2000000); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 8 * wgs * W, hipMemcpyDeviceToHost);
    // per role (wave index in the workgroup): histogram of SIMD ids; and per CU: waves of role 0 per SIMD
    std::map<int, std::map<int, int>> roleSimd;
    std::map<uint64_t, std::map<int, int>> cuRole0;
    for (int g = 0; g < wgs; ++g)
        for (int w = 0; w < W; ++w) {
            const uint32_t hw = h[2 * (g * W + w)], xcc = h[2 * (g * W + w) + 1] & 0xF;
            const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            roleSimd[w][simd]++;
            if (w == 0) cuRole0[((uint64_t)xcc << 16) | (se << 8) | (sh << 4) | cu][simd]++;
        }
    printf("%d waves per workgroup, %d workgroups:\n", W, wgs);
    for (auto& r : roleSimd) { printf("  wave %d of the workgroup: SIMD 0/1/2/3 = %d %d %d %d\n", r.first, r.second[0], r.second[1], r.second[2], r.second[3]); }
    int shown = 0;
    for (auto& c : cuRole0) if (shown++ < 4) printf("  CU %06llx: first waves on SIMD 0/1/2/3 = %d %d %d %d\n", (unsigned long long)c.first, c.second[0], c.second[1], c.second[2], c.second[3]);
    printf("  distinct CUs seen %zu\n", cuRole0.size());
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 8 * 256 * 32 * 2);
    run<2>(d, 256 * 9);
    run<3>(d, 256 * 6);
    run<1>(d, 256 * 9);
    return 0;
}
