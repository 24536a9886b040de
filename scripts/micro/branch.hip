// micro: what a uniform branch costs one wave (taken / not taken), against the same work without the branch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ void k_br(int iters, unsigned mask, unsigned long long* out)
{
    unsigned x = mask, acc = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) { acc = acc * 3u + (unsigned)k; }                                    // straight line
            else {
                // a scalar condition the compiler cannot fold: bit k of a rotating register
                if (__builtin_amdgcn_readfirstlane(x >> k) & 1u) { acc = acc * 3u + (unsigned)k; __asm__ volatile("" ::: "memory"); }
                else { __asm__ volatile("s_nop 0" ::: "memory"); }
            }
        }
        x = (x << 1) | (x >> 31);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x] = (t1 - t0) + (acc & 1);
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 8 * 256);
    unsigned long long h[256];
    const int iters = 20000;
    struct { const char* name; unsigned mask; int mode; } cfg[] = { {"straight", 0, 0}, {"all taken (mask ~0)", ~0u, 1}, {"none taken (mask 0)", 0u, 1}, {"alternating", 0x55555555u, 1} };
    for (auto& c : cfg) {
        if (c.mode == 0) hipLaunchKernelGGL(k_br<0>, dim3(256), dim3(64), 0, 0, iters, c.mask, d);
        else hipLaunchKernelGGL(k_br<1>, dim3(256), dim3(64), 0, 0, iters, c.mask, d);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 8 * 256, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("%-24s %.1f cycles per step (8 steps per iteration)\n", c.name, s / 256 / (iters * 8.0));
    }
    return 0;
}
