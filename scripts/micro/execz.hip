// micro: cost of the compiler's divergent-if skeleton (s_and_saveexec + s_cbranch_execz, not taken) against a select
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(int iters, unsigned long long* out, unsigned seed)
{
    unsigned x = threadIdx.x * 2654435761u + seed, acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) { const unsigned f = __builtin_ctz(x | 0x80000000u) >> 3; acc += ((x >> k) & 1u) ? f + k : 0u; }
            else { if ((x >> k) & 1u) { __asm__ volatile("" : "+v"(acc)); acc += (__builtin_ctz(x | 0x80000000u) >> 3) + k; __asm__ volatile("" : "+v"(acc)); } }
        }
        x = x * 1664525u + 1013904223u;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + (threadIdx.x & 63)] = (t1 - t0) + (acc & 1);
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 8 * 256 * 64);
    static unsigned long long h[256 * 64];
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64), 0, 0, iters, d, 12345u);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, iters, d, 12345u);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i * 64];
        printf("%s: %.1f cycles per step\n", mode == 0 ? "select" : "divergent if (saveexec + cbranch_execz)", s / 256 / (iters * 8.0));
    }
    return 0;
}
