// micro: cycles per hop of the scalar walk (v_readlane -> s_cmp -> s_cselect -> v_readlane ...), one wave alone and ten per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_hop(int iters, unsigned long long* out, int variant)
{
    const int lane = threadIdx.x & 63;
    int nh = (lane * 7 + 3) & 63;            // a permutation-ish successor table
    unsigned long long mm = 0; int w = 1;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (variant == 0) {
        for (int i = 0; i < iters; ++i) {
            for (int u = 0; u < 4; ++u) {
                const int n1 = __builtin_amdgcn_readlane(nh, w & 63);
                mm |= 1ull << (w & 63);
                w = (w < 64) ? n1 : 64;
            }
        }
    } else if (variant == 1) {               // no clamp: pure readlane chain
        for (int i = 0; i < iters; ++i) {
            for (int u = 0; u < 4; ++u) {
                const int n1 = __builtin_amdgcn_readlane(nh, w);
                mm |= 1ull << (w & 63);
                w = n1;
            }
        }
    } else {                                  // scalar-only chain of 5 dependent SALU ops per hop
        unsigned x = (unsigned)w;
        for (int i = 0; i < iters; ++i) {
            for (int u = 0; u < 4; ++u) {
                x = x * 5u + 1u; x ^= x >> 3; x += 7u; x = (x << 1) | (x >> 31); x ^= 0x55u;
                x = __builtin_amdgcn_readfirstlane(x);
            }
        }
        w = (int)x;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = (t1 - t0) + (mm & 1) + (w & 1); }
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    unsigned long long h[4096];
    const int iters = 20000;
    for (int variant = 0; variant < 3; ++variant)
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int blocks = cfg == 0 ? 1 : 256, threads = cfg == 2 ? 640 : 64;
            hipLaunchKernelGGL(k_hop, dim3(blocks), dim3(threads), 0, 0, iters, d, variant);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 8 * blocks * (threads / 64), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < blocks * (threads / 64); ++i) s += (double)h[i];
            printf("variant %d, %d workgroups x %d waves: %.1f cycles per hop\n", variant, blocks, threads / 64, s / (blocks * (threads / 64)) / (iters * 4.0));
        }
    return 0;
}
