// micro: what a SIMD sustains as its waves are added -- W waves per workgroup, one workgroup per CU (256), each wave a stream of
// independent (or dependent) v_add_u32 / v_and_or_b32; per-wave cycles (s_memtime), wall time, and where the waves sat (HW_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#define S4(X) X X X X
#define S32(X) S4(X) S4(X) S4(X) S4(X) S4(X) S4(X) S4(X) S4(X)
template <int KIND>
__global__ void k(int iters, uint32_t* out)
{
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9E3779B9u, c = b + 77u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(S32("v_add_u32 %0, %1, %2\n") : "=v"(a) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile(S32("v_add_u32 %0, %1, %0\n") : "+v"(a) : "v"(b));
        if (KIND == 2) asm volatile(S32("v_and_or_b32 %0, %1, %2, %1\n") : "=v"(a) : "v"(b), "v"(c));
        if (KIND == 3) asm volatile(S32("v_add_u32 %0, %1, %2\n s_add_u32 %3, %3, 1\n") : "=v"(a) : "v"(b), "v"(c), "s"(iters) : "scc");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); out[2 * w] = (uint32_t)(t1 - t0); out[2 * w + 1] = hw; }
    if (a == 0x12345u) out[0] = 1;
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 8 * 256 * 16);
    static uint32_t h[2 * 256 * 16];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 100000;
    k<1><<<256, 1024>>>(1000000, d); (void)hipDeviceSynchronize();
    const char* names[] = {"v_add_u32 independent", "v_add_u32 dependent", "v_and_or_b32 independent", "v_add + s_add pairs"};
    for (int kind = 0; kind < 4; ++kind)
        for (int waves : {1, 2, 4, 8, 12, 16}) {
            void (*fn)(int, uint32_t*) = kind == 0 ? k<0> : kind == 1 ? k<1> : kind == 2 ? k<2> : k<3>;
            fn<<<256, waves * 64>>>(1000, d); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); fn<<<256, waves * 64>>>(iters, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, d, 8 * 256 * waves, hipMemcpyDeviceToHost);
            double cyc = 0; std::map<uint32_t, int> perCu, perSimd;
            for (int w = 0; w < 256 * waves; ++w) {
                cyc += h[2 * w];
                const uint32_t hw = h[2 * w + 1];
                // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: [14:13]), ... xcc in XCC_ID
                perCu[(hw >> 8) & 0xFF]++; perSimd[(hw >> 4) & 0xFFF]++;
            }
            int maxSimd = 0; for (auto& kv : perSimd) maxSimd = kv.second > maxSimd ? kv.second : maxSimd;
            const double n = (double)iters * 32;
            printf("%-26s %2d waves/WG: per wave %6.2f cycles per instruction; wall %6.3f ns per instruction per wave = %6.3f ns per CU-instruction; distinct (se,sh,cu) ids %zu, (..,simd) ids %zu, most waves on one id %d\n",
                   names[kind], waves, cyc / (256.0 * waves) / n, ms * 1e6 / n, ms * 1e6 / n / waves, perCu.size(), perSimd.size(), maxSimd);
        }
    return 0;
}
