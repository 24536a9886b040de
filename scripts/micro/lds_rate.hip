// micro: what the LDS pipe of a CU sustains for the parser's accesses -- ds_read_b32 and ds_max_rtn_u32 at 64 random slots of a
// 16 KiB table, all lanes or only some active -- with 1, 4 and 10 waves per CU (each wave its own table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND, int W> __global__ __launch_bounds__(64 * W) void k(int iters, uint32_t* out, uint64_t activeMask)
{
    extern __shared__ uint32_t lds[];
    uint32_t* T = lds + (threadIdx.x >> 6) * 4096;
    for (int i = threadIdx.x & 63; i < 4096; i += 64) T[i] = i;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const bool on = (activeMask >> (threadIdx.x & 63)) & 1;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t slot = (x >> 12) & 4095u;
            if (KIND == 0) acc += T[slot];                                                     // ds_read_b32, random
            if (KIND == 1) acc += atomicMax(&T[slot], on ? x : 0u);                            // ds_max_rtn_u32, all lanes (inactive ones exchange 0)
            if (KIND == 2) { if (on) acc += atomicMax(&T[slot], x); }                          // only the active lanes
            if (KIND == 3) acc += T[(threadIdx.x & 63) + ((x >> 12) & 63u) * 64];              // conflict-free read (lane == bank)
            if (KIND == 4) acc += __builtin_amdgcn_ds_bpermute((int)(slot & 63u) << 2, (int)x); // bpermute
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * W + (threadIdx.x >> 6)] = (uint32_t)(t1 - t0);
    if (acc == 0x12345u) out[0] = 1;
}
template <int KIND, int W> void run(const char* name, uint32_t* d, uint64_t mask)
{
    static uint32_t h[256 * 16];
    const int iters = 20000;
    k<KIND, W><<<256, 64 * W, W * 16384>>>(iters, d, mask); (void)hipDeviceSynchronize();
    k<KIND, W><<<256, 64 * W, W * 16384>>>(iters, d, mask); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 4 * 256 * W, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256 * W; ++i) s += h[i];
    const double cyc = s / (256.0 * W) / (iters * 8.0);
    printf("%-44s %2d waves per CU: %7.1f cycles per access of one wave, %6.1f cycles of the CU per access\n", name, W, cyc, cyc / W);
}
#define ALL(K, NAME, MASK) run<K, 1>(NAME, d, MASK); run<K, 4>(NAME, d, MASK); run<K, 10>(NAME, d, MASK);
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 4 * 256 * 16);
    const uint64_t all = ~0ull, some = 0x1111111111111111ull /* 16 lanes */, few = 0x0101010101010101ull /* 8 lanes */;
    (void)hipFuncSetAttribute((const void*)k<0, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<1, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<2, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<3, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<4, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    ALL(0, "ds_read_b32, 64 random slots", all)
    ALL(3, "ds_read_b32, lane == bank", all)
    ALL(1, "ds_max_rtn_u32, 64 random slots (48 exchange 0)", some)
    ALL(2, "ds_max_rtn_u32, 16 active lanes", some)
    ALL(2, "ds_max_rtn_u32, 8 active lanes", few)
    ALL(2, "ds_max_rtn_u32, 64 active lanes", all)
    ALL(4, "ds_bpermute_b32", all)
    return 0;
}
