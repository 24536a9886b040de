// FETCH_SIZE calibration for this repo's load shapes (MI355X_MICROARCH.md §HBM: the counter is exact/2 for wide coalesced
// streams; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Four read-only kernels over a 2 GiB buffer (far beyond L2 + Infinity Cache), each touching every byte / line once:
//   k_stream16     coalesced 16 B per lane (the guide's reference shape)                          true bytes = 2 GiB
//   k_window16     lane l loads 16 B at p + l, p advancing 64 per step (the decoder's and encoder's window loads:
//                  64 overlapping unaligned requests per 79-byte span)                             true bytes = 2 GiB
//   k_scatter24a   one 24-byte read (16 + 8) per 128-byte line, inside one 64-byte half, lines in random order
//                  (a candidate window / a far match source)                                       one line (or sector) each
//   k_scatter24b   the same read straddling the two 64-byte halves of its line
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/_build/fetch_calib scripts/micro/fetch_calib.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f --output-format csv -- scripts/_build/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, may_alias)) v16 { uint32_t w[4]; };
struct __attribute__((packed, may_alias)) v8 { uint32_t w[2]; };
__global__ void k_stream16(const uint8_t* p, size_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i + 16 <= n; i += (size_t)gridDim.x * blockDim.x * 16) { v16 v = *(const v16*)(p + i); acc ^= v.w[0] ^ v.w[3]; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_window16(const uint8_t* p, size_t n, uint32_t* sink)     // one wave per 256 KiB region
{
    const size_t region = (size_t)256 << 10;
    const size_t lo = (size_t)blockIdx.x * region;
    uint32_t acc = 0;
    for (size_t q = lo; q + 64 + 16 <= lo + region && q + 80 <= n; q += 64) { v16 v = *(const v16*)(p + q + threadIdx.x); acc ^= v.w[0] ^ v.w[3]; }
    if (acc == 0x12345678u) *sink = acc;
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int kOff> __global__ void k_scatter24(const uint8_t* p, uint32_t lines, uint32_t* sink)   // lines: a power of two
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += gridDim.x * blockDim.x) {
        const uint32_t l = (i * 2654435761u + 12345u) & (lines - 1);          // an odd multiplier permutes the lines
        const uint8_t* q = p + (size_t)l * 128 + kOff;
        v16 a = *(const v16*)q; v8 b = *(const v8*)(q + 16);
        acc ^= a.w[0] ^ b.w[1];
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main()
{
    const size_t n = (size_t)2 << 30;
    uint8_t* d; uint32_t* sink;
    if (hipMalloc((void**)&d, n + 256) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) return 1;
    hipMemset(d, 1, n + 256);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, d, n, sink);
        hipLaunchKernelGGL(k_window16, dim3((unsigned)(n >> 18)), dim3(64), 0, 0, d, n, sink);
        hipLaunchKernelGGL(k_scatter24<8>, dim3(4096), dim3(256), 0, 0, d, (uint32_t)(n >> 7), sink);
        hipLaunchKernelGGL(k_scatter24<52>, dim3(4096), dim3(256), 0, 0, d, (uint32_t)(n >> 7), sink);
    }
    hipDeviceSynchronize();
    printf("fetch_calib: 2 GiB buffer, %u lines of 128 B\n", (unsigned)(n >> 7));
    return 0;
}
