// micro: issue cost of the vector instructions the level-1 parser and the decoder are made of, relative to v_add_u32 -- one wave
// per SIMD and four, 32 independent copies of the instruction per loop trip.  hipcc --offload-arch=gfx950 -O2 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(name, ASM, ...)                                                                                  \
    __global__ void name(int iters, uint32_t* out)                                                              \
    {                                                                                                           \
        uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9E3779B9u, c = b + 77u, d = c * 3u;                \
        uint64_t q = ((uint64_t)a << 32) | b, r = ((uint64_t)c << 32) | d;                                      \
        uint32_t s0 = 0; uint64_t m64 = 0;                                                                      \
        const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                 \
        for (int i = 0; i < iters; ++i) { REP32(asm volatile(ASM : __VA_ARGS__);) }                             \
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) { out[2] = (uint32_t)(t1 - t0); out[3] = (uint32_t)(r1 - r0); } \
        if (a + b + c + d + (uint32_t)q + (uint32_t)r + s0 + (uint32_t)m64 == 0x12345u) out[0] = 1;             \
    }

KERNEL(k_add,      "v_add_u32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_and_or,   "v_and_or_b32 %0, %1, %2, %0", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_mul_lo,   "v_mul_lo_u32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_mul_hi,   "v_mul_hi_u32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_mul_u24,  "v_mul_u32_u24 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_mad_u24,  "v_mad_u32_u24 %0, %1, %2, %0", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_mad64,    "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(q) : "v"(b), "v"(c) : "vcc")
KERNEL(k_shl64,    "v_lshlrev_b64 %0, %1, %0", "+v"(q) : "v"(b))
KERNEL(k_shr64,    "v_lshrrev_b64 %0, %1, %0", "+v"(q) : "v"(b))
KERNEL(k_addc,     "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc", "+v"(a), "+v"(c) : "v"(b) : "vcc")
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %1, %0, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_cmp32,    "v_cmp_lt_u32 vcc, %0, %1", : "v"(a), "v"(b) : "vcc")
KERNEL(k_cmp64,    "v_cmp_lt_u64 vcc, %0, %1", : "v"(q), "v"(r) : "vcc")
KERNEL(k_readlane, "v_readlane_b32 %0, %1, 5", "=s"(s0) : "v"(b))
KERNEL(k_readfl,   "v_readfirstlane_b32 %0, %1", "=s"(s0) : "v"(b))
KERNEL(k_dpp,      "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(a) : "v"(b))
KERNEL(k_bcast,    "v_add_u32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf", "+v"(a) : "v"(b))
KERNEL(k_bfe,      "v_bfe_u32 %0, %0, %1, 8", "+v"(a) : "v"(b))
KERNEL(k_perm,     "v_perm_b32 %0, %1, %0, %2", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %0, %2", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_alignbyte,"v_alignbyte_b32 %0, %1, %0, %2", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 3, %0", "+v"(a) : "v"(b))
KERNEL(k_add3,     "v_add3_u32 %0, %1, %2, %0", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_ffbh,     "v_ffbh_u32 %0, %0", "+v"(a) :)
KERNEL(k_ffbl,     "v_ffbl_b32 %0, %0", "+v"(a) :)
KERNEL(k_min,      "v_min_u32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_sdwa,     "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "+v"(a) : "v"(b))
KERNEL(k_mbcnt,    "v_mbcnt_lo_u32_b32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_salu,     "s_add_u32 %0, %0, 3", "+s"(s0) : : "scc")
KERNEL(k_bperm,    "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)", "+v"(a) : "v"(b))
KERNEL(k_swizzle,  "ds_swizzle_b32 %0, %0 offset:0x041F\n s_waitcnt lgkmcnt(0)", "+v"(a) :)

// (one asm statement each, so the compiler cannot put its own hazard nops between the copies)
#define S4(X) X X X X
#define S32(X) S4(X) S4(X) S4(X) S4(X) S4(X) S4(X) S4(X) S4(X)
#define KERNEL1(name, ASM, ...)                                                                                 \
    __global__ void name(int iters, uint32_t* out)                                                              \
    {                                                                                                           \
        uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9E3779B9u, c = b + 77u;                            \
        uint64_t m64 = (uint64_t)iters * 0x9E3779B97F4A7C15ull, q64 = ((uint64_t)a << 32) | b; uint32_t s0 = 0;  \
        const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                 \
        for (int i = 0; i < iters; ++i) { asm volatile(S32(ASM "\n") : __VA_ARGS__); }                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) { out[2] = (uint32_t)(t1 - t0); out[3] = (uint32_t)(r1 - r0); } \
        if (a + b + c + s0 + (uint32_t)m64 + (uint32_t)q64 == 0x12345u) out[0] = 1;                                             \
    }
KERNEL1(k1_cnd_vcc,  "v_cndmask_b32 %0, %1, %0, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL1(k1_cnd_sgpr, "v_cndmask_b32 %0, %1, %0, %2", "+v"(a) : "v"(b), "s"(m64))
KERNEL1(k1_cnd_indep,"v_cndmask_b32 %0, %1, %2, vcc", "=v"(a) : "v"(b), "v"(c) : "vcc")
KERNEL1(k1_and,      "v_and_b32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_or,       "v_or_b32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_xor,      "v_xor_b32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_sub,      "v_sub_u32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_shl,      "v_lshlrev_b32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_shr,      "v_lshrrev_b32 %0, 5, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_mov,      "v_mov_b32 %0, %1", "=v"(a) : "v"(b))
KERNEL1(k1_min,      "v_min_u32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_mul,      "v_mul_lo_u32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_bfe,      "v_bfe_u32 %0, %1, 3, 8", "=v"(a) : "v"(b))
KERNEL1(k1_andor,    "v_and_or_b32 %0, %1, %2, %1", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_lshladd,  "v_lshl_add_u32 %0, %1, 3, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_add3,     "v_add3_u32 %0, %1, %2, %1", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_addco,    "v_add_co_u32 %0, vcc, %1, %2", "=v"(a) : "v"(b), "v"(c) : "vcc")
KERNEL1(k1_cmp,      "v_cmp_lt_u32 vcc, %0, %1", : "v"(b), "v"(c) : "vcc")
KERNEL1(k1_cmpx,     "v_cmp_lt_u32 %0, %1, %2", "=s"(m64) : "v"(b), "v"(c))
KERNEL1(k1_shr64,    "v_lshrrev_b64 %0, 5, %1", "=v"(q64) : "v"(q64))
KERNEL1(k1_fma,      "v_fma_f32 %0, %1, %2, %1", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_addf,     "v_add_f32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_dppmov,   "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(a) : "v"(b))
KERNEL1(k1_perm,     "v_perm_b32 %0, %1, %2, %1", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_alignbit, "v_alignbit_b32 %0, %1, %2, 8", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_rl,       "v_readlane_b32 %0, %1, 7", "=s"(s0) : "v"(b))
KERNEL1(k1_sdwa,     "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_nop,      "s_nop 0", "+v"(a) : )
KERNEL1(k1_add_nop,  "v_add_u32 %0, %1, %0\n s_nop 0", "+v"(a) : "v"(b))
KERNEL1(k1_add_nop1, "v_add_u32 %0, %1, %0\n s_nop 1", "+v"(a) : "v"(b))
KERNEL1(k1_add_sadd, "v_add_u32 %0, %1, %0\n s_add_u32 %2, %2, 3", "+v"(a) : "v"(b), "s"(s0) : "scc")
KERNEL1(k1_add_indep,"v_add_u32 %0, %1, %2", "=v"(a) : "v"(b), "v"(c))
KERNEL1(k1_cmp_cnd,  "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL1(k1_cmp_cnd_s,"v_cmp_lt_u32 %1, %0, %2\n v_cndmask_b32 %0, %2, %0, %1", "+v"(a), "+s"(m64) : "v"(b))
KERNEL1(k1_rl_dep,   "v_readlane_b32 %1, %0, %1\n s_nop 3", "+v"(a), "+s"(s0) :)
KERNEL1(k1_ballot_rl,"v_cmp_lt_u32 vcc, %0, %1\n s_ff1_i32_b64 %2, vcc\n s_nop 3\n v_readlane_b32 %2, %0, %2", "+v"(a) : "v"(b), "s"(s0) : "vcc", "scc")
KERNEL1(k1_ldsw8,    "ds_write_b8 %0, %1", : "v"(a & 1023), "v"(b) : "memory")
KERNEL1(k1_ldsw64,   "ds_write_b64 %0, %1", : "v"((a & 127) * 8), "v"(q64) : "memory")
KERNEL1(k1_ldsr64,   "ds_read_b64 %0, %1", "=v"(q64) : "v"((a & 127) * 8) : "memory")

struct T { const char* name; void (*fn)(int, uint32_t*); int perRep; };
int main()
{
    uint32_t* d; hipMalloc(&d, 64);
    T tests[] = {{"v_add_u32", k_add, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
                 {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_mad_u64_u32", k_mad64, 1},
                 {"v_lshlrev_b64", k_shl64, 1}, {"v_lshrrev_b64", k_shr64, 1}, {"v_add_co+v_addc", k_addc, 2},
                 {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_lt_u32", k_cmp32, 1}, {"v_cmp_lt_u64", k_cmp64, 1},
                 {"v_readlane_b32", k_readlane, 1}, {"v_readfirstlane", k_readfl, 1}, {"v_add_u32_dpp shr", k_dpp, 1},
                 {"v_add_u32_dpp bcast", k_bcast, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_perm_b32", k_perm, 1},
                 {"v_alignbit_b32", k_alignbit, 1}, {"v_alignbyte_b32", k_alignbyte, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
                 {"v_add3_u32", k_add3, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_ffbl_b32", k_ffbl, 1}, {"v_min_u32", k_min, 1},
                 {"v_add_u32_sdwa", k_sdwa, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"s_add_u32", k_salu, 1},
                 {"ds_bpermute+wait", k_bperm, 1}, {"ds_swizzle+wait", k_swizzle, 1},
                 {"1: v_cndmask vcc", k1_cnd_vcc, 1}, {"1: v_cndmask sgpr", k1_cnd_sgpr, 1}, {"1: v_cndmask indep", k1_cnd_indep, 1},
                 {"1: v_and_b32", k1_and, 1}, {"1: v_or_b32", k1_or, 1}, {"1: v_xor_b32", k1_xor, 1}, {"1: v_sub_u32", k1_sub, 1},
                 {"1: v_lshlrev_b32", k1_shl, 1}, {"1: v_lshrrev_b32 imm", k1_shr, 1}, {"1: v_mov_b32", k1_mov, 1}, {"1: v_min_u32", k1_min, 1},
                 {"1: v_mul_lo_u32", k1_mul, 1}, {"1: v_bfe_u32", k1_bfe, 1}, {"1: v_and_or_b32", k1_andor, 1}, {"1: v_lshl_add_u32", k1_lshladd, 1},
                 {"1: v_add3_u32", k1_add3, 1}, {"1: v_add_co_u32", k1_addco, 1}, {"1: v_cmp vcc", k1_cmp, 1}, {"1: v_cmp sgpr", k1_cmpx, 1},
                 {"1: v_lshrrev_b64", k1_shr64, 1}, {"1: v_fma_f32", k1_fma, 1}, {"1: v_add_f32", k1_addf, 1}, {"1: v_mov_dpp", k1_dppmov, 1},
                 {"1: v_perm_b32", k1_perm, 1}, {"1: v_alignbit", k1_alignbit, 1}, {"1: v_readlane", k1_rl, 1}, {"1: v_add_sdwa", k1_sdwa, 1},
                 {"1: s_nop 0", k1_nop, 1}, {"1: v_add; s_nop 0", k1_add_nop, 1}, {"1: v_add; s_nop 1", k1_add_nop1, 1},
                 {"1: v_add; s_add", k1_add_sadd, 1}, {"1: v_add indep", k1_add_indep, 1}, {"1: v_cmp; v_cndmask", k1_cmp_cnd, 1},
                 {"1: v_cmp s; v_cnd s", k1_cmp_cnd_s, 1}, {"1: readlane chain", k1_rl_dep, 1}, {"1: cmp;ff1;readlane", k1_ballot_rl, 1},
                 {"1: ds_write_b8", k1_ldsw8, 1}, {"1: ds_write_b64", k1_ldsw64, 1}, {"1: ds_read_b64", k1_ldsr64, 1}};
    const int iters = 100000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double base[3] = {0, 0, 0};
    printf("%-22s %22s %22s %22s   (cycles per instruction: of one wave [s_memtime] / of the SIMD [x waves]; clock GHz)\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    k_add<<<256, 1024>>>(2000000, d); hipDeviceSynchronize();             // (clocks up)
    for (auto& t : tests) {
        printf("%-22s", t.name);
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int threads = 256 << cfg;                                   // 4 / 8 / 16 waves per CU
            t.fn<<<256, threads>>>(iters, d); hipDeviceSynchronize();
            uint32_t h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            const double cyc = (double)h[2] / ((double)iters * 32 * t.perRep);     // per instruction, of one wave
            const double ghz = (double)h[2] / ((double)h[3] * 10.0);               // s_memrealtime: 100 MHz
            printf("  %6.2f / %5.2f @%4.2f", cyc, cyc / (1 << cfg), ghz);
        }
        printf("\n");
    }
    return 0;
}
