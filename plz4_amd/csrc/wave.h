// wave.h -- the only abstraction layer in the device code.
//
// The codec kernels are written for ONE 64-lane CDNA4 wavefront per LZ4 block: wave-uniform control
// flow (block cursor, output cursor, parser state live in scalar registers) with short SPMD sections
// in which the 64 lanes probe / compare / copy in parallel and vote with a 64-bit ballot.
//
// The same source is compiled twice:
//   * by hipcc for gfx950 -- the product (LANES(...) is the implicit SIMT lane, BALLOT is v_cmp+s_mov
//     of the exec-wide mask, RL is v_readlane_b32, UNI is v_readfirstlane_b32);
//   * by g++ with -DPLZ4_EMU for tests/emu -- a lane-emulation harness in which LANES(...) is a
//     for-loop over 64 lanes and per-lane variables are 64-element arrays.  That build exists so the
//     kernel LOGIC can be unit-tested on a machine without a GPU; it is test infrastructure, it is
//     never loaded by plz4_amd and it is not a CPU fallback.
//
// Conventions: LV(T, x) declares a per-lane variable, accessed as x[I_]; everything else is
// wave-uniform.  Code inside LANES(...) may diverge freely; code outside must be uniform.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(PLZ4_EMU)
// ---------------------------------------------------------------- CPU lane emulation (tests only)
#include <string.h>
#define DEV            static inline
#define DEVM           inline                           /* member functions */
#define LV(T, x)       T x[64]
#define I_             lane_
#define LANE           lane_
#define LANES(...)     for (int lane_ = plz4_emu_first(); lane_ != plz4_emu_end(); lane_ += plz4_emu_step()) { __VA_ARGS__ }
#define BALLOT(c)      ([&]() { uint64_t m_ = 0; for (int lane_ = 0; lane_ < 64; ++lane_) if (c) m_ |= 1ull << lane_; return m_; }())
#define RL(x, w)       ((x)[(w)])
#define RLF(x, f, w)   ((x)[(w)].f)                     /* read field f of lane w of a per-lane struct */
#define WL(x, w, v)    do { const auto wl_v_ = (v); (x)[(w)] = wl_v_; } while (0)   /* write one lane (uniform lane index) */
#define SHFL(x, l)     ((x)[(l) & 63])                  /* read another lane's value (per-lane lane index) */
#define UNI(x)         (x)
#define LANE_IN(mask)  ((((uint64_t)(mask)) >> lane_) & 1ull)   /* is this lane in a wave-uniform 64-bit mask */
#define LVREF(T, x)    T (&x)[64]                       /* a per-lane variable as a function parameter */
#define SCAN_INCL(x)   do { for (int s_ = 1; s_ < 64; ++s_) (x)[s_] += (x)[s_ - 1]; } while (0)   /* inclusive prefix sum over lanes */
/* x[l] <- max of x over the lanes strictly below l (0 for lane 0); x >= 0 */
#define SCAN_MAX_EXCL(x) do { int m_ = 0; for (int s_ = 0; s_ < 64; ++s_) { const int v_ = (x)[s_]; (x)[s_] = m_; if (v_ > m_) m_ = v_; } } while (0)
#define WAVE_FENCE()   do {} while (0)
#define LDS_FENCE()    do {} while (0)
#define LDS_ORDER()    do {} while (0)
// Same-address LDS store conflicts inside one instruction are resolved in an unspecified lane order on
// hardware; the emulation can run lanes ascending or descending so tests cover both resolutions.
extern int plz4_emu_descending;
static inline int plz4_emu_first() { return plz4_emu_descending ? 63 : 0; }
static inline int plz4_emu_end()   { return plz4_emu_descending ? -1 : 64; }
static inline int plz4_emu_step()  { return plz4_emu_descending ? -1 : 1; }
#else
// ---------------------------------------------------------------- gfx950
#include <hip/hip_runtime.h>
#define DEV            __device__ __forceinline__
#define DEVM           __device__ __forceinline__
#define LV(T, x)       T x[1]
#define I_             0
#define LANE           ((int)(threadIdx.x & 63u))
#define LANES(...)     { __VA_ARGS__ }
#define BALLOT(c)      ((uint64_t)__builtin_amdgcn_ballot_w64((bool)(c)))
#define RL(x, w)       plz4_readlane((x)[0], (w))
#define RLF(x, f, w)   plz4_readlane((x)[0].f, (w))
/* v is evaluated by the whole wave BEFORE the select (it may contain ballots); then v_cmp + v_cndmask */
#define WL(x, w, v)    do { const auto wl_v_ = (v); (x)[0] = (LANE == (w)) ? wl_v_ : (x)[0]; } while (0)
#define SHFL(x, l)     plz4_bpermute((x)[0], (l))
#define UNI(x)         plz4_readfirstlane((x))
// a wave-uniform 64-bit mask as a per-lane condition: the scalar register pair itself is the select / exec mask, where
// ((mask >> LANE) & 1) costs two v_and and a 64-bit v_cmp
#define LANE_IN(mask)  (__builtin_amdgcn_inverse_ballot_w64((uint64_t)(mask)))
#define LVREF(T, x)    T (&x)[1]
#define SCAN_INCL(x)   do { (x)[0] = plz4_scan_incl((x)[0]); } while (0)
#define SCAN_MAX_EXCL(x) do { (x)[0] = plz4_scan_max_excl((x)[0]); } while (0)
// Same-wave producer/consumer through memory needs no cache action on CDNA (one TCP, in-order VMEM
// queue); the fence only stops the compiler from reordering the accesses.
#define WAVE_FENCE()   __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define LDS_FENCE()    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
// Ordering of one wave's own LDS accesses only: the LDS executes a wave's instructions in issue order, so all that is needed is
// that the compiler keeps them in program order (no counter wait: a fence also waits for every global store in flight).
#define LDS_ORDER()    __asm__ volatile("" ::: "memory")

__device__ __forceinline__ uint32_t plz4_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ int      plz4_readlane(int v, int l)      { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t plz4_readlane(uint64_t v, int l)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
// Inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips): log steps inside each row of 16,
// then row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3.  All 64 lanes must be active.
__device__ __forceinline__ int plz4_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast31 -> rows 2, 3
    return v;
}
// Exclusive prefix maximum of non-negative values, same network; the final step moves everything up one lane (wave_shr:1).
__device__ __forceinline__ int plz4_scan_max_excl(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); v = v > t ? v : t;
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);            // wave_shr:1, lane 0 <- 0
}
__device__ __forceinline__ int      plz4_bpermute(int v, int l)      { return __builtin_amdgcn_ds_bpermute(l << 2, v); }
__device__ __forceinline__ uint32_t plz4_bpermute(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_ds_bpermute(l << 2, (int)v); }
__device__ __forceinline__ uint32_t plz4_readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int      plz4_readfirstlane(int v)      { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint8_t  plz4_readfirstlane(uint8_t v)  { return (uint8_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint16_t plz4_readfirstlane(uint16_t v) { return (uint16_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t plz4_readfirstlane(uint64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
#endif

// ---------------------------------------------------------------- unaligned little-endian access
// gfx950 runs with unaligned-access-mode on, so these become single global_load/store_{ushort,dword,dwordx2,dwordx4}.
typedef uint16_t __attribute__((aligned(1), may_alias)) u16u_t;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32u_t;
typedef uint64_t __attribute__((aligned(1), may_alias)) u64u_t;
struct __attribute__((packed, may_alias)) v16u_t { uint32_t w[4]; };

DEV uint16_t ld16u(const uint8_t* p) { return *(const u16u_t*)p; }
DEV uint32_t ld32u(const uint8_t* p) { return *(const u32u_t*)p; }
DEV uint64_t ld64u(const uint8_t* p) { return *(const u64u_t*)p; }
DEV void     st16u(uint8_t* p, uint16_t v) { *(u16u_t*)p = v; }
DEV void     st32u(uint8_t* p, uint32_t v) { *(u32u_t*)p = v; }
DEV void     st64u(uint8_t* p, uint64_t v) { *(u64u_t*)p = v; }

DEV int ctz64(uint64_t v) { return __builtin_ctzll(v); }
DEV uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
template <class T> DEV T min_(T a, T b) { return a < b ? a : b; }
template <class T> DEV T max_(T a, T b) { return a > b ? a : b; }
