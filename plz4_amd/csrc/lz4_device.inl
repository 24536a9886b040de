// lz4_device.inl -- one-wavefront-per-block LZ4 block codec + xxHash32 for gfx950 (see wave.h for the
// two ways this file is compiled).  Bit-exact with liblz4 v1.10.0 as plz4 drives it:
//
//   wave_encode_block   == LZ4_compress_fast(accel 1)      /root/reference/internal/pkg/clz4/lz4.c:1453 -> :1382 -> :930-1338
//   wave_decode_block   == LZ4_decompress_safe             lz4.c:2451 -> :2022-2445
//   wave_xxh32          == xxh32.ChecksumZero              internal/pkg/xxh32/xxh32zero.go:238-280
//
// The reference parser is a strictly sequential greedy matcher whose output depends on the exact order of
// hash-table reads and writes.  It is NOT re-ordered here.  What the wave does instead:
//   * evaluates the next <=64 probe positions of the search loop at once, one lane per probe (their
//     positions are a closed form of the probe counter), commits them to the LDS hash table
//     speculatively, finds the first lane whose candidate passes the distance + 4-byte test with a ballot
//     and rolls the table back for the lanes behind the winner;  same-hash collisions inside a batch are
//     detected with a store/read-back and handled by shrinking the batch to its collision-free prefix, so
//     the table always holds exactly what the sequential parser would have written;
//   * runs LZ4_count as 64 x 8-byte compares + ballot, the backwards catch-up the same way, and copies
//     literals / match bytes cooperatively.
#pragma once
#include "wave.h"

// Diagnostics build only (-DPLZ4_STATS, scripts/stats_build.sh): per-wave event counts and cycle sums, added to a
// global array at the end of every block.  Compiled out of the product.
#if defined(PLZ4_STATS) && !defined(PLZ4_EMU)
#define STAT(i, v)   (st_[(i)] += (unsigned long long)(v))
#define STAT_DECL    unsigned long long st_[24] = {0}
#define STAT_NOW()   ((unsigned long long)__builtin_readcyclecounter())
#define STAT_FLUSH() do { if (LANE == 0) for (int i_ = 0; i_ < 24; ++i_) atomicAdd(&plz4_stats[i_], st_[i_]); } while (0)
extern __device__ unsigned long long plz4_stats[24];
#else
#define STAT(i, v)   do {} while (0)
#define STAT_DECL    do {} while (0)
#define STAT_NOW()   0ull
#define STAT_FLUSH() do {} while (0)
#endif
enum { S_GRID = 0, S_GENERIC, S_MISORDER, S_SEQ_GRID, S_SEQ_GEN, S_TWINSTOP, S_SAT, S_LONGBACK, S_MEMLIT, S_WALKITER,
       S_CYC_TOTAL, S_CYC_LOAD, S_CYC_WALK, S_CYC_FIX, S_CYC_GEN, S_CYC_SAT, S_CYC_MEMLIT, S_BLOCKS, S_LANES_EXEC };

namespace plz4 {

enum : int {
    kMinMatch = 4, kMfLimit = 12, kLastLiterals = 5, kMinLength = 13,
    k64KLimit = 65536 + kMfLimit - 1,            // lz4.c:710
    kMaxInput = 0x7E000000,                      // lz4.h:214
    kHashBytes = 16384                           // LZ4_HASHTABLESIZE, lz4.h:696
};
static constexpr uint32_t kMaxDist = 65535u;     // lz4.h:674

DEV int compress_bound(int n) { return ((unsigned)n > (unsigned)kMaxInput) ? 0 : n + n / 255 + 16; }

// ------------------------------------------------------------------------------------------ copies
// Forward byte copy by the whole wave; regions must not overlap (literals: input -> output).
DEV void wave_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int len)
{
    LANES({
        int i = LANE * 16;
        for (; i + 16 <= len; i += 64 * 16) *(v16u_t*)(dst + i) = *(const v16u_t*)(src + i);
        // tail: fewer than 16 bytes left for exactly one lane's slot; finish byte-wise across lanes
    })
    const int done = len & ~15;
    LANES({ const int i = done + LANE; if (i < len) dst[i] = src[i]; })
}

// dst[i] = 0xFF for i in [0,len)
DEV void wave_fill_ff(uint8_t* dst, int len)
{
    LANES({ for (int i = LANE; i < len; i += 64) dst[i] = 0xFF; })
}

// ------------------------------------------------------------------------------------------ xxHash32
static constexpr uint32_t XP1 = 2654435761u, XP2 = 2246822519u, XP3 = 3266489917u, XP4 = 668265263u, XP5 = 374761393u;

// Lanes 0..3 each own one accumulator (the four chains are independent; each is strictly sequential).
DEV uint32_t wave_xxh32(const uint8_t* p, int n)
{
    uint32_t h = (uint32_t)n;
    int rem = n;
    if (n >= 16) {
        LV(uint32_t, acc);
        const int stripes = n >> 4;
        LANES({
            const int l = LANE & 3;
            acc[I_] = (l == 0) ? XP1 + XP2 : (l == 1) ? XP2 : (l == 2) ? 0u : 0u - XP1;
            if (LANE < 4) {
                const uint8_t* q = p + 4 * l;
                int s = 0;
                for (; s + 8 <= stripes; s += 8) {          // 8 loads in flight per lane
                    uint32_t x0 = ld32u(q), x1 = ld32u(q + 16), x2 = ld32u(q + 32), x3 = ld32u(q + 48);
                    uint32_t x4 = ld32u(q + 64), x5 = ld32u(q + 80), x6 = ld32u(q + 96), x7 = ld32u(q + 112);
                    uint32_t a = acc[I_];
                    a = rotl32(a + x0 * XP2, 13) * XP1; a = rotl32(a + x1 * XP2, 13) * XP1;
                    a = rotl32(a + x2 * XP2, 13) * XP1; a = rotl32(a + x3 * XP2, 13) * XP1;
                    a = rotl32(a + x4 * XP2, 13) * XP1; a = rotl32(a + x5 * XP2, 13) * XP1;
                    a = rotl32(a + x6 * XP2, 13) * XP1; a = rotl32(a + x7 * XP2, 13) * XP1;
                    acc[I_] = a; q += 128;
                }
                for (; s < stripes; ++s) { acc[I_] = rotl32(acc[I_] + ld32u(q) * XP2, 13) * XP1; q += 16; }
            }
        })
        h += rotl32(RL(acc, 0), 1) + rotl32(RL(acc, 1), 7) + rotl32(RL(acc, 2), 12) + rotl32(RL(acc, 3), 18);
        p += (size_t)stripes << 4; rem -= stripes << 4;
    } else {
        h += XP5;
    }
    // <16 tail bytes: uniform
    while (rem >= 4) { h = rotl32(h + UNI(ld32u(p)) * XP3, 17) * XP4; p += 4; rem -= 4; }
    while (rem)      { h = rotl32(h + (uint32_t)UNI(*p) * XP5, 11) * XP1; p++; rem--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

// ------------------------------------------------------------------------------------------ encoder
// Hash of the bytes at p (lz4.c:777-806, 64-bit little-endian build).
template <bool U16> DEV uint32_t seq_hash(uint64_t seq8)
{
    if (U16) return ((uint32_t)seq8 * 2654435761u) >> 19;                   // hash4, 13 bits
    return (uint32_t)(((seq8 << 24) * 889523592379ull) >> 52);              // hash5, 12 bits
}
template <bool U16> DEV uint32_t tab_get(const void* t, uint32_t h) { return U16 ? (uint32_t)((const uint16_t*)t)[h] : ((const uint32_t*)t)[h]; }
template <bool U16> DEV void     tab_put(void* t, uint32_t h, uint32_t v) { if (U16) ((uint16_t*)t)[h] = (uint16_t)v; else ((uint32_t*)t)[h] = v; }

// Position of probe number i (i = 0,1,2,...) of a search that started at `base`, and the stride to
// the following probe.  lz4.c:1043-1053: stride is 1 for the first 65 probes, then grows by one every
// 64 probes (searchMatchNb starts at 64, stride = searchMatchNb++ >> 6).
DEV void probe_pos(int base, int i, int* pos, int* stride)
{
    if (i == 0) { *pos = base; *stride = 1; return; }
    const int m = i - 1;                                 // F(m) = sum_{j=1..m} (63+j)>>6
    const int t = (63 + m) >> 6;                         // tier of m (0 when m == 0)
    const int F = (m == 0) ? 0 : 32 * t * (t - 1) + (m - 64 * (t - 1)) * t;
    *pos = base + 1 + F;
    *stride = (63 + i) >> 6;
}

// 8 bytes at src+pos of which only `valid` may be touched (block tail): never reads past the block.
DEV uint64_t ld64_guard(const uint8_t* src, int pos, int valid)
{
    if (valid >= 8) return ld64u(src + pos);
    uint64_t v = 0;
    for (int b = 0; b < valid; ++b) v |= (uint64_t)src[pos + b] << (8 * b);
    return v;
}

// LZ4_count (lz4.c:680-703) == min(common prefix of src[a..] and src[b..], limit - a).  a > b.
DEV int wave_common_len(const uint8_t* src, int a, int b, int limit)
{
    int total = 0;
    for (;;) {
        LV(uint64_t, d);
        LANES({
            const int off = 8 * LANE;
            const int valid = limit - (a + off);                     // bytes this lane may compare
            if (valid <= 0) d[I_] = ~0ull;
            else {
                uint64_t x = ld64_guard(src, a + off, valid) ^ ld64_guard(src, b + off, valid);
                if (valid < 8) x |= ~0ull << (8 * valid);
                d[I_] = x;
            }
        })
        const uint64_t stop = BALLOT(d[I_] != 0);
        if (stop) {
            const int l = ctz64(stop);
            const uint64_t dl = RL(d, l);
            return total + 8 * l + (ctz64(dl) >> 3);
        }
        total += 512; a += 512; b += 512;
    }
}

// Backwards extension (lz4.c:1105-1109): how many bytes before (a, b) are equal, at most `maxBack`.
DEV int wave_common_back(const uint8_t* src, int a, int b, int maxBack)
{
    int total = 0;
    while (total < maxBack) {
        const uint64_t neq = BALLOT((total + LANE >= maxBack) || (src[a - 1 - total - LANE] != src[b - 1 - total - LANE]));
        if (neq) return total + ctz64(neq);
        total += 64;
    }
    return maxBack;
}

// Length bytes after a token: (len-15) as a run of 0xFF and a final byte <255 (lz4.c:1123-1128, :1213-1223).
DEV int emit_len_ext(uint8_t* dst, int op, int rest)
{
    const int nff = rest / 255;
    if (nff) wave_fill_ff(dst + op, nff);
    LANES({ if (LANE == 0) dst[op + nff] = (uint8_t)(rest - nff * 255); })
    return op + nff + 1;
}

// ---- grid mode helpers -----------------------------------------------------------------------------------
// 24 bytes around a position x: [x-4, x+20).  back = the 4 bytes before x (as a little-endian u32, byte 3 is
// x-1), seq = [x, x+8), f1 = [x+8, x+16), f2 = [x+16, x+20).
struct Win24 { uint32_t back; uint64_t seq; uint64_t f1; uint32_t f2; };

DEV Win24 load_win24(const uint8_t* src, int x)
{
    Win24 w;
    if (x >= 4) {
        const v16u_t a = *(const v16u_t*)(src + x - 4);           // [x-4, x+12)
        const uint64_t b = ld64u(src + x + 12);                     // [x+12, x+20)
        w.back = a.w[0];
        w.seq  = (uint64_t)a.w[1] | ((uint64_t)a.w[2] << 32);
        w.f1   = (uint64_t)a.w[3] | (b << 32);
        w.f2   = (uint32_t)(b >> 32);
    } else {                                                        // candidates in the first 4 bytes of the block
        uint32_t bk = 0;
        for (int i = 0; i < x; ++i) bk |= (uint32_t)src[x - 1 - i] << (8 * (3 - i));
        w.back = bk;
        w.seq = ld64u(src + x); w.f1 = ld64u(src + x + 8); w.f2 = ld32u(src + x + 16);
    }
    return w;
}

// equal bytes of [x+4, x+20) in two windows: 0..16
DEV int win_fwd(const Win24& p, const Win24& c)
{
    const uint32_t x0 = (uint32_t)(p.seq >> 32) ^ (uint32_t)(c.seq >> 32);
    if (x0) return __builtin_ctz(x0) >> 3;
    const uint64_t x1 = p.f1 ^ c.f1;
    if (x1) return 4 + (ctz64(x1) >> 3);
    const uint32_t x2 = p.f2 ^ c.f2;
    if (x2) return 12 + (__builtin_ctz(x2) >> 3);
    return 16;
}
// equal bytes going backwards from x-1: 0..4
DEV int win_bck(const Win24& p, const Win24& c)
{
    const uint32_t y = p.back ^ c.back;
    return y ? (__builtin_clz(y) >> 3) : 4;
}

#if defined(PLZ4_EMU)
static inline uint32_t lds_max_rtn(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
static inline void     lds_min(uint32_t* p, uint32_t v)     { if (v < *p) *p = v; }
static inline void     lds_max(uint32_t* p, uint32_t v)     { if (v > *p) *p = v; }
#else
__device__ __forceinline__ uint32_t lds_max_rtn(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
__device__ __forceinline__ void     lds_min(uint32_t* p, uint32_t v)     { atomicMin(p, v); }
__device__ __forceinline__ void     lds_max(uint32_t* p, uint32_t v)     { atomicMax(p, v); }
#endif

DEV uint64_t lane_range(int lo, int hi)          // bits lo..hi inclusive, 0 <= lo, hi <= 63; empty if hi < lo
{
    if (hi < lo) return 0;
    const uint64_t upto = (hi >= 63) ? ~0ull : ((1ull << (hi + 1)) - 1);
    return upto & (~0ull << lo);
}

// Returns the compressed size, or 0 when liblz4 would return 0 (`limited` and the output does not fit its
// conservative checks).  `tab` = 16 KiB of LDS owned by this wave.  noDict (independent blocks, no dictionary).
//
// Two kinds of batch, same parser state between them:
//   GRID  (byU32 only, away from both ends of the block, probe stride still 1): the 64 lanes are the 64
//         consecutive positions of an aligned window.  One LDS atomic-max per lane commits the position and
//         returns the candidate; one memory round trip brings every lane 24 bytes around its position and
//         around its candidate, so hit / forward length (<=16 beyond the 4) / backward length (<=4) are known
//         for all 64 positions at once.  A scalar walk then plays the sequential parser over the window --
//         usually several sequences per batch -- and the table is patched (atomic-min for positions the
//         parser skipped, atomic-max for the ones it executed) to exactly the sequential result.
//   GENERIC (everything else: byU16, block head/tail, stride > 1): one lane per probe of the search loop,
//         closed-form probe positions, plain store + read-back collision detection, one sequence per batch.
template <bool U16>
DEV int wave_encode_block_tt(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst,
                             const int cap, const bool limited, void* tab)
{
    // fresh zeroed table per block (LZ4_initStream, lz4.c:1384)
    LANES({ uint32_t* t = (uint32_t*)tab; for (int i = LANE; i < kHashBytes / 4; i += 64) t[i] = 0; })
    LDS_FENCE();

    const int lastProbe  = n - kMfLimit + 1;      // mflimitPlusOne (lz4.c:963)
    const int matchLimit = n - kLastLiterals;     // lz4.c:964
    int anchor = 0, op = 0;
    STAT_DECL;
    const unsigned long long tBlock0 = STAT_NOW();
    (void)tBlock0;

    if (n >= kMinLength) {
        // Parser state between batches.
        int  insPos  = 0;  bool hasIns = true;     // pending table insert ("First Byte" lz4.c:1005-1010; ip-2 lz4.c:1236-1242)
        int  rePos   = 0;  bool hasRe  = false;    // pending immediate re-test at ip after a match (lz4.c:1255-1294)
        int  sBase   = 1;  int  sIter  = 0;        // search started at sBase; next un-probed probe number
        int  width   = 16;                         // generic batches: 16 lanes first, 64 when a search drags on

        for (;;) {
            // ================================================================ GRID batch
            if (!U16 && sIter <= 64) {
                const int probeStart = hasRe ? rePos : sBase + sIter;
                const int firstPos   = hasIns ? insPos : probeStart;
                const int base       = firstPos & ~63;
                if (base >= 64 && base + 96 <= n) {
                    uint32_t* T = (uint32_t*)tab;
                    STAT(S_GRID, 1);
                    const unsigned long long tg0 = STAT_NOW(); (void)tg0;
                    LV(int, act); LV(uint32_t, h); LV(uint32_t, r); LV(uint32_t, lit8);
                    LV(int, hit); LV(int, fwd); LV(int, bck);
                    LANES({
                        const int q = base + LANE;
                        const int isIns = hasIns && q == insPos;
                        act[I_] = isIns || q >= probeStart;
                        hit[I_] = 0; fwd[I_] = 0; bck[I_] = 0; r[I_] = 0; h[I_] = 0;
                        const Win24 P = load_win24(src, q);         // every lane: its byte may be a pending literal
                        lit8[I_] = (uint32_t)(P.seq & 0xFF);
                        if (act[I_]) {
                            h[I_] = seq_hash<false>(P.seq);
                            r[I_] = lds_max_rtn(&T[h[I_]], (uint32_t)q);
                            if (!isIns && r[I_] < (uint32_t)q && r[I_] + kMaxDist >= (uint32_t)q) {
                                const Win24 Cw = load_win24(src, (int)r[I_]);
                                if ((uint32_t)Cw.seq == (uint32_t)P.seq) {
                                    hit[I_] = 1;
                                    fwd[I_] = win_fwd(P, Cw);
                                    bck[I_] = min_(win_bck(P, Cw), (int)r[I_]);
                                }
                            }
                        }
                    })
                    LDS_FENCE();
                    // LDS atomics on one slot are expected to resolve in ascending lane order (then r is the
                    // nearest earlier twin or the pre-batch value).  Any other order shows up as r >= q somewhere.
                    const uint64_t misorder = BALLOT(act[I_] && r[I_] >= (uint32_t)(base + LANE));
                    if (misorder) {
                        STAT(S_MISORDER, 1);
                        LANES({ if (act[I_]) lds_min(&T[h[I_]], r[I_]); })     // min over a slot's group == its pre-batch value
                        LDS_FENCE();
                        goto generic_batch;
                    }
                    {
                        const uint64_t hits   = BALLOT(hit[I_]);
                        const uint64_t twins  = BALLOT(act[I_] && r[I_] >= (uint32_t)firstPos);   // has an earlier twin in this batch
                        const uint64_t actM   = BALLOT(act[I_]);
                        const unsigned long long tg1 = STAT_NOW(); (void)tg1;
                        STAT(S_CYC_LOAD, tg1 - tg0);
                        uint64_t E = 0;                                       // lanes the sequential parser really executes
                        if (hasIns) { E |= 1ull << (insPos - base); hasIns = false; }
                        int  cur = probeStart - base;                         // next probe lane (may be >= 64: none here)
                        bool curIsRe = hasRe;
                        bool finished = false;
                        if (cur < 64) hasRe = false;                          // consumed by the walk below

                        while (cur < 64) {
                            STAT(S_WALKITER, 1);
                            const uint64_t hm = hits & (~0ull << cur);
                            const int w = hm ? ctz64(hm) : 64;
                            int lim = 63;                                     // probe number <= 65 keeps the stride at 1
                            { const int x = sBase + 65 - base; if (x < lim) lim = x; }
                            int last = min_(w, lim);
                            // a lane with an earlier twin is only right if that twin was itself executed
                            const int curL = cur; const uint64_t EE = E;
                            const uint64_t bad = twins & lane_range(cur, max_(last, cur) > 63 ? 63 : max_(last, cur)) &
                                BALLOT(((int)r[I_] - base) < curL && !((EE >> (((int)r[I_] - base) & 63)) & 1));
                            bool stopForTwin = false;
                            if (bad) { const int b = ctz64(bad); if (b <= last) { last = b - 1; stopForTwin = true; STAT(S_TWINSTOP, 1); } }

                            if (w <= last) {
                                // ---- match at lane w
                                E |= lane_range(cur, w);
                                STAT(S_SEQ_GRID, 1);
                                const bool isRe = curIsRe && (w == cur);
                                const int p0 = base + w, c0 = (int)RL(r, w);
                                int p = p0, c = c0;
                                if (!isRe) {                                   // catch-up (lz4.c:1105-1109)
                                    const int maxBack = min_(p - anchor, c);
                                    int back = min_((int)RL(bck, w), maxBack);
                                    if (back == 4 && maxBack > 4) { STAT(S_LONGBACK, 1); back += wave_common_back(src, p - 4, c - 4, maxBack - 4); }
                                    p -= back; c -= back;
                                }
                                const int lit = p - anchor;
                                const int tokPos = op++;
                                if (limited && !isRe && (int64_t)op + lit + (2 + 1 + kLastLiterals) + lit / 255 > cap) return 0;
                                if (lit >= 15) op = emit_len_ext(dst, op, lit - 15);
                                if (lit > 0) {
                                    if (anchor < base) { STAT(S_MEMLIT, 1); const unsigned long long tm0 = STAT_NOW(); (void)tm0;
                                        wave_copy(dst + op, src + anchor, min_(lit, base - anchor)); STAT(S_CYC_MEMLIT, STAT_NOW() - tm0); }
                                    const int opl = op, an = anchor, pe = p;
                                    LANES({ const int q = base + LANE; if (q >= an && q < pe) dst[opl + (q - an)] = (uint8_t)lit8[I_]; })
                                }
                                op += lit;
                                LANES({ if (LANE == 0) st16u(dst + op, (uint16_t)(p - c)); })
                                op += 2;
                                int mc = (int)RL(fwd, w);
                                if (mc == 16) { STAT(S_SAT, 1); const unsigned long long ts0 = STAT_NOW(); (void)ts0;
                                    mc += wave_common_len(src, p0 + 20, c0 + 20, matchLimit); STAT(S_CYC_SAT, STAT_NOW() - ts0); }
                                const int e = p0 + kMinMatch + mc;
                                mc += p0 - p;                                  // the match starts `back` bytes earlier
                                if (limited && (int64_t)op + (1 + kLastLiterals) + (mc + 240) / 255 > cap) return 0;
                                LANES({ if (LANE == 0) dst[tokPos] = (uint8_t)((min_(lit, 15) << 4) | min_(mc, 15)); })
                                if (mc >= 15) op = emit_len_ext(dst, op, mc - 15);

                                anchor = e;
                                if (e >= lastProbe) { finished = true; break; }                 // lz4.c:1233
                                if (e - 2 < base + 64) E |= 1ull << (e - 2 - base);             // lz4.c:1236-1242
                                else { hasIns = true; insPos = e - 2; }
                                sBase = e + 1; sIter = 0;
                                if (e < base + 64) { cur = e - base; curIsRe = true; continue; }
                                hasRe = true; rePos = e;
                                break;
                            }
                            // ---- no match up to `last`: those probes are misses
                            if (last >= cur) {
                                E |= lane_range(cur, last);
                                if (curIsRe) { curIsRe = false; }             // the re-test missed; search runs from sBase (= its pos + 1)
                            }
                            {
                                const int nextPos = base + last + 1;
                                if (curIsRe) { hasRe = true; rePos = nextPos; }   // nothing executed yet (twin stop right at the re-test)
                                else sIter = nextPos - sBase;
                            }
                            (void)stopForTwin;
                            break;
                        }
                        if (finished) break;
                        const unsigned long long tg2 = STAT_NOW(); (void)tg2;
                        STAT(S_CYC_WALK, tg2 - tg1);
                        STAT(S_LANES_EXEC, __builtin_popcountll(E));
                        // ---- patch the table to the sequential result
                        LANES({ if (act[I_] && !((E >> LANE) & 1)) lds_min(&T[h[I_]], r[I_]); })
                        if (twins) { LDS_FENCE(); LANES({ if ((E >> LANE) & 1) lds_max(&T[h[I_]], (uint32_t)(base + LANE)); }) }
                        LDS_FENCE();
                        (void)actM;
                        STAT(S_CYC_FIX, STAT_NOW() - tg2);
                        width = 64;
                        continue;
                    }
                }
            }
generic_batch:
            // ================================================================ GENERIC batch
            {
            STAT(S_GENERIC, 1);
            const unsigned long long tq0 = STAT_NOW(); (void)tq0;
            const int pre = (hasIns ? 1 : 0) + (hasRe ? 1 : 0);
            LV(int, q); LV(uint32_t, h); LV(uint32_t, old); LV(uint32_t, rb); LV(uint32_t, lo4);
            LV(int, ok);        // lane has a probe to make (not past the end-of-block stop, within width)
            LV(int, hit);

            // ---- positions + termination test (lz4.c:1051-1055): probe i happens only if pos+stride <= lastProbe
            LANES({
                const int k = LANE - pre;
                int pos, stride, fine = 1;
                if (k < 0) { pos = (hasIns && LANE == 0) ? insPos : rePos; }
                else { probe_pos(sBase, sIter + k, &pos, &stride); fine = (pos + stride <= lastProbe); }
                q[I_] = pos; ok[I_] = fine && (LANE < width);
            })
            const uint64_t okMask = BALLOT(ok[I_]);
            // lanes are usable up to the first not-ok lane; a not-ok lane inside `width` means "end of block reached"
            const int nproc = (~okMask) ? ctz64(~okMask) : 64;
            const bool endInBatch = nproc < width;

            // ---- hash, table read, speculative commit, read-back
            LANES({
                hit[I_] = 0;
                if (LANE < nproc) {
                    const uint64_t s8 = ld64u(src + q[I_]);
                    lo4[I_] = (uint32_t)s8;
                    h[I_] = seq_hash<U16>(s8);
                    old[I_] = tab_get<U16>(tab, h[I_]);
                }
            })
            LDS_FENCE();
            LANES({ if (LANE < nproc) tab_put<U16>(tab, h[I_], (uint32_t)q[I_]); })
            LDS_FENCE();
            LANES({ rb[I_] = (LANE < nproc) ? tab_get<U16>(tab, h[I_]) : (uint32_t)q[I_]; })
            // candidate test (lz4.c:1090-1099); the insert-only lane never matches
            LANES({
                if (LANE < nproc && !(hasIns && LANE == 0)) {
                    const uint32_t cur = (uint32_t)q[I_];
                    if (U16 || old[I_] + kMaxDist >= cur) hit[I_] = (ld32u(src + old[I_]) == lo4[I_]);
                }
            })
            const uint64_t losers = BALLOT((LANE < nproc) && rb[I_] != (uint32_t)q[I_]);
            const uint64_t hits   = BALLOT(hit[I_]);

            // collision-free prefix: lanes before the first lane that lost a same-slot store (lane 0 never has an earlier twin)
            int safe = nproc;
            if (losers) safe = min_(nproc, max_(ctz64(losers), 1));
            const uint64_t hitsSafe = hits & ((safe >= 64) ? ~0ull : ((1ull << safe) - 1));
            const int keep = hitsSafe ? ctz64(hitsSafe) + 1 : safe;          // lanes [0,keep) are really executed

            // ---- make the table exactly what the sequential parser would have left
            LANES({ if (LANE >= keep && LANE < nproc) tab_put<U16>(tab, h[I_], old[I_]); })
            if (losers) { LDS_FENCE(); LANES({ if (LANE < keep) tab_put<U16>(tab, h[I_], (uint32_t)q[I_]); }) }
            LDS_FENCE();

            if (!hitsSafe) {
                // no match among the executed lanes: advance the parser state past them
                int used = keep;
                if (hasIns && used > 0) { hasIns = false; used--; }
                if (hasRe  && used > 0) { hasRe = false; used--; }
                sIter += used;
                STAT(S_CYC_GEN, STAT_NOW() - tq0);
                if (keep == nproc && endInBatch && !hasIns && !hasRe) break;      // -> last literals (lz4.c:1055)
                width = 64;
                continue;
            }

            // ---- a match: winner lane w
            const int  w       = keep - 1;
            const bool isRe    = hasRe && (w == pre - 1);
            int        p       = RL(q, w);
            int        c       = (int)RL(old, w);

            // catch-up over pending literals (lz4.c:1105-1109); a re-test has none (ip == anchor)
            if (!isRe) {
                const int back = wave_common_back(src, p, c, min_(p - anchor, c));
                p -= back; c -= back;
            }

            // literals (lz4.c:1112-1136)
            const int lit = p - anchor;
            const int tokPos = op++;
            if (limited && !isRe && (int64_t)op + lit + (2 + 1 + kLastLiterals) + lit / 255 > cap) return 0;
            if (lit >= 15) op = emit_len_ext(dst, op, lit - 15);
            wave_copy(dst + op, src + anchor, lit);
            op += lit;

            // offset + match length (lz4.c:1155-1226)
            LANES({ if (LANE == 0) st16u(dst + op, (uint16_t)(p - c)); })
            op += 2;
            const int mc = wave_common_len(src, p + kMinMatch, c + kMinMatch, matchLimit);
            if (limited && (int64_t)op + (1 + kLastLiterals) + (mc + 240) / 255 > cap) return 0;
            LANES({ if (LANE == 0) dst[tokPos] = (uint8_t)((min_(lit, 15) << 4) | min_(mc, 15)); })
            if (mc >= 15) op = emit_len_ext(dst, op, mc - 15);

            const int ip = p + kMinMatch + mc;
            anchor = ip;
            if (ip >= lastProbe) break;                                        // lz4.c:1233

            // next batch: insert ip-2 (lz4.c:1236-1242), re-test ip (lz4.c:1255-1294), then search from ip+1 (lz4.c:1298)
            hasIns = true; insPos = ip - 2;
            hasRe = true;  rePos = ip;
            sBase = ip + 1; sIter = 0; width = 16;
            STAT(S_SEQ_GEN, 1);
            STAT(S_CYC_GEN, STAT_NOW() - tq0);
            }
        }
    }

    // last literals (lz4.c:1302-1329)
    {
        const int last = n - anchor;
        if (limited && (int64_t)op + last + 1 + (last + 255 - 15) / 255 > cap) return 0;
        if (last >= 15) {
            LANES({ if (LANE == 0) dst[op] = 0xF0; })
            op = emit_len_ext(dst, op + 1, last - 15);
        } else {
            LANES({ if (LANE == 0) dst[op] = (uint8_t)(last << 4); })
            op++;
        }
        wave_copy(dst + op, src + anchor, last);
        op += last;
    }
    STAT(S_CYC_TOTAL, STAT_NOW() - tBlock0);
    STAT(S_BLOCKS, 1);
    STAT_FLUSH();
    return op;
}

// LZ4_compress_fast_extState dispatch (lz4.c:1382-1403) + LZ4_compress_generic size screening (lz4.c:1360-1372)
DEV int wave_encode_block(const uint8_t* __restrict__ src, int n, uint8_t* __restrict__ dst, int cap, void* tab)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;
    const bool limited = !(cap >= compress_bound(n));
    if (n == 0) {
        if (limited && cap <= 0) return 0;
        LANES({ if (LANE == 0) dst[0] = 0; })
        return 1;
    }
    if (n < k64KLimit) return wave_encode_block_tt<true>(src, n, dst, cap, limited, tab);
    return wave_encode_block_tt<false>(src, n, dst, cap, limited, tab);
}

// ------------------------------------------------------------------------------------------ decoder
// Match copy inside the output: dst[op+i] = dst[op-offset+i], i < len, with LZ4's overlap semantics
// (offset < len replicates the pattern).  offset 0 zero-fills, like both liblz4 copy routines do
// (lz4.c:499-507, :2406-2414).
DEV void wave_copy_match(uint8_t* dst, int64_t op, int offset, int len)
{
    if (offset == 0) { LANES({ for (int i = LANE; i < len; i += 64) dst[op + i] = 0; }) return; }
    if (offset >= 64) {
        // a 64-byte chunk never reads what it writes; chunks are issued in order by one wave
        for (int base = 0; base < len; base += 64) {
            LANES({ const int i = base + LANE; if (i < len) dst[op + i] = dst[op - offset + i]; })
            WAVE_FENCE();
        }
        return;
    }
    // short period: every output byte is a copy of one of the `offset` bytes before op
    LANES({ for (int i = LANE; i < len; i += 64) dst[op + i] = dst[op - offset + (i % offset)]; })
}

// read_variable_length (lz4.c:1978-2014): sum of bytes until one is <255; -1 on running past `ilimit`.
DEV int64_t read_more_len(const uint8_t* src, int* ip, int ilimit, bool initialCheck)
{
    int64_t len = 0; uint32_t b;
    if (initialCheck && *ip >= ilimit) return -1;
    do {
        b = UNI(src[*ip]); (*ip)++;
        len += b;
        if (*ip > ilimit) return -1;
    } while (b == 255);
    return len;
}

// LZ4_decompress_safe, full block, no dictionary.  Returns decoded size or liblz4's negative error code
// -(input position)-1.  The reference's fast loop (>= 64 output bytes left) and safe loop reject at
// different points, so both sets of tests are reproduced (see oracle/plz4_oracle.c for the same shape).
DEV int wave_decode_block(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap)
{
    if (src == nullptr || cap < 0) return -1;
    const int iend = n;
    const int64_t oend = cap;
    int ip = 0; int64_t op = 0;
    if (cap == 0) return (n == 1 && UNI(src[0]) == 0) ? 0 : -1;
    if (n == 0) return -1;
    bool fast = (oend - op) >= 64;

    for (;;) {
        const uint32_t token = UNI(src[ip]); ip++;
        int64_t ll = token >> 4, ml; int offset; int64_t mpos;

        if (fast) {
            bool toSafeLit = false;
            if (ll == 15) {
                const int64_t a = read_more_len(src, &ip, iend - 15, true);
                if (a < 0) return -ip - 1;
                ll += a;
                if (op + ll > oend - 32 || (int64_t)ip + ll > iend - 32) toSafeLit = true;
            } else if (!(ip <= iend - 17)) {
                toSafeLit = true;
            }
            if (toSafeLit) { fast = false; goto safe_literals; }
            wave_copy(dst + op, src + ip, (int)ll);
            ip += (int)ll; op += ll;

            offset = UNI(ld16u(src + ip)); ip += 2;
            mpos = op - offset;
            ml = token & 15;
            if (ml == 15) {
                const int64_t a = read_more_len(src, &ip, iend - kLastLiterals + 1, false);
                if (a < 0) return -ip - 1;
                ml += a + kMinMatch;
                if (op + ml >= oend - 64) { fast = false; goto safe_match; }
            } else {
                ml += kMinMatch;
                if (op + ml >= oend - 64) { fast = false; goto safe_match; }
            }
            if (mpos < 0) return -ip - 1;                                   // checkOffset, lz4.c:2161
            WAVE_FENCE();
            wave_copy_match(dst, op, offset, (int)ml);
            WAVE_FENCE();
            op += ml;
            continue;
        }

        if (ll != 15 && ip < iend - 16 && op <= oend - 32) {                // shortcut, lz4.c:2230-2261
            wave_copy(dst + op, src + ip, (int)ll);
            op += ll; ip += (int)ll;
            ml = token & 15;
            offset = UNI(ld16u(src + ip)); ip += 2;
            mpos = op - offset;
            if (ml != 15 && offset >= 8 && mpos >= 0) {
                WAVE_FENCE();
                wave_copy_match(dst, op, offset, (int)ml + kMinMatch);
                WAVE_FENCE();
                op += ml + kMinMatch;
                continue;
            }
            goto match_len;
        }
        if (ll == 15) {
            const int64_t a = read_more_len(src, &ip, iend - 15, true);
            if (a < 0) return -ip - 1;
            ll += a;
        }
safe_literals:
        if (op + ll > oend - kMfLimit || (int64_t)ip + ll > iend - (2 + 1 + kLastLiterals)) {
            if ((int64_t)ip + ll != iend || op + ll > oend) return -ip - 1;  // lz4.c:2312-2318
            wave_copy(dst + op, src + ip, (int)ll);
            ip += (int)ll; op += ll;
            break;
        }
        wave_copy(dst + op, src + ip, (int)ll);
        ip += (int)ll; op += ll;
        offset = UNI(ld16u(src + ip)); ip += 2;
        mpos = op - offset;
        ml = token & 15;
match_len:
        if (ml == 15) {
            const int64_t a = read_more_len(src, &ip, iend - kLastLiterals + 1, false);
            if (a < 0) return -ip - 1;
            ml += a;
        }
        ml += kMinMatch;
safe_match:
        if (mpos < 0) return -ip - 1;                                       // lz4.c:2356
        if (op + ml > oend - kLastLiterals) return -ip - 1;                 // lz4.c:2421-2423
        WAVE_FENCE();
        wave_copy_match(dst, op, offset, (int)ml);
        WAVE_FENCE();
        op += ml;
    }
    return (int)op;
}

}  // namespace plz4
