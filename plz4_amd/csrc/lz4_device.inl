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
#define STAT_PTR     st_
extern __device__ unsigned long long plz4_stats[24];
#else
#define STAT(i, v)   do {} while (0)
#define STAT_DECL    do {} while (0)
#define STAT_NOW()   0ull
#define STAT_FLUSH() do {} while (0)
#define STAT_PTR     nullptr
#endif
enum { S_GRID = 0, S_GENERIC, S_MISORDER, S_SEQ_GRID, S_SEQ_GEN, S_TWINSTOP, S_SAT, S_LONGBACK, S_MEMLIT, S_WALKITER,
       S_CYC_TOTAL, S_CYC_LOAD, S_CYC_WALK, S_CYC_FIX, S_CYC_GEN, S_CYC_SAT, S_CYC_MEMLIT, S_BLOCKS, S_LANES_EXEC,
       S_DBATCH, S_DMEMB, S_DSEQ, S_DSEQ_ML15, S_DSEQ_LL15 };

namespace plz4 {

enum : int {
    kMinMatch = 4, kMfLimit = 12, kLastLiterals = 5, kMinLength = 13,
    k64KLimit = 65536 + kMfLimit - 1,            // lz4.c:710
    kMaxInput = 0x7E000000,                      // lz4.h:214
#if defined(PLZ4_EXP_TABBITS)
    kHashBytes = 4 << PLZ4_EXP_TABBITS           // EXPERIMENT (scripts/exp_smalltab.sh): a smaller table, valid but different output
#else
    kHashBytes = 16384                           // LZ4_HASHTABLESIZE, lz4.h:696
#endif
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


// The same digest with the payload staged through LDS: all 64 lanes fetch (16 bytes each, 2 KiB per chunk, the next chunk in
// flight while this one is consumed), the four accumulator lanes read their words back from LDS.  The four-lane version above
// waits for a memory round trip every eight rounds; here a round costs its arithmetic.  `lds`: 4 KiB owned by this wave.
DEV uint32_t wave_xxh32_staged(const uint8_t* p, int n, uint8_t* lds)
{
    if (n < 4096) return wave_xxh32(p, n);
    const int stripes = n >> 4;
    const int nChunks = (stripes + 127) >> 7;
    LV(uint32_t, acc); LV(v16u_t, r0); LV(v16u_t, r1);
    LANES({
        const int l = LANE & 3;
        acc[I_] = (l == 0) ? XP1 + XP2 : (l == 1) ? XP2 : (l == 2) ? 0u : 0u - XP1;
        for (int k = 0; k < 4; ++k) { r0[I_].w[k] = 0; r1[I_].w[k] = 0; }
        if (LANE < stripes)      r0[I_] = *(const v16u_t*)(p + 16 * LANE);
        if (64 + LANE < stripes) r1[I_] = *(const v16u_t*)(p + 16 * (64 + LANE));
    })
    for (int c = 0; c < nChunks; ++c) {
        uint8_t* buf = lds + ((c & 1) << 11);
        LANES({
            *(v16u_t*)(buf + 16 * LANE) = r0[I_];
            *(v16u_t*)(buf + 1024 + 16 * LANE) = r1[I_];
            const int s0 = (c + 1) * 128 + LANE;
            if (s0 < stripes)      r0[I_] = *(const v16u_t*)(p + 16 * (size_t)s0);
            if (s0 + 64 < stripes) r1[I_] = *(const v16u_t*)(p + 16 * (size_t)(s0 + 64));
        })
        LDS_FENCE();
        const int cnt = min_(128, stripes - c * 128);
        LANES({
            if (LANE < 4) {
                const uint8_t* q = buf + 4 * LANE;
                uint32_t a = acc[I_];
                int s = 0;
                for (; s + 8 <= cnt; s += 8) {
                    const uint32_t x0 = *(const uint32_t*)(q), x1 = *(const uint32_t*)(q + 16), x2 = *(const uint32_t*)(q + 32), x3 = *(const uint32_t*)(q + 48);
                    const uint32_t x4 = *(const uint32_t*)(q + 64), x5 = *(const uint32_t*)(q + 80), x6 = *(const uint32_t*)(q + 96), x7 = *(const uint32_t*)(q + 112);
                    a = rotl32(a + x0 * XP2, 13) * XP1; a = rotl32(a + x1 * XP2, 13) * XP1;
                    a = rotl32(a + x2 * XP2, 13) * XP1; a = rotl32(a + x3 * XP2, 13) * XP1;
                    a = rotl32(a + x4 * XP2, 13) * XP1; a = rotl32(a + x5 * XP2, 13) * XP1;
                    a = rotl32(a + x6 * XP2, 13) * XP1; a = rotl32(a + x7 * XP2, 13) * XP1;
                    q += 128;
                }
                for (; s < cnt; ++s) { a = rotl32(a + *(const uint32_t*)q * XP2, 13) * XP1; q += 16; }
                acc[I_] = a;
            }
        })
        LDS_FENCE();
    }
    uint32_t h = (uint32_t)n + rotl32(RL(acc, 0), 1) + rotl32(RL(acc, 1), 7) + rotl32(RL(acc, 2), 12) + rotl32(RL(acc, 3), 18);
    p += (size_t)stripes << 4;
    int rem = n - (stripes << 4);
    while (rem >= 4) { h = rotl32(h + UNI(ld32u(p)) * XP3, 17) * XP4; p += 4; rem -= 4; }
    while (rem)      { h = rotl32(h + (uint32_t)UNI(*p) * XP5, 11) * XP1; p++; rem--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

// Streaming xxHash32, seed 0 (xxh32.XXHZero, internal/pkg/xxh32/xxh32zero.go:58-86 Write, :204-235 Sum32; fed in block order by
// async/hash.go:99-111).  The state lives in device memory between calls; Write is what the wave does, Sum32 is finished by
// whoever reads the state back (it does not disturb it).  There is no combine operator for XXH32: the four chains are strictly
// sequential over the whole stream, so this is one wave with four busy lanes -- it runs beside the codec kernels, it is not a
// throughput kernel.
struct XxhStream { uint32_t v[4]; uint32_t totalLo, totalHi; uint32_t fill; uint8_t buf[16]; uint32_t pad; };

DEV void wave_xxh32_stream_reset(XxhStream* st)
{
    LANES({
        if (LANE < 4) st->v[LANE] = (LANE == 0) ? XP1 + XP2 : (LANE == 1) ? XP2 : (LANE == 2) ? 0u : 0u - XP1;
        if (LANE == 4) { st->totalLo = 0; st->totalHi = 0; st->fill = 0; st->pad = 0; }
        if (LANE >= 8 && LANE < 24) st->buf[LANE - 8] = 0;
    })
    WAVE_FENCE();
}

DEV void wave_xxh32_stream_update(XxhStream* st, const uint8_t* p, int64_t n)
{
    if (n <= 0) return;
    WAVE_FENCE();
    uint32_t fill = UNI(st->fill);
    {   // total += n
        const uint32_t lo = UNI(st->totalLo), hi = UNI(st->totalHi);
        const uint64_t t = (((uint64_t)hi << 32) | lo) + (uint64_t)n;
        LANES({ if (LANE == 0) { st->totalLo = (uint32_t)t; st->totalHi = (uint32_t)(t >> 32); } })
    }
    if ((int64_t)fill + n < 16) {                                  // still less than a stripe: keep the bytes (xxh32zero.go:64-69)
        LANES({ if (LANE < (int)n) st->buf[fill + LANE] = p[LANE]; })
        LANES({ if (LANE == 0) st->fill = fill + (uint32_t)n; })
        WAVE_FENCE();
        return;
    }
    LV(uint32_t, acc);
    LANES({ acc[I_] = st->v[LANE & 3]; })
    if (fill) {                                                    // complete the buffered stripe (:71-84)
        const int take = 16 - (int)fill;
        LANES({ if (LANE < take) st->buf[fill + LANE] = p[LANE]; })
        WAVE_FENCE();
        LANES({ if (LANE < 4) acc[I_] = rotl32(acc[I_] + ld32u(st->buf + 4 * LANE) * XP2, 13) * XP1; })
        p += take; n -= take;
    }
    const int64_t stripes = n >> 4;
    LANES({
        if (LANE < 4) {
            const uint8_t* q = p + 4 * LANE;
            uint32_t a = acc[I_];
            int64_t s = 0;
            for (; s + 8 <= stripes; s += 8) {                     // 8 loads in flight per lane
                const uint32_t x0 = ld32u(q), x1 = ld32u(q + 16), x2 = ld32u(q + 32), x3 = ld32u(q + 48);
                const uint32_t x4 = ld32u(q + 64), x5 = ld32u(q + 80), x6 = ld32u(q + 96), x7 = ld32u(q + 112);
                a = rotl32(a + x0 * XP2, 13) * XP1; a = rotl32(a + x1 * XP2, 13) * XP1;
                a = rotl32(a + x2 * XP2, 13) * XP1; a = rotl32(a + x3 * XP2, 13) * XP1;
                a = rotl32(a + x4 * XP2, 13) * XP1; a = rotl32(a + x5 * XP2, 13) * XP1;
                a = rotl32(a + x6 * XP2, 13) * XP1; a = rotl32(a + x7 * XP2, 13) * XP1;
                q += 128;
            }
            for (; s < stripes; ++s) { a = rotl32(a + ld32u(q) * XP2, 13) * XP1; q += 16; }
            acc[I_] = a;
        }
    })
    p += stripes << 4;
    const int tail = (int)(n & 15);
    LANES({
        if (LANE < 4) st->v[LANE] = acc[I_];
        if (LANE >= 8 && LANE - 8 < tail) st->buf[LANE - 8] = p[LANE - 8];
        if (LANE == 4) st->fill = (uint32_t)tail;
    })
    WAVE_FENCE();
}

// XXHZero.Sum32 over a copy of the state (plain code: host side of the C ABI, and the tests)
static inline uint32_t xxh32_stream_sum(const XxhStream& st)
{
    auto rol = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint64_t total = ((uint64_t)st.totalHi << 32) | st.totalLo;
    uint32_t h = (uint32_t)total;
    if (total >= 16) h += rol(st.v[0], 1) + rol(st.v[1], 7) + rol(st.v[2], 12) + rol(st.v[3], 18);
    else h += P5;
    uint32_t i = 0;
    for (; i + 4 <= st.fill; i += 4) {
        const uint32_t w = (uint32_t)st.buf[i] | ((uint32_t)st.buf[i + 1] << 8) | ((uint32_t)st.buf[i + 2] << 16) | ((uint32_t)st.buf[i + 3] << 24);
        h = rol(h + w * P3, 17) * P4;
    }
    for (; i < st.fill; ++i) h = rol(h + (uint32_t)st.buf[i] * P5, 11) * P1;
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

// ------------------------------------------------------------------------------------------ encoder
// Hash of the bytes at p (lz4.c:777-806, 64-bit little-endian build).
template <bool U16> DEV uint32_t seq_hash(uint64_t seq8)
{
    if (U16) return ((uint32_t)seq8 * 2654435761u) >> 19;                   // hash4, 13 bits
#if defined(PLZ4_EXP_TABBITS)
    return (uint32_t)(((seq8 << 24) * 889523592379ull) >> (64 - PLZ4_EXP_TABBITS));
#else
    return (uint32_t)(((seq8 << 24) * 889523592379ull) >> 52);              // hash5, 12 bits
#endif
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
// 24 bytes around a position x: [x-4, x+20), as the six dwords the two loads deliver (dwordx4 + dwordx2).  They are kept
// exactly as loaded -- no repacking into wider fields -- so that nothing has to touch them (and wait for them) before the
// point of use: w[0] = the 4 bytes before x (byte 3 is x-1), w[1..2] = [x, x+8), w[3..5] = [x+8, x+20).
struct Win24 { uint32_t w[6]; };
struct __attribute__((packed, may_alias)) v8u_t { uint32_t w[2]; };
DEV uint64_t win_seq(const Win24& p) { return (uint64_t)p.w[1] | ((uint64_t)p.w[2] << 32); }

template <bool kMayBeLow>
DEV Win24 load_win24(const uint8_t* src, int x)
{
    Win24 w;
    if (!kMayBeLow || x >= 4) {
        // three loads whose register tuples are the tuples the users want ([x, x+8) and [x+8, x+16) as 64-bit pairs): a
        // different split makes the compiler re-pack right behind the loads, i.e. wait for them on the spot
        const v8u_t  a = *(const v8u_t*)(src + x);                // [x, x+8)
        const v16u_t b = *(const v16u_t*)(src + x + 8);           // [x+8, x+24): 4 bytes more than needed, still inside the block
        w.w[0] = ld32u(src + x - 4);
        w.w[1] = a.w[0]; w.w[2] = a.w[1]; w.w[3] = b.w[0]; w.w[4] = b.w[1]; w.w[5] = b.w[2];
    } else {                                                        // candidates in the first 4 bytes of the block
        uint32_t bk = 0;
        for (int i = 0; i < x; ++i) bk |= (uint32_t)src[x - 1 - i] << (8 * (3 - i));
        w.w[0] = bk;
        for (int k = 0; k < 5; ++k) w.w[1 + k] = ld32u(src + x + 4 * k);
    }
    return w;
}

// equal bytes of [x+4, x+20) in two windows: 0..16
DEV int win_fwd(const Win24& p, const Win24& c)
{
    const uint32_t x2 = p.w[2] ^ c.w[2], x3 = p.w[3] ^ c.w[3], x4 = p.w[4] ^ c.w[4], x5 = p.w[5] ^ c.w[5];
    int n = x5 ? 12 + (__builtin_ctz(x5) >> 3) : 16;
    n = x4 ? 8 + (__builtin_ctz(x4) >> 3) : n;
    n = x3 ? 4 + (__builtin_ctz(x3) >> 3) : n;
    n = x2 ? (__builtin_ctz(x2) >> 3) : n;
    return n;
}
// equal bytes going backwards from x-1: 0..4
DEV int win_bck(const Win24& p, const Win24& c)
{
    const uint32_t y = p.w[0] ^ c.w[0];
    return y ? (__builtin_clz(y) >> 3) : 4;
}

#if defined(PLZ4_EMU)
static inline uint32_t lds_max_rtn(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
static inline void     lds_min(uint32_t* p, uint32_t v)     { if (v < *p) *p = v; }
static inline uint32_t lds_add_rtn(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
static inline void     lds_max(uint32_t* p, uint32_t v)     { if (v > *p) *p = v; }
#else
__device__ __forceinline__ uint32_t lds_max_rtn(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
__device__ __forceinline__ void     lds_min(uint32_t* p, uint32_t v)     { atomicMin(p, v); }
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
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
//         parser skipped, atomic-max for the ones it executed) to exactly the sequential result.  The batch only
//         records its sequences; they are written one batch later, in the gap between "candidate bytes requested"
//         and "candidate bytes needed" (emit_compute / emit_store), so the writer hides that memory latency.
//   GENERIC (everything else: byU16, block head/tail, stride > 1): one lane per probe of the search loop,
//         closed-form probe positions, plain store + read-back collision detection, one sequence per batch.
//
// Table entries (byU32): liblz4 stores the position.  Positions of a block of <= 4 MiB need 22 bits, so the
// private LDS copy keeps (position << 10) | tag, tag = 10 hash bits of the 4 bytes at that position.  Ordering by
// entry == ordering by position, and a candidate whose tag differs cannot pass the 4-byte test, so its bytes are
// never fetched: that removes most of the random memory reads.  An empty liblz4 slot is position 0, therefore
// the table is initialised with position 0's entry, not with zero.  Larger blocks (raw block API) run untagged.
DEV uint32_t seq_tag(uint32_t seq4) { return (seq4 * 2246822519u) >> 22; }

// kExt (the streaming flavours of lz4.c:930-1338 that plz4 reaches through LZ4_compress_fast_continue, see ExtEnc below): the
// block has an external segment of `pfx` bytes lying IMMEDIATELY BEFORE it in memory, `src` points at the segment's first
// byte, positions count from there, the block is [pfx, n).  Table entries then hold liblz4's own index, position + delta
// (delta = 64 KiB - pfx once a dictionary was loaded), so that an empty slot -- index 0 -- is what it is for liblz4: the
// segment's first byte when the segment is a full 64 KiB (noDictIssue), an invalid candidate otherwise (dictSmall,
// lz4.c:1085-1087).  A candidate inside the segment needs no second code path: its bytes are at src + position like any
// other, and a match that runs from the segment into the block is one run of contiguous bytes (lz4.c:1168-1180); the one
// difference kept is that a candidate in the block cannot be extended backwards into the segment (lowLimit, lz4.c:1065-1079).
struct ExtEnc {
    int pfx;                       // bytes of the segment in front of the block
    uint32_t delta;                // liblz4 index of position 0
    int prime;                     // 0: empty table; 1: LZ4_loadDict over the segment (lz4.c:1587-1646); 2: copy `table`
    const uint32_t* table;         // prime 2: the dictionary context's table (liblz4 indices, LZ4_loadDictSlow; clz4.go:96-120)
};

template <bool U16, bool kExt = false>
DEV int wave_encode_block_tt(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst,
                             const int cap, const bool limited, void* tab, const ExtEnc x = ExtEnc{0, 0u, 0, nullptr})
{
    const int      P0      = kExt ? x.pfx : 0;                       // first position of the block
    const uint32_t delta   = kExt ? x.delta : 0u;
    const int      sh      = kExt ? (((uint32_t)n + delta <= (1u << 23)) ? 9 : 0) : ((!U16 && n <= (1 << 22)) ? 10 : 0);
    const uint32_t tagMask = (1u << sh) - 1u;
    // fresh table per block (LZ4_initStream, lz4.c:1384): every slot = "index 0"
    {
        const uint32_t e0 = (sh && n >= 4 && delta == 0u) ? (seq_tag(UNI(ld32u(src))) & tagMask) : 0u;
        uint32_t* t = (uint32_t*)tab;
        if (kExt && x.prime == 2) {
            // LZ4_memcpy(streamPtr, dictCtx) (lz4.c:1762-1768): liblz4 indices, tagged here with the bytes they point at
            LANES({ for (int i = LANE; i < kHashBytes / 4; i += 64) {
                const uint32_t idx = x.table[i];
                t[i] = (idx >= delta && sh) ? ((idx << sh) | (seq_tag(ld32u(src + (idx - delta))) & tagMask)) : (idx << sh);
            } })
        } else {
            LANES({ for (int i = LANE; i < kHashBytes / 4; i += 64) t[i] = e0; })
        }
        if (kExt && x.prime == 1) {
            LDS_FENCE();
            // LZ4_loadDict (lz4.c:1621-1628): every 3rd position of the segment, later entries overwrite earlier ones ==
            // the largest index per slot
            const int cnt = (x.pfx - 8) / 3 + 1;
            LANES({ for (int k = LANE; k < cnt; k += 64) {
                const uint64_t s8 = ld64u(src + 3 * k);
                lds_max(&t[seq_hash<false>(s8)], (((uint32_t)(3 * k) + delta) << sh) | (seq_tag((uint32_t)s8) & tagMask));
            } })
        }
    }
    LDS_FENCE();

    const int lastProbe  = n - kMfLimit + 1;      // mflimitPlusOne (lz4.c:963)
    const int matchLimit = n - kLastLiterals;     // lz4.c:964
    int anchor = P0, op = 0;
    STAT_DECL;
    const unsigned long long tBlock0 = STAT_NOW();
    (void)tBlock0;

    if (n - P0 >= kMinLength) {
        // Parser state between batches.
        int  insPos  = P0; bool hasIns = true;     // pending table insert ("First Byte" lz4.c:1005-1010; ip-2 lz4.c:1236-1242)
        int  rePos   = P0; bool hasRe  = false;    // pending immediate re-test at ip after a match (lz4.c:1255-1294)
        int  sBase   = P0 + 1; int sIter = 0;      // search started at sBase; next un-probed probe number
        int  width   = 16;                         // generic batches: 16 lanes first, 64 when a search drags on
        LV(Win24, Pn); int prefBase = -1;          // next window's bytes, requested one batch ahead
        LV(Win24, Pc);                             // this window's bytes (kept: a wrong twin is repaired from them)
        LANES({ for (int k = 0; k < 6; ++k) Pn[I_].w[k] = 0; Pc[I_] = Pn[I_]; })

        // A grid batch does not write its sequences right away: they are kept here and written while the NEXT batch's
        // candidate bytes are on their way from memory (or before anything else touches the output).
        uint64_t pMm = 0; int pBase = 0, pAnchor0 = 0, pCur0 = 0; bool pRe0 = false;
        LV(int, pFwd); LV(int, pBck); LV(uint32_t, pR); LV(uint32_t, pLit8); LV(int, pStA); LV(int, pHasPm);
        LANES({ pFwd[I_] = 0; pBck[I_] = 0; pR[I_] = 0; pLit8[I_] = 0; pStA[I_] = 0; pHasPm[I_] = 0; })
        // how far a candidate at position r may be extended backwards: to the start of what it lies in (lz4.c:1065-1079)
        auto back_room = [&](uint32_t r) -> int { return (int)r - ((kExt && (int)r >= P0) ? P0 : 0); };
        // ---- 4. emit every recorded sequence of the pending batch at once, in two halves: emit_compute works out where
        // everything goes (no stores; false: liblz4 would return 0, limitedOutput), emit_store issues the stores.  Between
        // the two the grid batch waits for its candidate bytes: with the stores behind that wait, it is a wait for those loads
        // only (the count of younger memory operations is fixed), not for whatever the writer has just stored.
        uint64_t sMm = 0; int sWin = 0, mlDst = 0, mlSrc = 0, mlLen = 0;
        LV(int, lit); LV(int, mcT); LV(int, extL); LV(int, extM); LV(int, tok); LV(int, litAt);
        LANES({ lit[I_] = 0; mcT[I_] = 0; extL[I_] = 0; extM[I_] = 0; tok[I_] = 0; litAt[I_] = -1; })
        auto emit_compute = [&]() -> bool {
            if (!pMm) return true;
            const uint64_t mm = pMm; const int base = pBase; const int anchor0 = pAnchor0; const int cur0 = pCur0; const bool re0 = pRe0;
            pMm = 0;
            LV(int, anc); LV(int, size); LV(int, pS); LV(int, litDst);
            LANES({ anc[I_] = pHasPm[I_] ? base + pStA[I_] : anchor0; })
            {   // catch-up that may go past the 4 speculative bytes (rare): finish it now that the anchors are final
                const uint64_t mmB = mm;
                uint64_t deep = BALLOT(((mmB >> LANE) & 1) && pBck[I_] == 4 && min_(base + LANE - anc[I_], back_room(pR[I_])) > 4);
                for (; deep; deep &= deep - 1) {
                    const int w = ctz64(deep);
                    const int p0 = base + w, c0 = (int)RL(pR, w);
                    const int maxBack = min_(p0 - RL(anc, w), back_room((uint32_t)c0));
                    STAT(S_LONGBACK, 1);
                    WL(pBck, w, 4 + wave_common_back(src, p0 - 4, c0 - 4, maxBack - 4));
                }
            }
            LANES({
                const int q = base + LANE;
                const int bk = min_(pBck[I_], min_(q - anc[I_], back_room(pR[I_])));     // lz4.c:1105-1109 (0 for a re-test)
                pS[I_]   = q - bk;
                lit[I_]  = q - bk - anc[I_];
                mcT[I_]  = pFwd[I_] + bk;
                extL[I_] = lit[I_] >= 15 ? (lit[I_] - 15) / 255 + 1 : 0;
                extM[I_] = mcT[I_] >= 15 ? (mcT[I_] - 15) / 255 + 1 : 0;
                size[I_] = 1 + extL[I_] + lit[I_] + 2 + extM[I_];
            })
            const uint64_t mmL = mm;
            {   // output cursor per sequence, in order: a prefix sum over the match lanes
                LV(int, acc);
                LANES({ acc[I_] = ((mmL >> LANE) & 1) ? size[I_] : 0; })
                SCAN_INCL(acc);
                const int op0 = op;
                LANES({ tok[I_] = op0 + acc[I_] - size[I_]; })
                op = op0 + RL(acc, 63);
            }
            // lz4.c:1114-1117 and :1187-1210, per sequence.  Every left-hand side is <= (end of this batch's output) + 7,
            // so the exact per-sequence test is only needed near the end of the capacity.
            if (limited && (int64_t)op + 8 > cap) {
                const uint64_t over = BALLOT(((mmL >> LANE) & 1) &&
                    ((!(pHasPm[I_] ? (LANE == pStA[I_]) : (re0 && LANE == cur0)) &&
                      (int64_t)tok[I_] + 1 + lit[I_] + (2 + 1 + kLastLiterals) + lit[I_] / 255 > cap) ||
                     ((int64_t)tok[I_] + 1 + extL[I_] + lit[I_] + 2 + (1 + kLastLiterals) + (mcT[I_] + 240) / 255 > cap)));
                if (over) return false;
            }
            // literals: every lane looks up the next recorded match at or after it
            LANES({ litDst[I_] = tok[I_] + 1 + extL[I_]; })
            LANES({
                const uint64_t ahead = mmL >> LANE;
                const int m = ahead ? LANE + ctz64(ahead) : LANE;   // all lanes take part in the exchange
                const int a = SHFL(anc, m), pe = SHFL(pS, m), ld = SHFL(litDst, m);
                const int q = base + LANE;
                litAt[I_] = (ahead && q >= a && q < pe) ? ld + (q - a) : -1;
            })
            {   // the part of the first run that lies before this window comes from memory
                const int w0 = ctz64(mm);
                const int a0 = RL(anc, w0);
                mlLen = 0;
                if (a0 < base) { STAT(S_MEMLIT, 1); mlDst = RL(litDst, w0); mlSrc = a0; mlLen = min_(RL(lit, w0), base - a0); }
            }
            sMm = mm; sWin = base;
            return true;
        };
        auto emit_store = [&]() {
            if (!sMm) return;
            const uint64_t mmL = sMm; const int base = sWin;
            sMm = 0;
            LANES({
                if ((mmL >> LANE) & 1) {
                    const int t = tok[I_];
                    dst[t] = (uint8_t)((min_(lit[I_], 15) << 4) | min_(mcT[I_], 15));
                    int o = t + 1;
                    if (extL[I_]) { int rest = lit[I_] - 15; for (; rest >= 255; rest -= 255) dst[o++] = 255; dst[o++] = (uint8_t)rest; }
                    o += lit[I_];
                    st16u(dst + o, (uint16_t)((uint32_t)(base + LANE) - pR[I_]));
                    o += 2;
                    if (extM[I_]) { int rest = mcT[I_] - 15; for (; rest >= 255; rest -= 255) dst[o++] = 255; dst[o++] = (uint8_t)rest; }
                }
                if (litAt[I_] >= 0) dst[litAt[I_]] = (uint8_t)pLit8[I_];
            })
            if (mlLen) wave_copy(dst + mlDst, src + mlSrc, mlLen);
        };
        auto flush_emit = [&]() -> bool { if (!emit_compute()) return false; emit_store(); return true; };

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
                    LV(uint32_t, ent); LV(uint32_t, rent);           // my table entry / the entry I displaced
                    LV(int, hit); LV(int, fwd); LV(int, bck); LV(int, eLane); LV(int, cand); LV(Win24, Cw);
                    // ---- 1. every lane: its window, the table exchange, the request for its candidate's window
                    LANES({
                        const int q = base + LANE;
                        const int isIns = hasIns && q == insPos;
                        act[I_] = isIns || q >= probeStart;
                        hit[I_] = 0; fwd[I_] = 0; bck[I_] = 0; r[I_] = 0; h[I_] = 0; ent[I_] = 0; rent[I_] = 0; cand[I_] = 0;
                        const Win24 P = (base == prefBase) ? Pn[I_] : load_win24<false>(src, q);   // every lane: its byte may be a pending literal
                        Pc[I_] = P;
                        lit8[I_] = P.w[1] & 0xFF;
                        if (act[I_]) {
                            h[I_] = seq_hash<false>(win_seq(P));
                            const uint32_t tg = seq_tag(P.w[1]) & tagMask;
                            ent[I_]  = (((uint32_t)q + delta) << sh) | tg;
                            rent[I_] = lds_max_rtn(&T[h[I_]], ent[I_]);
                            const uint32_t ri = rent[I_] >> sh;
                            const bool live = !kExt || ri >= delta;                      // dictSmall: an index below the segment (lz4.c:1085-1087)
                            r[I_]    = live ? ri - delta : 0u;
                            cand[I_] = !isIns && live && r[I_] < (uint32_t)q && r[I_] + kMaxDist >= (uint32_t)q && (rent[I_] & tagMask) == tg;
                            if (cand[I_] && r[I_] >= 4) Cw[I_] = load_win24<false>(src, (int)r[I_]);
                        }
                        if (base + 160 <= n) Pn[I_] = load_win24<false>(src, q + 64);   // request the next window now, use it next batch
                    })
                    prefBase = (base + 160 <= n) ? base + 64 : -1;
                    // the previous batch's sequences are worked out while those bytes travel, and stored behind the wait for them
                    if (!emit_compute()) return 0;
                    LANES({
                        if (act[I_] && cand[I_] && Cw[I_].w[1] == Pc[I_].w[1]) {
                            hit[I_] = 1;
                            fwd[I_] = win_fwd(Pc[I_], Cw[I_]);
                            bck[I_] = min_(win_bck(Pc[I_], Cw[I_]), back_room(r[I_]));
                        }
                        eLane[I_] = LANE + kMinMatch + fwd[I_];     // lane index just past a match that starts here
                    })
                    emit_store();
                    LDS_FENCE();
                    // LDS atomics on one slot are expected to resolve in ascending lane order (then r is the
                    // nearest earlier twin or the pre-batch value).  Any other order shows up as r >= q somewhere.
                    // (A candidate in the first 4 bytes of the block has no 4 bytes before it to load: equally rare, same way out.)
                    const uint64_t misorder = BALLOT(act[I_] && (r[I_] >= (uint32_t)(base + LANE) || (cand[I_] && r[I_] < 4)));
                    if (misorder) {
                        STAT(S_MISORDER, 1);
                        LANES({ if (act[I_]) lds_min(&T[h[I_]], rent[I_]); })  // min over a slot's group == its pre-batch value
                        LDS_FENCE();
                        goto generic_batch;
                    }
                    {
                        uint64_t hits  = BALLOT(hit[I_]);
                        const unsigned long long tg1 = STAT_NOW(); (void)tg1;
                        STAT(S_CYC_LOAD, tg1 - tg0);
                        uint64_t twins = BALLOT(act[I_] && r[I_] >= (uint32_t)firstPos);          // earlier twin inside this batch
                        uint64_t cwValid = BALLOT(act[I_] && cand[I_]);       // lanes whose Cw holds the window of their rent's position
                        const uint64_t anyTwins = twins;                                          // (bits of repaired lanes get cleared below)
                        uint64_t specialLeft = BALLOT(hit[I_] && fwd[I_] == 16);           // longer than the speculative window: the hop needs its end
                        const int  cur0 = probeStart - base;                  // first probe lane (>= 64: none in this batch)
                        const bool re0  = hasRe;
                        const uint64_t insBit0 = hasIns ? (1ull << (insPos - base)) : 0;
                        int lim0 = sBase + 65 - base; if (lim0 > 63) lim0 = 63;   // probe number <= 65 keeps the stride at 1

                        // ---- 2. scalar hop over the recorded matches only; everything else is derived per lane
                        uint64_t mm = 0;           // executed match lanes
                        int      eL = 0;           // end lane of the last executed match
                        uint64_t E = 0;            // executed lanes (probes + inserts)
                        int      Send = 0;
                        bool     finished = false;
                        LV(int, stA); LV(int, hasPm);
                        // after a twin repair at lane b the walk is redone from b only: what it did below b does not depend on b
                        uint64_t keep = 0; int resume = -1;
                        for (;;) {
                            const unsigned long long th0 = STAT_NOW(); (void)th0;
                            LV(int, nextHit);      // first recorded match at or after the end of the match that starts here (64: none)
                            {
                                const uint64_t hitsL = hits;
                                LANES({
                                    const uint64_t ah = (eLane[I_] < 64) ? (hitsL >> eLane[I_]) : 0;
                                    nextHit[I_] = ah ? eLane[I_] + ctz64(ah) : 64;
                                })
                            }
                            mm = keep; eL = 0; finished = false;
                            const unsigned long long th1 = STAT_NOW(); (void)th1;
                            STAT(S_DBATCH, th1 - th0);
                            int w = 64;
                            {
                                const int start = resume >= 0 ? resume : cur0;     // lane b was a probe: the parser is searching there
                                if (start < 64) { const uint64_t hm = hits & (~0ull << start); if (hm) w = ctz64(hm); }
                            }
                            if (w < 64 && (mm != 0 || w <= lim0)) {               // (the stride limit concerns the first match only)
                                // Hops come four to a branch (a taken branch costs more than the hop).  A hop is: mark the lane,
                                // fetch its successor.  Once the walk has ended (w == 64) the remaining hops of a group only
                                // touch bit 0 of the mask, which no hop but the very first can legitimately set.
                                // Matches longer than the speculative window are not known to the hop: it walks through them
                                // as if they ended there, and the first one it touched is put right afterwards.
                                for (;;) {
                                    const uint64_t low = (mm & 1) | (w == 0 ? 1ull : 0ull);
                                    do {
                                        STAT(S_WALKITER, 1);
                                        for (int u = 0; u < 4; ++u) {
                                            const int n1 = RL(nextHit, w & 63);
                                            mm |= 1ull << (w & 63);
                                            w = (w < 64) ? n1 : 64;
                                        }
                                    } while (w < 64);
                                    mm = (mm & ~1ull) | low;
                                    const uint64_t sp = mm & specialLeft;
                                    if (!sp) break;
                                    const int ws = ctz64(sp);
                                    mm &= (2ull << ws) - 1;                       // what the walk did after it is void
                                    const int p0 = base + ws, c0 = (int)RL(r, ws);
                                    int mc0 = (int)RL(fwd, ws);
                                    if (mc0 == 16) { STAT(S_SAT, 1); mc0 += wave_common_len(src, p0 + 20, c0 + 20, matchLimit); WL(fwd, ws, mc0); }
                                    specialLeft &= ~(1ull << ws);
                                    const int e1 = ws + kMinMatch + mc0;
                                    WL(eLane, ws, e1);
                                    if (base + e1 >= lastProbe) { finished = true; break; }   // lz4.c:1233 (only a long match gets there)
                                    const uint64_t hm = (e1 < 64) ? (hits & (~0ull << e1)) : 0;
                                    w = hm ? ctz64(hm) : 64;
                                    if (w >= 64) break;
                                }
                            }
                            if (mm) eL = RL(eLane, 63 - __builtin_clzll(mm));
                            // ---- 3. which lanes did the sequential parser execute
                            const unsigned long long th2 = STAT_NOW(); (void)th2;
                            STAT(S_DMEMB, th2 - th1);
                            Send = mm ? 64 : min_(64, lim0 + 1);
                            const uint64_t mmL = mm; const int SendL = Send;
                            // end of the last executed match below each lane: ends grow along the walk, so an exclusive
                            // prefix maximum over the match lanes (DPP, no LDS round trip) is that value
                            LANES({ stA[I_] = ((mmL >> LANE) & 1) ? eLane[I_] : 0; })
                            SCAN_MAX_EXCL(stA);
                            LANES({
                                hasPm[I_] = stA[I_] > 0;
                                stA[I_]   = hasPm[I_] ? stA[I_] : cur0;                    // where probing resumed before me
                            })
                            const uint64_t probes = BALLOT(LANE >= stA[I_] && LANE < SendL && LANE >= cur0);
                            E = probes | insBit0 | BALLOT(hasPm[I_] && stA[I_] == LANE + 2);   // + the ip-2 inserts (lz4.c:1236-1242)
                            const unsigned long long th3 = STAT_NOW(); (void)th3;
                            STAT(S_DSEQ, th3 - th2);
                            if (twins & probes) {
                                // A probe whose candidate is an earlier lane of this batch is only right if that lane was executed.
                                // Otherwise the sequential parser saw what that lane displaced (or what *it* displaced, ...):
                                // repair the first such lane in place and redo the (cheap) hop.
                                const uint64_t EL = E;
                                const uint64_t bad = twins & probes & BALLOT(!((EL >> (((int)r[I_] - base) & 63)) & 1));
                                if (bad) {
                                    STAT(S_TWINSTOP, 1);
                                    const int b = ctz64(bad);
                                    uint32_t ce = RL(rent, b);                          // entry lane b displaced
                                    int tl = -1;                                        // the lane whose displaced entry `ce` is
                                    for (;;) {
                                        const uint32_t ci = ce >> sh;
                                        if (ci < (uint32_t)firstPos + delta) break;     // a pre-batch entry
                                        const int t = (int)(ci - delta) - base;
                                        if ((E >> t) & 1) break;                        // an executed lane of this batch
                                        ce = RL(rent, t);                               // a skipped lane: what it displaced
                                        tl = t;
                                    }
                                    const bool live = !kExt || (ce >> sh) >= delta;
                                    const uint32_t cp = live ? (ce >> sh) - delta : 0u, qb = (uint32_t)(base + b);
                                    int nhit = 0, nfwd = 0, nbck = 0;
                                    if (live && cp + kMaxDist >= qb && (ce & tagMask) == (RL(ent, b) & tagMask)) {
                                        Win24 Pb; for (int k = 0; k < 6; ++k) Pb.w[k] = RLF(Pc, w[k], b);
                                        Win24 Cn;
                                        if (cp >= (uint32_t)base) { const int t = (int)cp - base; for (int k = 0; k < 6; ++k) Cn.w[k] = RLF(Pc, w[k], t); }
                                        // a pre-batch entry was displaced by lane tl, and if that lane took it for a candidate (a twin
                                        // usually repeats the very same bytes) its window already holds what is needed: no memory round trip
                                        else if (tl >= 0 && ((cwValid >> tl) & 1) && cp >= 4) { STAT(S_MEMLIT, 0); for (int k = 0; k < 6; ++k) Cn.w[k] = RLF(Cw, w[k], tl); }
                                        else { Cn = load_win24<true>(src, (int)cp); for (int k = 0; k < 6; ++k) Cn.w[k] = UNI(Cn.w[k]); }
                                        if (Cn.w[1] == Pb.w[1]) { nhit = 1; nfwd = win_fwd(Pb, Cn); nbck = min_(win_bck(Pb, Cn), back_room(cp)); }
                                    }
                                    WL(rent, b, ce); WL(r, b, cp); WL(hit, b, nhit); WL(fwd, b, nfwd); WL(bck, b, nbck);
                                    WL(eLane, b, b + kMinMatch + nfwd);
                                    const uint64_t bit = 1ull << b;
                                    hits = nhit ? (hits | bit) : (hits & ~bit);
                                    specialLeft = (nhit && nfwd == 16) ? (specialLeft | bit) : (specialLeft & ~bit);
                                    twins &= ~bit;
                                    cwValid &= ~bit;                                    // lane b's Cw no longer belongs to its (new) rent
                                    keep = mm & (bit - 1); resume = b;
                                    STAT(S_DSEQ_ML15, STAT_NOW() - th3);
                                    continue;
                                }
                            }
                            break;
                        }
                        // lz4.c:1233: a match that ends at or past the last probe position ends the block.  Decided here, from the
                        // final walk: the pass that finished a long match may have been redone after a twin repair, and the redo
                        // sees that match as an ordinary one.
                        finished = (mm != 0) && (base + eL >= lastProbe);
                        const unsigned long long tg2 = STAT_NOW(); (void)tg2;
                        STAT(S_CYC_WALK, tg2 - tg1);
                        STAT(S_SEQ_GRID, __builtin_popcountll(mm));
                        STAT(S_LANES_EXEC, __builtin_popcountll(E));

                        // ---- 4. hand the recorded sequences to the deferred writer (flush_emit)
                        if (mm) {
                            pMm = mm; pBase = base; pAnchor0 = anchor; pCur0 = cur0; pRe0 = re0;
                            LANES({ pFwd[I_] = fwd[I_]; pBck[I_] = bck[I_]; pR[I_] = r[I_]; pLit8[I_] = lit8[I_];
                                    pStA[I_] = stA[I_]; pHasPm[I_] = hasPm[I_]; })
                            anchor = base + eL;
                        }
                        const unsigned long long tg3 = STAT_NOW(); (void)tg3;
                        STAT(S_CYC_SAT, tg3 - tg2);
                        if (finished) break;

                        // ---- 5. parser state after this batch
                        hasIns = false;
                        if (cur0 < 64) hasRe = false;
                        if (mm) {
                            sBase = base + eL + 1; sIter = 0;
                            if (eL - 2 >= 64) { hasIns = true; insPos = base + eL - 2; }
                            if (eL >= Send) { hasRe = true; rePos = base + eL; }       // its re-test is not executed in this batch
                            else sIter = (base + Send) - sBase;                         // lanes eL..Send-1 missed (eL was the re-test)
                        } else if (cur0 < 64) {
                            if (Send > cur0) sIter = (base + Send) - sBase;             // lanes cur0..Send-1 missed
                            else if (re0) { hasRe = true; rePos = base + cur0; }
                        }

                        // ---- 6. patch the table to the sequential result
                        const uint64_t EL2 = E;
                        LANES({ if (act[I_] && !((EL2 >> LANE) & 1)) lds_min(&T[h[I_]], rent[I_]); })
                        if (anyTwins) { LDS_FENCE(); LANES({ if ((EL2 >> LANE) & 1) lds_max(&T[h[I_]], ent[I_]); }) }
                        LDS_FENCE();
                        STAT(S_CYC_FIX, STAT_NOW() - tg3);
                        width = 64;
                        continue;
                    }
                }
            }
generic_batch:
            // ================================================================ GENERIC batch
            {
            if (!flush_emit()) return 0;
            STAT(S_GENERIC, 1);
            const unsigned long long tq0 = STAT_NOW(); (void)tq0;
            const int pre = (hasIns ? 1 : 0) + (hasRe ? 1 : 0);
            LV(int, q); LV(uint32_t, h); LV(uint32_t, old); LV(uint32_t, rb); LV(uint32_t, lo4);
            LV(uint32_t, ent); LV(uint32_t, oldE);
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
                    ent[I_]  = (((uint32_t)q[I_] + delta) << sh) | (seq_tag(lo4[I_]) & tagMask);
                    oldE[I_] = tab_get<U16>(tab, h[I_]);
                    old[I_]  = oldE[I_] >> sh;                                   // liblz4's index; a position once `delta` is taken off
                }
            })
            LDS_FENCE();
            LANES({ if (LANE < nproc) tab_put<U16>(tab, h[I_], ent[I_]); })
            LDS_FENCE();
            LANES({ rb[I_] = (LANE < nproc) ? tab_get<U16>(tab, h[I_]) : ent[I_]; })
            // candidate test (lz4.c:1090-1099); the insert-only lane never matches
            LANES({
                if (LANE < nproc && !(hasIns && LANE == 0)) {
                    const uint32_t cur = (uint32_t)q[I_];
                    const bool live = !kExt || old[I_] >= delta;                 // dictSmall (lz4.c:1085-1087)
                    old[I_] = live ? old[I_] - delta : 0u;
                    if (live && (U16 || old[I_] + kMaxDist >= cur) && ((oldE[I_] ^ ent[I_]) & tagMask) == 0)
                        hit[I_] = (ld32u(src + old[I_]) == lo4[I_]);
                }
            })
            const uint64_t losers = BALLOT((LANE < nproc) && rb[I_] != ent[I_]);
            const uint64_t hits   = BALLOT(hit[I_]);

            // collision-free prefix: lanes before the first lane that lost a same-slot store (lane 0 never has an earlier twin)
            int safe = nproc;
            if (losers) safe = min_(nproc, max_(ctz64(losers), 1));
            const uint64_t hitsSafe = hits & ((safe >= 64) ? ~0ull : ((1ull << safe) - 1));
            const int keep = hitsSafe ? ctz64(hitsSafe) + 1 : safe;          // lanes [0,keep) are really executed

            // ---- make the table exactly what the sequential parser would have left
            LANES({ if (LANE >= keep && LANE < nproc) tab_put<U16>(tab, h[I_], oldE[I_]); })
            if (losers) { LDS_FENCE(); LANES({ if (LANE < keep) tab_put<U16>(tab, h[I_], ent[I_]); }) }
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
                const int back = wave_common_back(src, p, c, min_(p - anchor, back_room((uint32_t)c)));
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
        if (!flush_emit()) return 0;          // a grid batch may have been the last one
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

// ------------------------------------------------------------------------------------------ encoder, dictionary modes
// LZ4_compress_fast_continue as plz4 drives it (clz4.go:160-179 StreamIndieCtx, :224-248 StreamLinkedCtx; lz4.c:1707-1783):
// always byU32 + limitedOutput.  Which flavour of lz4.c:930-1338 runs depends on how the stream was primed:
enum : int {
    kDictFreshPrefix = 0,   // linked block 0, no dictionary: withPrefix64k, empty prefix, indices start at 0
    kDictLoad        = 1,   // linked block k>0: LZ4_loadDict(previous block's last <=64 KiB) then usingExtDict (dictSmall if < 64 KiB)
    kDictCtxCopy     = 2,   // dictionary context attached, n > 4 KiB: its table is copied in, usingExtDict, noDictIssue
    kDictCtxLookup   = 3,   // dictionary context attached, n <= 4 KiB: usingDictCtx (own empty table + the context's table)
    kDictNonePrefix  = 4    // a dictionary shorter than 8 bytes is dropped by LZ4_loadDict: withPrefix64k, indices start at 64 KiB, dictSmall
};
struct DictEnc {
    const uint8_t*  dict;       // the (<= 64 KiB) dictionary bytes in device memory; nullptr in modes 0 and 4
    int             dictSize;
    int             mode;
    const uint32_t* dictTable;  // modes 2 and 3: the 4096-entry table LZ4_loadDictSlow builds (prepared on the host, clz4.go:96-120)
};

// 8 bytes at p of which only `valid` may be touched
DEV uint64_t ld64p_guard(const uint8_t* p, int valid)
{
    if (valid >= 8) return ld64u(p);
    uint64_t v = 0;
    for (int b = 0; b < valid; ++b) v |= (uint64_t)p[b] << (8 * b);
    return v;
}
// number of equal bytes of a[0..) and b[0..), at most maxLen (== LZ4_count with pInLimit = a + maxLen)
DEV int wave_count_ptr(const uint8_t* a, const uint8_t* b, int maxLen)
{
    int total = 0;
    for (;;) {
        LV(uint64_t, d);
        LANES({
            const int off = total + 8 * LANE;
            const int valid = maxLen - off;
            if (valid <= 0) d[I_] = ~0ull;
            else {
                uint64_t x = ld64p_guard(a + off, valid) ^ ld64p_guard(b + off, valid);
                if (valid < 8) x |= ~0ull << (8 * valid);
                d[I_] = x;
            }
        })
        const uint64_t stop = BALLOT(d[I_] != 0);
        if (stop) { const int l = ctz64(stop); return total + 8 * l + (ctz64(RL(d, l)) >> 3); }
        total += 512;
    }
}
// equal bytes going backwards from a[-1], b[-1], at most maxBack
DEV int wave_back_ptr(const uint8_t* a, const uint8_t* b, int maxBack)
{
    int total = 0;
    while (total < maxBack) {
        const uint64_t neq = BALLOT((total + LANE >= maxBack) || (a[-1 - total - LANE] != b[-1 - total - LANE]));
        if (neq) return total + ctz64(neq);
        total += 64;
    }
    return maxBack;
}

DEV int wave_encode_block_dict(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap,
                               const DictEnc dc, void* tab)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;
    if (n == 0) { if (cap <= 0) return 0; LANES({ if (LANE == 0) dst[0] = 0; }) return 1; }     // lz4.c:1361-1371 (always limitedOutput)

    uint32_t* T = (uint32_t*)tab;
    const int       mode       = dc.mode;
    const uint32_t  startIndex = (mode == kDictFreshPrefix) ? 0u : 65536u;
    const bool      extMem     = (mode == kDictLoad || mode == kDictCtxCopy || mode == kDictCtxLookup);
    const uint32_t  dictSize   = extMem ? (uint32_t)dc.dictSize : 0u;
    const uint8_t*  dictEnd    = extMem ? dc.dict + dc.dictSize : nullptr;
    const bool      dictSmall  = (mode == kDictNonePrefix) || (mode == kDictLoad && dictSize < 65536u);
    const uint32_t  prefixIdxLimit = startIndex - dictSize;                                      // lz4.c:959

    // ---- prime the table
    if (mode == kDictCtxCopy) {
        LANES({ for (int i = LANE; i < kHashBytes / 4; i += 64) T[i] = dc.dictTable[i]; })        // LZ4_memcpy(streamPtr, dictCtx), lz4.c:1767
    } else {
        LANES({ for (int i = LANE; i < kHashBytes / 4; i += 64) T[i] = 0; })
        if (mode == kDictLoad) {
            LDS_FENCE();
            // LZ4_loadDict (lz4.c:1621-1628): every 3rd position, later entries overwrite earlier ones == max index per slot
            const int cnt = (dc.dictSize - 8) / 3 + 1;
            const uint32_t idx0 = 65536u - dictSize;
            LANES({ for (int k = LANE; k < cnt; k += 64) lds_max(&T[seq_hash<false>(ld64u(dc.dict + 3 * k))], idx0 + 3u * (uint32_t)k); })
        }
    }
    LDS_FENCE();

    const int lastProbe  = n - kMfLimit + 1;
    const int matchLimit = n - kLastLiterals;
    int anchor = 0, op = 0;

    if (n >= kMinLength) {
        int  insPos = 0;  bool hasIns = true;
        int  rePos  = 0;  bool hasRe  = false;
        int  sBase  = 1;  int  sIter  = 0;
        int  width  = 16;
        for (;;) {
            const int pre = (hasIns ? 1 : 0) + (hasRe ? 1 : 0);
            LV(int, q); LV(uint32_t, h); LV(uint32_t, old); LV(uint32_t, rb); LV(uint32_t, lo4); LV(uint32_t, cidx);
            LV(int, ok); LV(int, hit); LV(int, inD);
            LANES({
                const int k = LANE - pre;
                int pos, stride, fine = 1;
                if (k < 0) { pos = (hasIns && LANE == 0) ? insPos : rePos; }
                else { probe_pos(sBase, sIter + k, &pos, &stride); fine = (pos + stride <= lastProbe); }
                q[I_] = pos; ok[I_] = fine && (LANE < width);
            })
            const uint64_t okMask = BALLOT(ok[I_]);
            const int nproc = (~okMask) ? ctz64(~okMask) : 64;
            const bool endInBatch = nproc < width;

            LANES({
                hit[I_] = 0; inD[I_] = 0; cidx[I_] = 0;
                if (LANE < nproc) {
                    const uint64_t s8 = ld64u(src + q[I_]);
                    lo4[I_] = (uint32_t)s8;
                    h[I_] = seq_hash<false>(s8);
                    old[I_] = T[h[I_]];
                }
            })
            LDS_FENCE();
            LANES({ if (LANE < nproc) T[h[I_]] = (uint32_t)q[I_] + startIndex; })
            LDS_FENCE();
            LANES({ rb[I_] = (LANE < nproc) ? T[h[I_]] : (uint32_t)q[I_] + startIndex; })
            // candidate by dictionary mode (lz4.c:1058-1099)
            LANES({
                if (LANE < nproc && !(hasIns && LANE == 0)) {
                    const uint32_t cur = (uint32_t)q[I_] + startIndex;
                    uint32_t mi = old[I_];
                    const uint8_t* mp;
                    if (extMem && mi < startIndex) {
                        if (mode == kDictCtxLookup) mi = dc.dictTable[h[I_]];                    // dictCtx->currentOffset == 64 KiB == startIndex
                        mp = dictEnd - (startIndex - mi);
                        inD[I_] = 1;
                    } else {
                        mp = src + (mi - startIndex);
                    }
                    cidx[I_] = mi;
                    if (!(dictSmall && mi < prefixIdxLimit) && mi + kMaxDist >= cur) hit[I_] = (ld32u(mp) == lo4[I_]);
                }
            })
            const uint64_t losers = BALLOT((LANE < nproc) && rb[I_] != (uint32_t)q[I_] + startIndex);
            const uint64_t hits   = BALLOT(hit[I_]);
            int safe = nproc;
            if (losers) safe = min_(nproc, max_(ctz64(losers), 1));
            const uint64_t hitsSafe = hits & ((safe >= 64) ? ~0ull : ((1ull << safe) - 1));
            const int keep = hitsSafe ? ctz64(hitsSafe) + 1 : safe;
            LANES({ if (LANE >= keep && LANE < nproc) T[h[I_]] = old[I_]; })
            if (losers) { LDS_FENCE(); LANES({ if (LANE < keep) T[h[I_]] = (uint32_t)q[I_] + startIndex; }) }
            LDS_FENCE();

            if (!hitsSafe) {
                int used = keep;
                if (hasIns && used > 0) { hasIns = false; used--; }
                if (hasRe  && used > 0) { hasRe = false; used--; }
                sIter += used;
                if (keep == nproc && endInBatch && !hasIns && !hasRe) break;
                width = 64;
                continue;
            }
            const int  w     = keep - 1;
            const bool isRe  = hasRe && (w == pre - 1);
            int        p     = RL(q, w);
            const uint32_t mi = RL(cidx, w);
            const bool inDict = RL(inD, w) != 0;
            const uint32_t offset = (uint32_t)p + startIndex - mi;                               // lz4.c:1097 (before catch-up; unchanged by it)
            const uint8_t* mp    = inDict ? dictEnd - (startIndex - mi) : src + (mi - startIndex);
            const uint8_t* floor = inDict ? dc.dict : src;                                       // lowLimit (lz4.c:1065-1079)

            if (!isRe) {
                const int back = wave_back_ptr(src + p, mp, min_(p - anchor, (int)(mp - floor)));
                p -= back; mp -= back;
            }
            const int lit = p - anchor;
            const int tokPos = op++;
            if (!isRe && (int64_t)op + lit + (2 + 1 + kLastLiterals) + lit / 255 > cap) return 0;
            if (lit >= 15) op = emit_len_ext(dst, op, lit - 15);
            wave_copy(dst + op, src + anchor, lit);
            op += lit;
            LANES({ if (LANE == 0) st16u(dst + op, (uint16_t)offset); })
            op += 2;
            int mc, ip;
            if (inDict) {                                                                        // lz4.c:1168-1180
                int lim = p + (int)(dictEnd - mp);
                if (lim > matchLimit) lim = matchLimit;
                mc = wave_count_ptr(src + p + kMinMatch, mp + kMinMatch, lim - (p + kMinMatch));
                ip = p + kMinMatch + mc;
                if (ip == lim) { const int more = wave_count_ptr(src + lim, src, matchLimit - lim); mc += more; ip += more; }
            } else {
                mc = wave_count_ptr(src + p + kMinMatch, mp + kMinMatch, matchLimit - (p + kMinMatch));
                ip = p + kMinMatch + mc;
            }
            if ((int64_t)op + (1 + kLastLiterals) + (mc + 240) / 255 > cap) return 0;
            LANES({ if (LANE == 0) dst[tokPos] = (uint8_t)((min_(lit, 15) << 4) | min_(mc, 15)); })
            if (mc >= 15) op = emit_len_ext(dst, op, mc - 15);
            anchor = ip;
            if (ip >= lastProbe) break;
            hasIns = true; insPos = ip - 2;
            hasRe = true;  rePos = ip;
            sBase = ip + 1; sIter = 0; width = 16;
        }
    }
    {
        const int last = n - anchor;
        if ((int64_t)op + last + 1 + (last + 255 - 15) / 255 > cap) return 0;
        if (last >= 15) { LANES({ if (LANE == 0) dst[op] = 0xF0; }) op = emit_len_ext(dst, op + 1, last - 15); }
        else { LANES({ if (LANE == 0) dst[op] = (uint8_t)(last << 4); }) op++; }
        wave_copy(dst + op, src + anchor, last);
        op += last;
    }
    return op;
}

// Every priming but the dictionary-context lookup (kDictCtxLookup, blocks <= 4 KiB: wave_encode_block_dict above) runs the
// full encoder -- grid batches included -- in its external-segment mode: for kDictLoad / kDictCtxCopy the `segLen` bytes of the
// previous block's tail / of the dictionary lie right before `blk` (the kernels copy them there).
DEV int wave_encode_block_ext(const uint8_t* blk, const int n, uint8_t* __restrict__ dst, const int cap, const int mode,
                              const int segLen, const uint32_t* dictTable, void* tab)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;
    if (n == 0) { if (cap <= 0) return 0; LANES({ if (LANE == 0) dst[0] = 0; }) return 1; }     // lz4.c:1361-1371 (always limitedOutput)
    ExtEnc x{0, 0u, 0, nullptr};
    if (mode == kDictNonePrefix) x.delta = 65536u;
    else if (mode == kDictLoad)    { x.pfx = segLen; x.delta = 65536u - (uint32_t)segLen; x.prime = 1; }
    else if (mode == kDictCtxCopy) { x.pfx = segLen; x.delta = 65536u - (uint32_t)segLen; x.prime = 2; x.table = dictTable; }
    return wave_encode_block_tt<false, true>(blk - x.pfx, x.pfx + n, dst, cap, true, tab, x);
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

// Vector fast path of the decoder: 64 lanes look at the next 64 input bytes, each as if a sequence started at its
// byte (it holds the 16 bytes from there on, so token, literals and offset of a short sequence are in registers);
// a scalar hop follows the real token chain through the window while the sequences are "plain" (literal run < 14,
// literals inside the window, match length in the nibble or in one extension byte); output positions come from a
// wave prefix sum on the DPP network; matches whose source lies before this batch's output are copied together, the
// others ("near": the source holds bytes of this very batch) follow in rounds of mutually independent copies, long and
// overlapping ones as one cooperative copy each.  Everything else -- long literal runs, very long matches, bad offsets,
// the ends of the block -- is left to the exact sequential step of
// wave_decode_block, which also owns liblz4's accept/reject rules.  Returns the number of sequences decoded.
// Caller guarantees ip0 + 160 <= iend and op0 + 1088 <= oend (the reference is in its fast loop there).
// kLds: the batch's output is assembled in `lb`, kDecLdsBytes of LDS owned by this wave (the last kDecTail bytes already
// written, then up to 1024 new ones), and stored to memory in one coalesced sweep.  Near matches then read LDS instead of
// waiting a memory round trip per dependency round; `tailAt` is the output position the LDS tail is valid for.
enum : int { kDecTail = 32, kDecDump = kDecTail + 1024 + 64, kDecLdsBytes = kDecDump + 16 };    // (kDecDump: 16 bytes nobody reads)
template <bool kLds>
DEV int wave_decode_plain_batch(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int* ipp, int64_t* opp,
                                LVREF(v16u_t, win), int* winIp, uint8_t* lb = nullptr, int64_t* tailAt = nullptr,
                                unsigned long long* st_ = nullptr)           // (diagnostics build: the caller's counters)
{
    const int ip0 = *ipp; const int64_t op0 = *opp;
    (void)st_;
    const unsigned long long td0 = STAT_NOW(); (void)td0;
    // every lane holds the 16 input bytes from its window position on; normally requested by the previous batch
    if (*winIp != ip0) { LANES({ win[I_] = *(const v16u_t*)(src + ip0 + LANE); }) }
    LV(uint32_t, b0); LV(int, ll); LV(int, ml); LV(int, off); LV(int, nxt); LV(int, outLen); LV(int, plain); LV(int, coop);
    LV(int, longLit);
    LANES({
        const uint64_t w0 = (uint64_t)win[I_].w[0] | ((uint64_t)win[I_].w[1] << 32);
        const uint64_t w1 = (uint64_t)win[I_].w[2] | ((uint64_t)win[I_].w[3] << 32);
        const uint32_t t = (uint32_t)(w0 & 0xFF);
        const int l = (int)(t >> 4), mn = (int)(t & 15);
        const int offAt = LANE + 1 + l;                      // window index of the offset's low byte
        // the offset sits at bytes 1+l, 2+l of the lane's 16: pick the 8-byte word that holds both (l <= 13)
        const uint64_t sel = (l <= 5) ? w0 : (l <= 9 ? ((w0 >> 32) | (w1 << 32)) : w1);
        const int      sft = 8 * (1 + l - (l <= 5 ? 0 : (l <= 9 ? 4 : 8)));
        const uint32_t o16 = (uint32_t)((sel >> sft) & 0xFFFF);
        // a long match (nibble 15) with a single extension byte, byte 3+l of the 16 (l <= 12)
        const bool     lng = (mn == 15);
        const int      ek  = 3 + l;
        const uint32_t e   = (uint32_t)(((ek < 8) ? (w0 >> (8 * ek)) : (w1 >> (8 * (ek & 7)))) & 0xFF);
        const int      mlen = lng ? 19 + (int)e : mn + kMinMatch;
        const bool parse = (l < 14) && (offAt <= 64) && (!lng || (l <= 12 && e < 255));
        const bool simple = parse && !lng && o16 >= (uint32_t)mlen;     // per-lane copy of <= 18 bytes, no overlap
        b0[I_] = t; ll[I_] = l; ml[I_] = mlen; off[I_] = (int)o16;
        nxt[I_] = offAt + 2 + (lng ? 1 : 0); outLen[I_] = l + mlen;
        coop[I_]  = parse && !simple && o16 >= 1;                       // long or overlapping: one cooperative copy
        plain[I_] = simple || coop[I_];
        // a long literal run (nibble 15, one extension byte) whose literals still end inside the window: its offset and
        // match length sit in another lane's registers, fetched below
        const uint32_t e1 = (uint32_t)((w0 >> 8) & 0xFF);
        longLit[I_] = (l == 15) && (e1 < 255) && (LANE + 2 + 15 + (int)e1 <= 64);
    })
    if (BALLOT(longLit[I_])) {
        // rare (token bytes >= 0xF0): one lane exchange brings the 4 bytes at the offset's position to the sequence's lane
        LV(uint32_t, w0lo); LV(int, srcLane); LV(uint32_t, got);
        LANES({
            w0lo[I_] = win[I_].w[0];
            const int l = 15 + (int)((win[I_].w[0] >> 8) & 0xFF);
            const int offAt = LANE + 2 + l;                              // token, extension byte, l literals
            srcLane[I_] = longLit[I_] ? min_(offAt, 63) : LANE;
        })
        LANES({ got[I_] = SHFL(w0lo, srcLane[I_]); })
        LANES({
            if (longLit[I_]) {
                const int l = 15 + (int)((win[I_].w[0] >> 8) & 0xFF);
                const int offAt = LANE + 2 + l;
                const uint32_t v = got[I_] >> (8 * (offAt - srcLane[I_]));          // bytes offAt.. of the window (shift 0 or 8)
                const uint32_t o16 = v & 0xFFFF;
                const int mn = (int)(b0[I_] & 15);
                const bool lng = (mn == 15);
                const uint32_t e = (v >> 16) & 0xFF;
                const bool have = !lng || e < 255;                                  // the extension byte is byte 2 of those
                const int mlen = lng ? 19 + (int)e : mn + kMinMatch;
                const bool simple = have && !lng && o16 >= (uint32_t)mlen;
                ll[I_] = l | (1 << 16);                                             // bit 16: the literals start one byte later
                ml[I_] = mlen; off[I_] = (int)o16;
                nxt[I_] = offAt + 2 + (lng ? 1 : 0); outLen[I_] = l + mlen;
                coop[I_]  = have && !simple && o16 >= 1;
                plain[I_] = simple || coop[I_];
            }
        })
    }
    const uint64_t plainMask = BALLOT(plain[I_]);
    const unsigned long long td1 = STAT_NOW(); (void)td1;
    STAT(0, td1 - td0);
    if (!(plainMask & 1)) return 0;
    // Follow the token chain through the whole window, four hops to a branch (mark the lane, fetch its successor; a
    // finished walk only re-marks lane 0, where it started), then cut the chain at the first sequence that is not plain.
    uint64_t members = 0;
    {
        // (successors clamped to 64 per lane up front: the hop's & 63 then reads a finished walk back as lane 0, where the chain
        // started -- walking on from there only re-marks its own members, so the hop needs no test; `seen` remembers the end)
        LV(int, nxtC);
        LANES({ nxtC[I_] = min_(nxt[I_], 64); })
        int cur = 0, seen = 0;
        do {
            for (int u = 0; u < 4; ++u) {
                members |= 1ull << (cur & 63);
                cur = RL(nxtC, cur & 63);
                seen |= cur;
            }
        } while (seen < 64);
        const uint64_t odd = members & ~plainMask;
        if (odd) members &= (1ull << ctz64(odd)) - 1;
    }

    // exclusive prefix sum of the members' output lengths -> where each sequence writes
    LV(int, acc); LV(int, outStart); LV(int, sp);
    { const uint64_t mL = members; LANES({ acc[I_] = LANE_IN(mL) ? outLen[I_] : 0; }) }
    SCAN_INCL(acc);
    {
        const uint64_t mL = members;
        LANES({
            outStart[I_] = (int)op0 + acc[I_] - (LANE_IN(mL) ? outLen[I_] : 0);
            sp[I_]       = outStart[I_] + (ll[I_] & 0xFFFF) - off[I_];                     // where the match bytes come from
        })
        // a source before the start of the output is the sequential step's business (error, or a dictionary);
        // and the batch's output stays inside the room the caller checked
        // (LDS staging: a long match whose source starts before the staged tail and runs into this batch's output as well)
        const uint64_t stop = BALLOT(LANE_IN(mL) && (sp[I_] < 0 || acc[I_] > 1024 ||
                                     (kLds && coop[I_] && sp[I_] < op0 - kDecTail && (int64_t)sp[I_] + ml[I_] > op0)));
        if (stop) members &= (1ull << ctz64(stop)) - 1;
    }
    const unsigned long long td2 = STAT_NOW(); (void)td2;
    STAT(1, td2 - td1);
    if (!members) return 0;
    const uint64_t mL = members;
    const int last = 63 - __builtin_clzll(members);
    const int ipn  = ip0 + RL(nxt, last);
    // request the next batch's window now: its latency hides behind this batch's copies (ipn + 63 + 16 < ip0 + 160 <= iend)
    LANES({ win[I_] = *(const v16u_t*)(src + ipn + LANE); })
    *winIp = ipn;
    // Match bytes, 4..18 per member lane.  "Far" sources end before this batch's output: all of them are copied at once.
    // A "near" source may contain bytes this batch produces; those go in dependency order below.
    const uint64_t coopM = mL & BALLOT(coop[I_]);
    const uint64_t far = mL & ~coopM & BALLOT((int64_t)sp[I_] + ml[I_] <= op0);
    if (kLds) {
        const int total = RL(acc, last);                       // bytes this batch produces
        // A match of 4..18 bytes travels as four overlapping pieces: [0,4) and [len-4,len) always, [len-8,len-4) from 8 bytes on,
        // [4,12) from 12 on -- together they cover every length, and where two overlap they carry the same bytes.  Every piece
        // has a fixed place per lane and batch, so a round is four loads and four stores; a lane that is not part of the round
        // (or a piece its length does not have) stores to the dump bytes behind the buffer: a divergent `if` costs this machine
        // ~45 cycles whether any lane takes it or not, a redirected store costs a select.
        LV(int, mdst); LV(int, qoff);
        LANES({
            mdst[I_] = kDecTail + (outStart[I_] - (int)op0) + (ll[I_] & 0xFFFF);       // where the match goes, as an offset into lb
            qoff[I_] = mdst[I_] - off[I_];                                             // ... and where it comes from
        })
        LV(uint32_t, g0); LV(uint32_t, g1); LV(uint32_t, g2); LV(uint64_t, g3);
        auto put_pieces = [&](const uint64_t who) {
            LANES({
                const bool on = LANE_IN(who);
                const int len = on ? ml[I_] : 0, at = on ? mdst[I_] : (int)kDecDump;
                st32u(lb + at, g0[I_]);
                st32u(lb + at + max_(len - 4, 0), g1[I_]);
                st32u(lb + (len >= 8 ? at + len - 8 : (int)kDecDump), g2[I_]);
                st64u(lb + (len >= 12 ? at + 4 : (int)kDecDump + 8), g3[I_]);
            })
        };
        // far matches: their bytes are requested from memory first ...
        LANES({
            g0[I_] = 0; g1[I_] = 0; g2[I_] = 0; g3[I_] = 0;
            if (LANE_IN(far)) {
                const uint8_t* q = dst + sp[I_]; const int len = ml[I_];
                g0[I_] = ld32u(q); g1[I_] = ld32u(q + len - 4); g2[I_] = ld32u(q + max_(len - 8, 0)); g3[I_] = ld64u(q + 4);
            }
        })
        // ... the tail of what is already written, if LDS does not hold it (after a sequential step)
        if (*tailAt != op0) {
            LANES({ if (LANE < kDecTail) { const int64_t g = op0 - kDecTail + LANE; lb[LANE] = g >= 0 ? dst[g] : (uint8_t)0; } })
        }
        // literal bytes: every window byte finds the member it follows
        LANES({
            const uint64_t upto = mL & ((LANE >= 63) ? ~0ull : ((2ull << LANE) - 1));
            const int m  = upto ? 63 - __builtin_clzll(upto) : LANE;
            const int os = SHFL(outStart, m), lp = SHFL(ll, m);
            const int lm = lp & 0xFFFF, m1 = m + (lp >> 16);              // m1: the byte before the first literal
            if (upto && LANE > m1 && LANE <= m1 + lm) lb[kDecTail + (os - (int)op0) + (LANE - m1 - 1)] = (uint8_t)b0[I_];
        })
        const unsigned long long td3 = STAT_NOW(); (void)td3;
        STAT(2, td3 - td2);
        put_pieces(far);
        const unsigned long long td4 = STAT_NOW(); (void)td4;
        STAT(3, td4 - td3);
        // the rest in dependency order, LDS to LDS (see the memory version below for the rule)
        for (uint64_t pend = mL & ~far; pend; ) {
            const int f  = ctz64(pend);
            const int lo = RL(mdst, f);                                                 // (an offset into lb)
            LDS_FENCE();
            if ((coopM >> f) & 1) {
                const int len = RL(ml, f); const int64_t s0 = op0 + (RL(qoff, f) - kDecTail);
                if (s0 + len <= op0) { LANES({ for (int i = LANE; i < len; i += 64) lb[lo + i] = dst[s0 + i]; }) }   // older than the window: from memory
                else wave_copy_match(lb, lo, RL(off, f), len);
                pend &= pend - 1;
            } else {
                STAT(8, 1);
                const uint64_t go = pend & ~coopM & BALLOT(qoff[I_] + ml[I_] <= lo);
                LANES({
                    if (LANE_IN(go)) {
                        const uint8_t* q = lb + qoff[I_]; const int len = ml[I_];
                        g0[I_] = ld32u(q); g1[I_] = ld32u(q + len - 4); g2[I_] = ld32u(q + max_(len - 8, 0)); g3[I_] = ld64u(q + 4);
                    }
                })
                LDS_FENCE();
                put_pieces(go);
                pend &= ~go;
            }
        }
        LDS_FENCE();
        const unsigned long long td5 = STAT_NOW(); (void)td5;
        STAT(4, td5 - td4); STAT(6, 1);
        // one sweep to memory, 16 bytes per lane (total <= 1024); the last chunk is written whole -- the bytes behind `total` are
        // the next batch's (or the sequential step's) to write, and the caller has checked that 1088 bytes of room are left --
        // and the last kDecTail bytes are kept for the next batch: all LDS reads first, one wait, then the stores
        LV(v16u_t, sw); LV(uint32_t, t8);
        LANES({
            const uint64_t q0 = ld64u(lb + kDecTail + LANE * 16), q1 = ld64u(lb + kDecTail + LANE * 16 + 8);
            sw[I_].w[0] = (uint32_t)q0; sw[I_].w[1] = (uint32_t)(q0 >> 32); sw[I_].w[2] = (uint32_t)q1; sw[I_].w[3] = (uint32_t)(q1 >> 32);
            t8[I_] = (uint32_t)lb[total + (LANE & (kDecTail - 1))];
        })
        LDS_FENCE();
        LANES({
            if (LANE * 16 < total) *(v16u_t*)(dst + op0 + LANE * 16) = sw[I_];
            if (LANE < kDecTail) lb[LANE] = (uint8_t)t8[I_];
        })
        *tailAt = op0 + total;
        *ipp = ipn;
        *opp = op0 + total;
        STAT(5, STAT_NOW() - td5); STAT(7, __builtin_popcountll(far));
        return __builtin_popcountll(members);
    }
    auto copy_matches = [&](const uint64_t who) {
        LANES({
            if (LANE_IN(who)) {
                const uint8_t* s = dst + sp[I_];
                uint8_t*       d = dst + outStart[I_] + (ll[I_] & 0xFFFF);
                const v16u_t a = *(const v16u_t*)s;
                const uint32_t b = ld16u(s + 16);
                const uint64_t lo = (uint64_t)a.w[0] | ((uint64_t)a.w[1] << 32), hi = (uint64_t)a.w[2] | ((uint64_t)a.w[3] << 32);
                int rem = ml[I_]; uint64_t cur;
                if (rem >= 16)     { *(v16u_t*)d = a; d += 16; rem -= 16; cur = b; }
                else if (rem >= 8) { st64u(d, lo); d += 8; rem -= 8; cur = hi; }
                else cur = lo;
                if (rem & 4) { st32u(d, (uint32_t)cur); d += 4; cur >>= 32; }
                if (rem & 2) { st16u(d, (uint16_t)cur); d += 2; cur >>= 16; }
                if (rem & 1) { *d = (uint8_t)cur; }
            }
        })
    };
    copy_matches(far);
    // literal bytes: every window byte finds the member it follows
    LANES({
        const uint64_t upto = mL & ((LANE >= 63) ? ~0ull : ((2ull << LANE) - 1));
        const int m  = upto ? 63 - __builtin_clzll(upto) : LANE;
        const int os = SHFL(outStart, m), lp = SHFL(ll, m);
        const int lm = lp & 0xFFFF, m1 = m + (lp >> 16);                  // m1: the byte before the first literal
        if (upto && LANE > m1 && LANE <= m1 + lm) dst[os + (LANE - m1 - 1)] = (uint8_t)b0[I_];
    })
    // the rest in dependency order.  Near matches: a round takes every pending one whose source ends before the first
    // pending match's output (the first one always qualifies: offset >= length), so nothing it reads is still to be
    // written.  Long or overlapping matches: one cooperative copy each, when they come first.
    for (uint64_t pend = mL & ~far; pend; ) {
        const int f  = ctz64(pend);
        const int lo = RL(outStart, f) + (RL(ll, f) & 0xFFFF);
        WAVE_FENCE();
        if ((coopM >> f) & 1) {
            wave_copy_match(dst, lo, RL(off, f), RL(ml, f));
            pend &= pend - 1;
        } else {
            const uint64_t go = pend & ~coopM & BALLOT(sp[I_] + ml[I_] <= lo);
            copy_matches(go);
            pend &= ~go;
        }
    }
    *ipp = ipn;
    *opp = (int64_t)RL(outStart, last) + RL(outLen, last);
    return __builtin_popcountll(members);
}

// LZ4_decompress_safe, full block, no dictionary.  Returns decoded size or liblz4's negative error code
// -(input position)-1.  The reference's fast loop (>= 64 output bytes left) and safe loop reject at
// different points, so both sets of tests are reproduced (see oracle/plz4_oracle.c for the same shape).
// With a dictionary (LZ4_decompress_safe_usingDict, lz4.c:2719-2732 -> forceExtDict): matches may start in `dict`
// (<= 64 KiB, not adjacent to dst) and run on into the block (lz4.c:2166-2196, :2358-2384).
template <bool kLds = false>
DEV int wave_decode_block(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap,
                          const uint8_t* dict = nullptr, const int dictLen = 0, uint8_t* lb = nullptr)
{
    if (src == nullptr || cap < 0) return -1;
    const bool ext = (dict != nullptr && dictLen > 0);
    const int64_t dictSize = ext ? dictLen : 0;
    const uint8_t* const dictEnd = ext ? dict + dictLen : nullptr;
    const bool checkOffset = dictSize < 65536;                                   // lz4.c:2047
    const int iend = n;
    const int64_t oend = cap;
    int ip = 0; int64_t op = 0;
    if (cap == 0) return (n == 1 && UNI(src[0]) == 0) ? 0 : -1;
    if (n == 0) return -1;
    bool fast = (oend - op) >= 64;
    int64_t tailAt = -1;                          // LDS staging: output position the staged tail belongs to
    LV(v16u_t, win); int winIp = -1;              // the vector path's input window, requested one batch ahead
    LANES({ win[I_].w[0] = 0; win[I_].w[1] = 0; win[I_].w[2] = 0; win[I_].w[3] = 0; })
    STAT_DECL;

    for (;;) {
        if (fast && ip + 160 <= iend && op + 1088 <= oend) {
            WAVE_FENCE();
            const int nm = wave_decode_plain_batch<kLds>(src, dst, &ip, &op, win, &winIp, lb, &tailAt, STAT_PTR);
            STAT(S_DBATCH, 1); STAT(S_DMEMB, nm);
            if (nm > 0) { WAVE_FENCE(); continue; }
        }
        const uint32_t token = UNI(src[ip]); ip++;
        STAT(S_DSEQ, 1); STAT(S_DSEQ_ML15, (token & 15) == 15); STAT(S_DSEQ_LL15, (token >> 4) == 15);
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
            if (checkOffset && mpos + dictSize < 0) return -ip - 1;         // lz4.c:2161
            if (ext && mpos < 0) {
                if (op + ml > oend - kLastLiterals) return -ip - 1;         // lz4.c:2168-2175
                goto dict_copy;
            }
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
        if (checkOffset && mpos + dictSize < 0) return -ip - 1;             // lz4.c:2356
        if (ext && mpos < 0) {
            if (op + ml > oend - kLastLiterals) return -ip - 1;             // lz4.c:2360-2363
            goto dict_copy;
        }
        if (op + ml > oend - kLastLiterals) return -ip - 1;                 // lz4.c:2421-2423
        WAVE_FENCE();
        wave_copy_match(dst, op, offset, (int)ml);
        WAVE_FENCE();
        op += ml;
        continue;
dict_copy:                                                                   // lz4.c:2177-2195 == :2365-2383
        {
            const int back = (int)(-mpos);                                   // bytes of the match that lie in the dictionary
            WAVE_FENCE();
            if (ml <= back) {
                wave_copy(dst + op, dictEnd - back, (int)ml);
                op += ml;
            } else {
                wave_copy(dst + op, dictEnd - back, back);
                op += back;
                WAVE_FENCE();
                wave_copy_match(dst, op, (int)op, (int)(ml - back));         // the rest comes from the start of the block
                op += ml - back;
            }
            WAVE_FENCE();
        }
    }
    STAT_FLUSH();
    return (int)op;
}

}  // namespace plz4
