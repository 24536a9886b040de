/*
 * plz4_oracle.c -- CPU restatement of plz4's per-block hot path.  TEST INFRASTRUCTURE ONLY
 * (see plz4_oracle.h for the rules on who may use it and for the parity-pinning statement).
 *
 * The arithmetic lives in liblz4 v1.10.0, vendored unmodified by the reference at
 * internal/pkg/clz4/{lz4.c,lz4.h}; the glue is Go (internal/pkg/{blk,header,trailer,descriptor,xxh32}).
 * Each function below names the reference lines it restates.  Nothing here is copied: the
 * encoder/decoder are re-expressed over integer positions with explicit mode flags instead of the
 * reference's pointer/goto/template style, but every accept/reject inequality and every table
 * read/write happens in the same order, because that order defines the output bytes.
 */
#include "plz4_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------ little helpers */
static inline uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void     st16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static inline void     st32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* ================================================================== xxHash32 (seed 0)
 * internal/pkg/xxh32/xxh32zero.go:10-19 (primes), :238-280 (one-shot), :58-86/:204-235 (streaming). */
#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u

static inline uint32_t xround(uint32_t acc, uint32_t lane) { return rotl(acc + lane * XP2, 13) * XP1; }

static uint32_t xfinish(uint32_t h, const uint8_t* p, size_t rem)
{
    while (rem >= 4) { h = rotl(h + ld32(p) * XP3, 17) * XP4; p += 4; rem -= 4; }   /* :263-266 */
    while (rem)      { h = rotl(h + (uint32_t)*p * XP5, 11) * XP1; p++; rem--; }     /* :267-271 */
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;                    /* :273-277 */
    return h;
}

uint32_t orc_xxh32(const uint8_t* p, size_t n)
{
    uint32_t h = (uint32_t)n;
    size_t rem = n;
    if (n >= 16) {
        uint32_t a = XP1 + XP2, b = XP2, c = 0, d = 0u - XP1;                        /* :245-248 */
        while (rem >= 16) {
            a = xround(a, ld32(p)); b = xround(b, ld32(p + 4));
            c = xround(c, ld32(p + 8)); d = xround(d, ld32(p + 12));
            p += 16; rem -= 16;
        }
        h += rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18);                    /* :259 */
    } else {
        h += XP5;                                                                    /* :242-243 */
    }
    return xfinish(h, p, rem);
}

void orc_xxh32_reset(orc_xxh32_state* s)
{
    s->acc[0] = XP1 + XP2; s->acc[1] = XP2; s->acc[2] = 0; s->acc[3] = 0u - XP1;
    s->total = 0; s->fill = 0;
}

void orc_xxh32_update(orc_xxh32_state* s, const uint8_t* p, size_t n)
{
    if (s->total == 0) orc_xxh32_reset(s);                                           /* :59-61 */
    s->total += n;
    if (s->fill) {
        size_t need = 16 - s->fill;
        if (n < need) { memcpy(s->buf + s->fill, p, n); s->fill += (uint32_t)n; return; }
        memcpy(s->buf + s->fill, p, need); p += need; n -= need;
        for (int i = 0; i < 4; i++) s->acc[i] = xround(s->acc[i], ld32(s->buf + 4 * i));
        s->fill = 0;
    }
    while (n >= 16) {
        for (int i = 0; i < 4; i++) s->acc[i] = xround(s->acc[i], ld32(p + 4 * i));
        p += 16; n -= 16;
    }
    if (n) { memcpy(s->buf, p, n); s->fill = (uint32_t)n; }
}

uint32_t orc_xxh32_digest(const orc_xxh32_state* s)
{
    uint32_t h = (uint32_t)s->total;
    if (h >= 16 || s->total >= 0x100000000ull)                                       /* :210 */
        h += rotl(s->acc[0], 1) + rotl(s->acc[1], 7) + rotl(s->acc[2], 12) + rotl(s->acc[3], 18);
    else
        h += XP5;
    return xfinish(h, s->buf, s->fill);
}

/* ================================================================== LZ4 block encoder, level 1
 * lz4.c:242-263 constants; :680-703 LZ4_count; :777-806 hashing; :930-1338 the parser. */
enum { TT_CLEARED = 0, TT_U32 = 2, TT_U16 = 3 };                    /* lz4.c:717 */
enum { DM_NONE = 0, DM_PREFIX, DM_EXT, DM_CTX };                    /* lz4.c:742 */

#define K_MINMATCH      4
#define K_MFLIMIT       12
#define K_LASTLITERALS  5
#define K_MINLENGTH     13          /* lz4.c:249 */
#define K_MAXDIST       65535u      /* lz4.h:674 */
#define K_64KLIMIT      (65536 + K_MFLIMIT - 1)   /* lz4.c:710 */
#define K_MAXINPUT      0x7E000000  /* lz4.h:214 */

int orc_compress_bound(int n)
{
    return ((unsigned)n > (unsigned)K_MAXINPUT) ? 0 : n + n / 255 + 16;
}

/* lz4.c:777-806 -- 64-bit little-endian build: byU16 hashes 4 bytes to 13 bits, everything else
 * hashes the low 5 bytes of an 8-byte read to 12 bits. */
static inline uint32_t pos_hash(const uint8_t* p, int tt)
{
    if (tt == TT_U16) return (ld32(p) * 2654435761u) >> 19;
    return (uint32_t)(((ld64(p) << 24) * 889523592379ull) >> 52);
}
static inline uint32_t tab_get(const uint32_t* t, uint32_t h, int tt)
{
    return tt == TT_U16 ? ((const uint16_t*)t)[h] : t[h];
}
static inline void tab_put(uint32_t* t, uint32_t h, uint32_t idx, int tt)
{
    if (tt == TT_U16) ((uint16_t*)t)[h] = (uint16_t)idx; else t[h] = idx;
}

/* lz4.c:680-703.  Equals min(common prefix, limit - a): the 8/4/2/1 ladder never reads past limit. */
static unsigned common_len(const uint8_t* a, const uint8_t* b, const uint8_t* limit)
{
    const uint8_t* s = a;
    while (a + 8 <= limit) {
        uint64_t d = ld64(a) ^ ld64(b);
        if (d) return (unsigned)(a - s) + (unsigned)(__builtin_ctzll(d) >> 3);
        a += 8; b += 8;
    }
    while (a < limit && *a == *b) { a++; b++; }
    return (unsigned)(a - s);
}

/* One candidate as the parser sees it: where its bytes are, how far back it may be extended,
 * and its index in the (virtual) index space. */
typedef struct { const uint8_t* p; const uint8_t* floor; uint32_t idx; int inDict; } cand_t;

typedef struct {
    orc_stream* s; const orc_stream* dctx;
    const uint8_t* src; const uint8_t* dict; const uint8_t* dictEnd;
    uint32_t startIndex, dictSize, dictDelta;
    int tt, dm;
} enc_t;

/* lz4.c:1058-1083 and :1259-1282: turn a table index into a candidate, by dictionary mode. */
static inline cand_t locate(const enc_t* e, uint32_t h, uint32_t idx)
{
    cand_t c; c.idx = idx; c.inDict = 0;
    if (e->dm == DM_CTX && idx < e->startIndex) {
        uint32_t di = e->dctx->table[h];
        c.p = e->dictEnd - (e->dctx->currentOffset - di);
        c.idx = di + e->dictDelta; c.floor = e->dict; c.inDict = 1;
    } else if (e->dm == DM_EXT && idx < e->startIndex) {
        c.p = e->dictEnd - (e->startIndex - idx);
        c.floor = e->dict; c.inDict = 1;
    } else {
        c.p = e->src + ((int64_t)idx - (int64_t)e->startIndex);
        c.floor = (e->dm == DM_PREFIX) ? e->src - e->dictSize : e->src;
    }
    return c;
}

static int encode_core(orc_stream* s, const uint8_t* src, int n, uint8_t* dst, int cap,
                       int limited, int tt, int dm, int dictSmall)
{
    enc_t e;
    e.s = s; e.dctx = s->dictCtx; e.src = src; e.tt = tt; e.dm = dm;
    e.startIndex = s->currentOffset;
    e.dict     = (dm == DM_CTX) ? e.dctx->dictionary : s->dictionary;                /* :951-954 */
    e.dictSize = (dm == DM_CTX) ? e.dctx->dictSize   : s->dictSize;
    e.dictDelta = (dm == DM_CTX) ? e.startIndex - e.dctx->currentOffset : 0;         /* :955-956 */
    e.dictEnd  = e.dict ? e.dict + e.dictSize : e.dict;
    const int      extMem = (dm == DM_EXT) || (dm == DM_CTX);
    const uint32_t prefixIdxLimit = e.startIndex - e.dictSize;                       /* :959 */
    const int      lastProbe  = n - K_MFLIMIT + 1;   /* mflimitPlusOne as a position   :963 */
    const int      matchLimit = n - K_LASTLITERALS;  /*                                 :964 */
    uint32_t* const T = s->table;

    /* context bookkeeping happens before any parsing (:990-1000) */
    if (dm == DM_CTX) { s->dictCtx = NULL; s->dictSize = (uint32_t)n; }
    else              { s->dictSize += (uint32_t)n; }
    s->currentOffset += (uint32_t)n;
    s->tableType = (uint32_t)tt;

    int ip = 0, anchor = 0, op = 0;

    if (n >= K_MINLENGTH) {                                                          /* :1002 */
        tab_put(T, pos_hash(src, tt), e.startIndex, tt);                             /* :1005-1010 */
        ip = 1;

        for (;;) {
            cand_t   c;
            uint32_t offset = 0;
            int      tok;

            /* ---- search: one probe per position, stride grows by one every 64 misses (:1042-1101) */
            {
                int      fwd = ip, stride = 1, misses = 1 << 6;
                uint32_t hNext = pos_hash(src + fwd, tt);
                for (;;) {
                    const uint32_t h   = hNext;
                    const uint32_t cur = (uint32_t)fwd + e.startIndex;
                    const uint32_t idx = tab_get(T, h, tt);
                    ip = fwd; fwd += stride; stride = (misses++ >> 6);
                    if (fwd > lastProbe) goto tail;                                  /* :1055 */
                    c = locate(&e, h, idx);
                    hNext = pos_hash(src + fwd, tt);
                    tab_put(T, h, cur, tt);
                    if (dictSmall && c.idx < prefixIdxLimit) continue;               /* :1088 */
                    if (tt != TT_U16 && c.idx + K_MAXDIST < cur) continue;           /* :1090-1093 */
                    if (ld32(c.p) == ld32(src + ip)) { if (extMem) offset = cur - c.idx; break; }
                }
            }

            /* ---- extend backwards over the pending literals (:1105-1109) */
            while (ip > anchor && c.p > c.floor && src[ip - 1] == c.p[-1]) { ip--; c.p--; }

            /* ---- literals (:1112-1136) */
            {
                const unsigned lit = (unsigned)(ip - anchor);
                tok = op++;
                if (limited && (int64_t)op + lit + (2 + 1 + K_LASTLITERALS) + lit / 255 > cap) return 0;
                if (lit >= 15) {
                    unsigned r = lit - 15;
                    dst[tok] = 0xF0;
                    for (; r >= 255; r -= 255) dst[op++] = 255;
                    dst[op++] = (uint8_t)r;
                } else {
                    dst[tok] = (uint8_t)(lit << 4);
                }
                memcpy(dst + op, src + anchor, lit);
                op += (int)lit;
            }

            /* ---- one or more back-to-back matches (:1138-1294) */
            for (;;) {
                unsigned mc;
                st16(dst + op, (uint16_t)(extMem ? offset : (uint32_t)((src + ip) - c.p)));  /* :1155-1163 */
                op += 2;

                if (extMem && c.inDict) {                                            /* :1168-1180 */
                    int64_t lim = ip + (e.dictEnd - c.p);
                    if (lim > matchLimit) lim = matchLimit;
                    mc = common_len(src + ip + K_MINMATCH, c.p + K_MINMATCH, src + lim);
                    ip += (int)mc + K_MINMATCH;
                    if (ip == lim) {
                        unsigned more = common_len(src + lim, src, src + matchLimit);
                        mc += more; ip += (int)more;
                    }
                } else {                                                             /* :1182-1184 */
                    mc = common_len(src + ip + K_MINMATCH, c.p + K_MINMATCH, src + matchLimit);
                    ip += (int)mc + K_MINMATCH;
                }

                if (limited && (int64_t)op + (1 + K_LASTLITERALS) + (mc + 240) / 255 > cap) return 0;  /* :1187-1210 */

                if (mc >= 15) {                                                      /* :1213-1225 */
                    dst[tok] += 15; mc -= 15;
                    for (; mc >= 255; mc -= 255) dst[op++] = 255;
                    dst[op++] = (uint8_t)mc;
                } else {
                    dst[tok] += (uint8_t)mc;
                }

                anchor = ip;
                if (ip >= lastProbe) goto tail;                                      /* :1233 */

                tab_put(T, pos_hash(src + ip - 2, tt), (uint32_t)(ip - 2) + e.startIndex, tt);  /* :1236-1242 */

                {   /* immediate re-test at ip (:1255-1294) */
                    const uint32_t h   = pos_hash(src + ip, tt);
                    const uint32_t cur = (uint32_t)ip + e.startIndex;
                    const uint32_t idx = tab_get(T, h, tt);
                    c = locate(&e, h, idx);
                    tab_put(T, h, cur, tt);
                    if ((dictSmall ? (c.idx >= prefixIdxLimit) : 1)
                        && ((tt == TT_U16) ? 1 : (c.idx + K_MAXDIST >= cur))
                        && ld32(c.p) == ld32(src + ip)) {
                        tok = op++; dst[tok] = 0;
                        if (extMem) offset = cur - c.idx;
                        continue;
                    }
                }
                break;
            }
            ip++;                                                                    /* :1298 */
        }
    }

tail:                                                                                /* :1302-1329 */
    {
        const size_t last = (size_t)(n - anchor);
        if (limited && (int64_t)op + (int64_t)last + 1 + (int64_t)((last + 255 - 15) / 255) > cap) return 0;
        if (last >= 15) {
            size_t r = last - 15;
            dst[op++] = 0xF0;
            for (; r >= 255; r -= 255) dst[op++] = 255;
            dst[op++] = (uint8_t)r;
        } else {
            dst[op++] = (uint8_t)(last << 4);
        }
        memcpy(dst + op, src + anchor, last);
        op += (int)last;
    }
    return op;
}

/* lz4.c:1344-1379: size screening + the empty-input special case. */
static int encode_generic(orc_stream* s, const uint8_t* src, int n, uint8_t* dst, int cap,
                          int limited, int tt, int dm, int dictSmall)
{
    if ((uint32_t)n > (uint32_t)K_MAXINPUT) return 0;
    if (n == 0) {
        if (limited && cap <= 0) return 0;
        dst[0] = 0;
        return 1;
    }
    return encode_core(s, src, n, dst, cap, limited, tt, dm, dictSmall);
}

void orc_stream_init(orc_stream* s) { memset(s, 0, sizeof(*s)); }

/* lz4.c:1382-1403 + :1453-1469: fresh zeroed state per call, accel fixed at 1 (clz4.go:31). */
int orc_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    static _Thread_local orc_stream st;
    orc_stream_init(&st);
    const int limited = !(cap >= orc_compress_bound(n));
    const int tt = (n < K_64KLIMIT) ? TT_U16 : TT_U32;
    return encode_generic(&st, src, n, dst, limited ? cap : 0, limited, tt, DM_NONE, 0);
}

/* lz4.c:883-922 with (inputSize=0, tableType=byU32), as called by LZ4_resetStream_fast :1570. */
void orc_stream_reset_fast(orc_stream* s)
{
    if (s->tableType != TT_CLEARED) {
        if (s->tableType != TT_U32 || s->currentOffset > (1u << 30)) {
            memset(s->table, 0, sizeof(s->table));
            s->currentOffset = 0;
            s->tableType = TT_CLEARED;
        }
    }
    if (s->currentOffset != 0) s->currentOffset += 65536;
    s->dictCtx = NULL; s->dictionary = NULL; s->dictSize = 0;
}

/* lz4.c:1587-1646 */
int orc_stream_load_dict(orc_stream* s, const uint8_t* d, int n, int slow)
{
    const uint8_t* p = d;
    const uint8_t* const end = d + n;
    orc_stream_init(s);
    s->currentOffset += 65536;
    if (n < 8) return 0;
    if (end - p > 65536) p = end - 65536;
    s->dictionary = p;
    s->dictSize = (uint32_t)(end - p);
    s->tableType = TT_U32;
    uint32_t idx = s->currentOffset - s->dictSize;
    for (const uint8_t* q = p; q <= end - 8; q += 3, idx += 3)
        s->table[pos_hash(q, TT_U32)] = idx;                     /* later entries overwrite */
    if (slow) {
        const uint32_t limit = s->currentOffset - 65536;
        idx = s->currentOffset - s->dictSize;
        for (const uint8_t* q = p; q <= end - 8; q++, idx++) {
            const uint32_t h = pos_hash(q, TT_U32);
            if (s->table[h] <= limit) s->table[h] = idx;         /* only fills untouched slots */
        }
    }
    return (int)s->dictSize;
}

/* lz4.c:1658-1684 */
void orc_stream_attach(orc_stream* s, const orc_stream* dictStream)
{
    const orc_stream* dc = dictStream;
    if (dc != NULL) {
        if (s->currentOffset == 0) s->currentOffset = 65536;
        if (dc->dictSize == 0) dc = NULL;
    }
    s->dictCtx = dc;
}

/* lz4.c:1687-1704 */
static void renorm(orc_stream* s, int nextSize)
{
    if (s->currentOffset + (unsigned)nextSize > 0x80000000u) {
        const uint32_t delta = s->currentOffset - 65536;
        const uint8_t* dictEnd = s->dictionary + s->dictSize;
        for (int i = 0; i < 4096; i++) s->table[i] = (s->table[i] < delta) ? 0 : s->table[i] - delta;
        s->currentOffset = 65536;
        if (s->dictSize > 65536) s->dictSize = 65536;
        s->dictionary = dictEnd - s->dictSize;
    }
}

/* lz4.c:1707-1783, accel = 1.  Always byU32 + limitedOutput. */
int orc_stream_compress(orc_stream* s, const uint8_t* src, int n, uint8_t* dst, int cap)
{
    const uint8_t* dictEnd = s->dictSize ? s->dictionary + s->dictSize : NULL;
    renorm(s, n);

    if (s->dictSize < 4 && dictEnd != src && n > 0 && s->dictCtx == NULL) {          /* :1723-1733 */
        s->dictSize = 0; s->dictionary = src; dictEnd = src;
    }
    {   const uint8_t* const srcEnd = src + n;                                       /* :1736-1743 */
        if (srcEnd > s->dictionary && srcEnd < dictEnd) {
            s->dictSize = (uint32_t)(dictEnd - srcEnd);
            if (s->dictSize > 65536) s->dictSize = 65536;
            if (s->dictSize < 4) s->dictSize = 0;
            s->dictionary = dictEnd - s->dictSize;
        }
    }
    if (dictEnd == src) {                                                            /* :1746-1751 */
        const int small = (s->dictSize < 65536) && (s->dictSize < s->currentOffset);
        return encode_generic(s, src, n, dst, cap, 1, TT_U32, DM_PREFIX, small);
    }
    {   int r;                                                                       /* :1754-1782 */
        if (s->dictCtx) {
            if (n > 4096) {
                const orc_stream* dc = s->dictCtx;
                memcpy(s, dc, sizeof(*s));
                r = encode_generic(s, src, n, dst, cap, 1, TT_U32, DM_EXT, 0);
            } else {
                r = encode_generic(s, src, n, dst, cap, 1, TT_U32, DM_CTX, 0);
            }
        } else {
            const int small = (s->dictSize < 65536) && (s->dictSize < s->currentOffset);
            r = encode_generic(s, src, n, dst, cap, 1, TT_U32, DM_EXT, small);
        }
        s->dictionary = src;
        s->dictSize = (uint32_t)n;
        return r;
    }
}

/* ================================================================== LZ4 block decoder
 * lz4.c:1978-2014 (length bytes), :2022-2445 (LZ4_decompress_generic, full-block, noDict / usingExtDict).
 * The reference runs a "fast loop" while >= 64 output bytes remain and a "safe loop" after; they
 * reject at different points, so both are restated, as one loop with a `fast` flag. */

/* read_variable_length: returns -1 on error; *ip is advanced exactly as the reference advances it. */
static int64_t more_len(const uint8_t* src, int* ip, int ilimit, int initialCheck)
{
    int64_t len = 0; unsigned b;
    if (initialCheck && *ip >= ilimit) return -1;
    do {
        b = src[(*ip)++];
        len += b;
        if (*ip > ilimit) return -1;
    } while (b == 255);
    return len;
}

static void copy_back(uint8_t* dst, int64_t op, int64_t offset, int64_t len)
{
    /* offset 0 is not an error for liblz4: both its copy routines zero-fill (lz4.c:499-507, :2406-2414) */
    if (offset == 0) { memset(dst + op, 0, (size_t)len); return; }
    for (int64_t i = 0; i < len; i++) dst[op + i] = dst[op - offset + i];
}

static int decode_generic(const uint8_t* src, int n, uint8_t* dst, int cap,
                          const uint8_t* dict, int64_t dictSize)
{
    if (src == NULL || cap < 0) return -1;                                           /* :2036 */
    const int ext = (dict != NULL && dictSize > 0);
    const uint8_t* const dictEnd = ext ? dict + dictSize : NULL;
    const int checkOffset = dictSize < 65536;                                        /* :2047 */
    const int iend = n;
    const int64_t oend = cap;
    int     ip = 0;
    int64_t op = 0;

    if (cap == 0) return (n == 1 && src[0] == 0) ? 0 : -1;                           /* :2064-2068 */
    if (n == 0) return -1;                                                           /* :2069 */

    int fast = (oend - op) >= 64;                                                    /* :2076 */

    for (;;) {
        const unsigned token = src[ip++];
        int64_t ll = token >> 4, ml, offset, mpos;

        if (fast) {
            /* ---- literals, fast loop (:2092-2115) */
            int toSafeLit = 0;
            if (ll == 15) {
                int64_t a = more_len(src, &ip, iend - 15, 1);
                if (a < 0) goto fail;
                ll += a;
                if (op + ll > oend - 32 || (int64_t)ip + ll > iend - 32) toSafeLit = 1;
            } else if (!(ip <= iend - 17)) {
                toSafeLit = 1;
            }
            if (toSafeLit) { fast = 0; goto safe_literals; }
            memcpy(dst + op, src + ip, (size_t)ll);
            ip += (int)ll; op += ll;

            /* ---- match, fast loop (:2118-2208) */
            offset = ld16(src + ip); ip += 2;
            mpos = op - offset;
            ml = token & 15;
            if (ml == 15) {
                int64_t a = more_len(src, &ip, iend - K_LASTLITERALS + 1, 0);
                if (a < 0) goto fail;
                ml += a + K_MINMATCH;
                if (op + ml >= oend - 64) { fast = 0; goto safe_match; }
            } else {
                ml += K_MINMATCH;
                if (op + ml >= oend - 64) { fast = 0; goto safe_match; }
                if (mpos >= 0 && offset >= 8) { copy_back(dst, op, offset, ml); op += ml; continue; }
            }
            if (checkOffset && mpos + dictSize < 0) goto fail;                       /* :2161 */
            if (ext && mpos < 0) {                                                   /* :2166-2196 */
                if (op + ml > oend - K_LASTLITERALS) goto fail;
                goto dict_copy;
            }
            copy_back(dst, op, offset, ml);
            op += ml;
            continue;
        }

        /* ---- safe loop (:2215-2435) */
        if (ll != 15 && ip < iend - 16 && op <= oend - 32) {                         /* shortcut :2230-2261 */
            memcpy(dst + op, src + ip, (size_t)ll);
            op += ll; ip += (int)ll;
            ml = token & 15;
            offset = ld16(src + ip); ip += 2;
            mpos = op - offset;
            if (ml != 15 && offset >= 8 && mpos >= 0) {
                copy_back(dst, op, offset, ml + K_MINMATCH);
                op += ml + K_MINMATCH;
                continue;
            }
            goto match_len;
        }
        if (ll == 15) {                                                              /* :2264-2270 */
            int64_t a = more_len(src, &ip, iend - 15, 1);
            if (a < 0) goto fail;
            ll += a;
        }
safe_literals:                                                                       /* :2273-2334 */
        if (op + ll > oend - K_MFLIMIT || (int64_t)ip + ll > iend - (2 + 1 + K_LASTLITERALS)) {
            /* must be the final literal run, ending exactly at the end of the input */
            if ((int64_t)ip + ll != iend || op + ll > oend) goto fail;
            memmove(dst + op, src + ip, (size_t)ll);
            ip += (int)ll; op += ll;
            break;
        }
        memcpy(dst + op, src + ip, (size_t)ll);
        ip += (int)ll; op += ll;

        offset = ld16(src + ip); ip += 2;                                            /* :2337-2341 */
        mpos = op - offset;
        ml = token & 15;
match_len:                                                                           /* :2344-2351 */
        if (ml == 15) {
            int64_t a = more_len(src, &ip, iend - K_LASTLITERALS + 1, 0);
            if (a < 0) goto fail;
            ml += a;
        }
        ml += K_MINMATCH;
safe_match:                                                                          /* :2354-2434 */
        if (checkOffset && mpos + dictSize < 0) goto fail;
        if (ext && mpos < 0) {
            if (op + ml > oend - K_LASTLITERALS) goto fail;
            goto dict_copy;
        }
        if (op + ml > oend - 12 && op + ml > oend - K_LASTLITERALS) goto fail;       /* :2421-2423 */
        copy_back(dst, op, offset, ml);
        op += ml;
        continue;

dict_copy:                                                                           /* :2177-2195 == :2365-2383 */
        {
            const int64_t back = -mpos;      /* bytes of the match that lie in the dictionary */
            if (ml <= back) {
                memmove(dst + op, dictEnd - back, (size_t)ml);
                op += ml;
            } else {
                const int64_t rest = ml - back;
                memcpy(dst + op, dictEnd - back, (size_t)back);
                op += back;
                for (int64_t i = 0; i < rest; i++) dst[op + i] = dst[i];
                op += rest;
            }
        }
    }
    return (int)op;

fail:
    return -ip - 1;                                                                  /* :2443 */
}

int orc_decompress_safe(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    return decode_generic(src, n, dst, cap, NULL, 0);
}

int orc_decompress_safe_dict(const uint8_t* src, int n, uint8_t* dst, int cap,
                             const uint8_t* dict, int dictLen)
{
    if (dictLen == 0) return decode_generic(src, n, dst, cap, NULL, 0);              /* :2721-2722 */
    return decode_generic(src, n, dst, cap, dict, dictLen);
}

/* ================================================================== plz4 glue */

/* internal/pkg/blk/blk.go:69-109 + descriptor/data.go:3-16 */
int orc_block_record(const uint8_t* src, int n, int bsz, int blockChecksum, uint8_t* rec)
{
    int c = orc_compress_fast(src, n, rec + 4, bsz);
    uint32_t word;
    if (c == 0) {                                   /* ErrCompress -> store raw, high bit set */
        memcpy(rec + 4, src, (size_t)n);
        c = n;
        word = 0x80000000u | ((uint32_t)c & 0x7FFFFFFFu);
    } else {
        word = (uint32_t)c & 0x7FFFFFFFu;
    }
    st32(rec, word);
    if (blockChecksum) { st32(rec + 4 + c, orc_xxh32(rec + 4, (size_t)c)); return c + 8; }
    return c + 4;
}

/* internal/pkg/header/write.go:23-73, descriptor/flags.go, descriptor/block.go:22-29 */
int orc_frame_header(uint8_t* out, int bsIdx, int linked, int blockChecksum, int contentChecksum,
                     int hasContentSize, uint64_t contentSize, int hasDictId, uint32_t dictId)
{
    int len = 7;
    uint8_t flg = (uint8_t)(1u << 6);               /* version 01 */
    out[0] = 0x04; out[1] = 0x22; out[2] = 0x4d; out[3] = 0x18;
    if (!linked)         flg |= 1u << 5;
    if (blockChecksum)   flg |= 1u << 4;
    if (contentChecksum) flg |= 1u << 2;
    if (hasContentSize) {
        flg |= 1u << 3;
        len = 15;
        for (int i = 0; i < 8; i++) out[6 + i] = (uint8_t)(contentSize >> (8 * i));
    }
    if (hasDictId) {
        flg |= 1u << 0;
        st32(out + len - 1, dictId);
        len += 4;
    }
    out[4] = flg;
    out[5] = (uint8_t)((bsIdx & 7) << 4);
    out[len - 1] = (uint8_t)((orc_xxh32(out + 4, (size_t)(len - 5)) >> 8) & 0xFF);
    return len;
}

static int bs_from_idx(int idx)
{
    switch (idx) { case 4: return 64 << 10; case 5: return 256 << 10; case 6: return 1 << 20; case 7: return 4 << 20; }
    return 0;
}

/* internal/pkg/sync/writer.go:52-122 (Write), :133-167 (Close), :265-290 (_writeFrame) */
int64_t orc_frame_encode(const uint8_t* src, int64_t n, int bsIdx, int blockChecksum, int contentChecksum,
                         uint8_t* out, int64_t outCap)
{
    const int bsz = bs_from_idx(bsIdx);
    int64_t w = 0;
    orc_xxh32_state xs; xs.total = 0; xs.fill = 0;
    if (!bsz || outCap < 19) return -1;
    w += orc_frame_header(out, bsIdx, 0, blockChecksum, contentChecksum, 0, 0, 0, 0);
    for (int64_t off = 0; off < n; off += bsz) {
        const int len = (int)((n - off < bsz) ? (n - off) : bsz);
        if (w + bsz + 8 > outCap) return -1;
        if (contentChecksum) orc_xxh32_update(&xs, src + off, (size_t)len);
        w += orc_block_record(src + off, len, bsz, blockChecksum, out + w);
    }
    if (w + 8 > outCap) return -1;
    st32(out + w, 0); w += 4;                                       /* trailer/trailer.go:10-19 */
    if (contentChecksum) {
        if (xs.total == 0) orc_xxh32_reset(&xs);
        st32(out + w, orc_xxh32_digest(&xs)); w += 4;
    }
    return w;
}

/* header/read.go:26-119, blk/frame.go:54-127, sync/reader.go:49-87, blk/blk.go:50-61 */
int64_t orc_frame_decode(const uint8_t* f, int64_t n, uint8_t* out, int64_t outCap)
{
    int64_t r = 0, w = 0;
    if (n < 7) return ORC_ERR_HEADER_READ;
    if (!(f[0] == 0x04 && f[1] == 0x22 && f[2] == 0x4d && f[3] == 0x18)) return ORC_ERR_MAGIC;
    const uint8_t flg = f[4], bd = f[5];
    if (((flg >> 6) & 3) != 1) return ORC_ERR_VERSION;
    if (flg & 2) return ORC_ERR_RESERVED_BIT;
    if (((bd >> 4) & 7) < 4 || (bd & 0x80) || (bd & 0x0F)) return ORC_ERR_BLOCK_DESCRIPTOR;
    int hlen = 7;
    if (flg & 8) hlen += 8;
    if (flg & 1) hlen += 4;
    if (n < hlen) return ORC_ERR_HEADER_READ;
    if (f[hlen - 1] != (uint8_t)((orc_xxh32(f + 4, (size_t)(hlen - 5)) >> 8) & 0xFF)) return ORC_ERR_HEADER_HASH;
    if (!(flg & 0x20) || (flg & 1)) return ORC_ERR_UNSUPPORTED;   /* linked blocks / dictionary: not this entry point */
    const int bsz = bs_from_idx((bd >> 4) & 7);
    const int blkCheck = (flg >> 4) & 1, srcCheck = (flg >> 2) & 1;
    orc_xxh32_state xs; xs.total = 0; xs.fill = 0;
    r = hlen;
    for (;;) {
        if (n - r < 4) return ORC_ERR_BLOCK_SIZE_READ;
        const uint32_t word = ld32(f + r); r += 4;
        if (word == 0) break;
        int64_t sz = word & 0x7FFFFFFFu;
        if (sz > bsz) return ORC_ERR_BLOCK_SIZE_OVERFLOW;
        const int64_t total = sz + (blkCheck ? 4 : 0);
        if (n - r < total) return ORC_ERR_BLOCK_READ;
        if (blkCheck && ld32(f + r + sz) != orc_xxh32(f + r, (size_t)sz)) return ORC_ERR_BLOCK_HASH;
        if (word & 0x80000000u) {
            if (w + sz > outCap) return ORC_ERR_DST_TOO_SMALL;
            memcpy(out + w, f + r, (size_t)sz);
            if (srcCheck) orc_xxh32_update(&xs, out + w, (size_t)sz);
            w += sz;
        } else {
            /* the reference decodes into a pooled bsz+8 buffer (blk/pool.go:23-26, blk.go:51-53) */
            static _Thread_local uint8_t tmp[(4 << 20) + 8];
            const int d = orc_decompress_safe(f + r, (int)sz, tmp, bsz + 8);
            if (d < 0) return ORC_ERR_DECOMPRESS;
            if (w + d > outCap) return ORC_ERR_DST_TOO_SMALL;
            memcpy(out + w, tmp, (size_t)d);
            if (srcCheck) orc_xxh32_update(&xs, tmp, (size_t)d);
            w += d;
        }
        r += total;
    }
    if (srcCheck) {
        if (n - r < 4) return ORC_ERR_CONTENT_HASH_READ;
        if (xs.total == 0) orc_xxh32_reset(&xs);
        if (ld32(f + r) != orc_xxh32_digest(&xs)) return ORC_ERR_CONTENT_HASH;
    }
    return w;
}
