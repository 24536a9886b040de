// lz4hc_device.inl -- LZ4 HC, every level plz4 routes to LZ4_compress_HC (2..12), for one independent block.
//
//   hc_compress      ==  LZ4_compress_HC(src, dst, n, cap, level)
//      /root/reference/internal/pkg/clz4/lz4hc.c:1519-1535 -> :1500-1510 -> :242-259 (init, indices start at 64 KiB)
//      -> :1373-1415 with the level table :92-106:
//        level 2       hc_compress_mid    LZ4MID_compress :521-775 (two direct-mapped tables, 4- and 7-byte hashes)
//        levels 3..9   hc_compress_chain  LZ4HC_compress_hashChain :1121-1363 (lazy 3-match parser over the hash chain)
//        levels 10..12 hc_compress_opt    LZ4HC_compress_optimal :1823-2123 (BASELINE config 4 = level 12)
//      match finder  LZ4HC_InsertAndGetWiderMatch :884-1104 (patternAnalysis, chainSwap, look-back), LZ4HC_FindLongerMatch
//      :1802-1820, LZ4HC_Insert :781-802, pattern helpers :811-868, LZ4HC_countBack :202-225, price model :1778-1800,
//      LZ4HC_encodeSequence :268-354.
//
// Parity first: this round the parser runs as ONE logical thread per block (all 64 lanes of the wave execute the same
// scalar program on uniform data; loads broadcast, stores coalesce), with its 256 KiB hash/chain tables and the 64 KiB
// price table in a per-wave HBM workspace.  Only the single-segment case exists here (independent block, no dictionary):
// lowLimit == dictLimit == 64 KiB, so every "extDict" / "dictCtx" branch of the reference is dead and is not restated.
// Compiled for the CPU as-is by tests/emu (checked there against the real liblz4 in oracle/_ref).
#pragma once
#include "wave.h"

namespace plz4 {

struct HcOpt { int price, off, mlen, litlen; };                       // LZ4HC_optimal_t, lz4hc.c:1770-1775
struct HcWork {
    uint32_t* hash;     // 32768 entries  (LZ4HC_HASHTABLESIZE, lz4hc.h:226-227)
    uint16_t* chain;    // 65536 entries  (LZ4HC_MAXD, lz4hc.h:222-223)
    HcOpt*    opt;      // LZ4_OPT_NUM + 3 entries (lz4hc.c:76, :1836-1841)
};
enum : int { kHcHashEntries = 32768, kHcChainEntries = 65536, kHcOptNum = 4096, kHcTrailing = 3,
             kHcWorkBytes = kHcHashEntries * 4 + kHcChainEntries * 2 + (kHcOptNum + kHcTrailing + 1) * 16 };
static constexpr uint32_t kHcBase = 65536u;                           // LZ4HC_init_internal: first index (lz4hc.c:252-258)

struct HcState {
    const uint8_t* src; int n;
    HcWork w;
    uint32_t nextToUpdate;
};

DEV uint32_t hc_hash(const uint8_t* p) { return (ld32u(p) * 2654435761u) >> 17; }          // lz4hc.c:120-122 (15 bits)

// LZ4_count (lz4.c:680-703) as a plain loop: equal bytes of a[] and b[], a limited by `limit`
DEV int hc_count(const uint8_t* a, const uint8_t* b, const uint8_t* limit)
{
    const uint8_t* s = a;
    while (a + 8 <= limit) {
        const uint64_t d = ld64u(a) ^ ld64u(b);
        if (d) return (int)(a - s) + (ctz64(d) >> 3);
        a += 8; b += 8;
    }
    while (a < limit && *a == *b) { a++; b++; }
    return (int)(a - s);
}

// LZ4HC_Insert (lz4hc.c:781-802): chain every position below `pos`
DEV void hc_insert(HcState& s, int pos)
{
    const uint32_t target = (uint32_t)pos + kHcBase;
    for (uint32_t idx = s.nextToUpdate; idx < target; ++idx) {
        const uint32_t h = hc_hash(s.src + (idx - kHcBase));
        uint32_t delta = idx - s.w.hash[h];
        if (delta > 65535u) delta = 65535u;
        s.w.chain[idx & 0xFFFFu] = (uint16_t)delta;
        s.w.hash[h] = idx;
    }
    s.nextToUpdate = target;
}

// LZ4HC_countPattern (lz4hc.c:811-845), little-endian 64-bit build
DEV unsigned hc_count_pattern(const uint8_t* ip, const uint8_t* iEnd, uint32_t pattern32)
{
    const uint8_t* const start = ip;
    const uint64_t pattern = (uint64_t)pattern32 | ((uint64_t)pattern32 << 32);
    while (ip + 8 <= iEnd) {
        const uint64_t d = ld64u(ip) ^ pattern;
        if (!d) { ip += 8; continue; }
        return (unsigned)(ip - start) + (unsigned)(ctz64(d) >> 3);
    }
    uint64_t pb = pattern;
    while (ip < iEnd && *ip == (uint8_t)pb) { ip++; pb >>= 8; }
    return (unsigned)(ip - start);
}
// LZ4HC_reverseCountPattern (lz4hc.c:850-866)
DEV unsigned hc_rcount_pattern(const uint8_t* ip, const uint8_t* iLow, uint32_t pattern)
{
    const uint8_t* const start = ip;
    while (ip >= iLow + 4) { if (ld32u(ip - 4) != pattern) break; ip -= 4; }
    int k = 3;
    while (ip > iLow) { if (ip[-1] != (uint8_t)(pattern >> (8 * k))) break; ip--; k--; }
    return (unsigned)(start - ip);
}

struct HcMatch { int len, off, back; };

// LZ4HC_countBack (lz4hc.c:202-225): common bytes before ip / match, as a value <= 0
DEV int hc_count_back(const uint8_t* ip, const uint8_t* match, const uint8_t* iMin, const uint8_t* mMin)
{
    int back = 0;
    const int a = (int)(iMin - ip), b = (int)(mMin - match);
    const int mn = a > b ? a : b;
    while (back - mn > 3) {
        const uint32_t v = ld32u(ip + back - 4) ^ ld32u(match + back - 4);
        if (v) return back - (int)(__builtin_clz(v) >> 3);
        back -= 4;
    }
    while (back > mn && ip[back - 1] == match[back - 1]) back--;
    return back;
}

// LZ4HC_InsertAndGetWiderMatch (lz4hc.c:884-1104) for a single prefix segment: `lowLimit` is iLowLimit (how far the match
// may be extended backwards), `longest` the length to beat, patternAnalysis / chainSwap as in the reference.
DEV HcMatch hc_find_wider(HcState& s, int pos, int lowLimit, int highLimit, int longest, int nbSearches,
                          bool patternAnalysis, bool chainSwap)
{
    const uint8_t* const src = s.src;
    const uint8_t* const ip = src + pos;
    const uint8_t* const iLow = src + lowLimit;
    const uint8_t* const iHigh = src + highLimit;
    const uint32_t ipIndex = (uint32_t)pos + kHcBase;
    const uint32_t lowest = (kHcBase + 65536u > ipIndex) ? kHcBase : ipIndex - 65535u;          // :899-900
    const int lookBack = pos - lowLimit;
    int offset = 0, sBack = 0;
    int attempts = nbSearches;
    uint32_t chainPos = 0;
    const uint32_t pattern = ld32u(ip);
    int repeat = 0;                       // 0 untested, 1 not, 2 confirmed
    size_t srcPatternLength = 0;

    hc_insert(s, pos);
    uint32_t mi = s.w.hash[hc_hash(ip)];

    while (mi >= lowest && attempts > 0) {
        int mlen = 0;
        attempts--;
        {
            const uint8_t* const mp = src + (mi - kHcBase);
            if (ld16u(iLow + longest - 1) == ld16u(mp - lookBack + longest - 1)) {              // :929
                if (ld32u(mp) == pattern) {
                    const int back = lookBack ? hc_count_back(ip, mp, iLow, src) : 0;
                    mlen = kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, iHigh);
                    mlen -= back;
                    if (mlen > longest) { longest = mlen; offset = (int)(ipIndex - mi); sBack = back; }
                }
            }
        }
        if (chainSwap && mlen == longest) {                                                      // :964-987
            if (mi + (uint32_t)longest <= ipIndex) {
                uint32_t distNext = 1;
                const int end = longest - kMinMatch + 1;
                int step = 1, accel = 1 << 4;
                for (int p2 = 0; p2 < end; p2 += step) {
                    const uint32_t cd = s.w.chain[(mi + (uint32_t)p2) & 0xFFFFu];
                    step = (accel++ >> 4);
                    if (cd > distNext) { distNext = cd; chainPos = (uint32_t)p2; accel = 1 << 4; }
                }
                if (distNext > 1) {
                    if (distNext > mi) break;
                    mi -= distNext;
                    continue;
                }
            }
        }
        {
            const uint32_t dn = s.w.chain[mi & 0xFFFFu];
            if (patternAnalysis && dn == 1 && chainPos == 0) {                                   // :989-1062
                const uint32_t mci = mi - 1;
                if (repeat == 0) {
                    if (((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24))) {
                        repeat = 2;
                        srcPatternLength = hc_count_pattern(ip + 4, iHigh, pattern) + 4;
                    } else repeat = 1;
                }
                if (repeat == 2 && mci >= lowest) {           // LZ4HC_protectDictEnd is always true above the prefix start (:873-876)
                    const uint8_t* const mp = src + (mci - kHcBase);
                    if (ld32u(mp) == pattern) {
                        const size_t fwd = hc_count_pattern(mp + 4, iHigh, pattern) + 4;
                        size_t back = hc_rcount_pattern(mp, src, pattern);
                        {   const uint32_t far = mci - (uint32_t)back;                           // limit to lowestMatchIndex, :1022-1023
                            back = mci - (far > lowest ? far : lowest); }
                        const size_t seg = back + fwd;
                        if (seg >= srcPatternLength && fwd <= srcPatternLength) {
                            mi = mci + (uint32_t)fwd - (uint32_t)srcPatternLength;               // :1027-1036
                        } else {
                            mi = mci - (uint32_t)back;                                           // :1038-1058
                            if (lookBack == 0) {
                                const size_t maxML = seg < srcPatternLength ? seg : srcPatternLength;
                                if ((size_t)longest < maxML) {
                                    if (ipIndex - mi > 65535u) break;
                                    longest = (int)maxML;
                                    offset = (int)(ipIndex - mi);
                                }
                                const uint32_t dp = s.w.chain[mi & 0xFFFFu];
                                if (dp > mi) break;
                                mi -= dp;
                            }
                        }
                        continue;
                    }
                }
            }
        }
        mi -= s.w.chain[(mi + chainPos) & 0xFFFFu];                                              // :1065
    }
    HcMatch m; m.len = longest; m.off = offset; m.back = sBack;
    return m;
}

// LZ4HC_FindLongerMatch (lz4hc.c:1802-1820): forward-only search with pattern analysis and chain swap
DEV HcMatch hc_find_longer(HcState& s, int pos, int highLimit, int minLen, int nbSearches)
{
    HcMatch m = hc_find_wider(s, pos, pos, highLimit, minLen, nbSearches, true, true);
    if (m.len <= minLen) { m.len = 0; m.off = 0; }                                               // :1815
    return m;
}

DEV int hc_lit_price(int ll) { return ll >= 15 ? ll + 1 + (ll - 15) / 255 : ll; }                // :1778-1785
DEV int hc_seq_price(int ll, int ml)                                                             // :1788-1800
{
    int price = 1 + 2 + hc_lit_price(ll);
    if (ml >= 19) price += 1 + (ml - 19) / 255;
    return price;
}

// LZ4HC_encodeSequence (lz4hc.c:268-354).  Returns true on output overflow (limited mode).
DEV bool hc_encode_seq(const uint8_t* src, int* ip, uint8_t* dst, int* op, int* anchor, int ml, int off, bool limited, int oend)
{
    const int lit = *ip - *anchor;
    const int tok = (*op)++;
    if (limited && (int64_t)*op + lit / 255 + lit + (2 + 1 + kLastLiterals) > oend) return true;
    if (lit >= 15) {
        int r = lit - 15;
        dst[tok] = 0xF0;
        for (; r >= 255; r -= 255) dst[(*op)++] = 255;
        dst[(*op)++] = (uint8_t)r;
    } else dst[tok] = (uint8_t)(lit << 4);
    for (int i = 0; i < lit; ++i) dst[*op + i] = src[*anchor + i];
    *op += lit;
    st16u(dst + *op, (uint16_t)off); *op += 2;
    int r = ml - kMinMatch;
    if (limited && (int64_t)*op + r / 255 + (1 + kLastLiterals) > oend) return true;
    if (r >= 15) {
        dst[tok] = (uint8_t)(dst[tok] + 15);
        r -= 15;
        for (; r >= 510; r -= 510) { dst[(*op)++] = 255; dst[(*op)++] = 255; }
        if (r >= 255) { r -= 255; dst[(*op)++] = 255; }
        dst[(*op)++] = (uint8_t)r;
    } else dst[tok] = (uint8_t)(dst[tok] + r);
    *ip += ml;
    *anchor = *ip;
    return false;
}

// Last literals of every HC strategy (lz4hc.c:1325-1352, :719-746), limitedOutput / notLimited only.  Returns the
// compressed size, or 0 when the run does not fit.
DEV int hc_last_literals(const uint8_t* src, int n, int anchor, uint8_t* dst, int op, bool limited, int oend)
{
    const int last = n - anchor;
    const int llAdd = (last + 255 - 15) / 255;
    if (limited && (int64_t)op + 1 + llAdd + last > oend) return 0;
    if (last >= 15) {
        int r = last - 15;
        dst[op++] = 0xF0;
        for (; r >= 255; r -= 255) dst[op++] = 255;
        dst[op++] = (uint8_t)r;
    } else dst[op++] = (uint8_t)(last << 4);
    for (int i = 0; i < last; ++i) dst[op + i] = src[anchor + i];
    return op + last;
}

DEV void hc_reset_tables(HcWork w)
{
    // LZ4_initStreamHC zeroes the whole state (hash AND chain), lz4hc.c:1582-1583
    for (int i = 0; i < kHcHashEntries; ++i) w.hash[i] = 0;
    for (int i = 0; i < kHcChainEntries; ++i) w.chain[i] = 0;
}

// Levels 3..9: LZ4HC_compress_hashChain (lz4hc.c:1121-1363), nbSearches = 1 << (level - 1) (table :92-106),
// patternAnalysis only above 128 attempts (level 9).  The goto structure of the reference is kept: it IS the algorithm.
DEV int hc_compress_chain(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap, const int level, HcWork w)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;                                             // :1388
    const bool limited = cap < compress_bound(n);
    const int  maxNb = 1 << (level - 1);
    const bool pa = maxNb > 128;
    hc_reset_tables(w);
    HcState s; s.src = src; s.n = n; s.w = w; s.nextToUpdate = kHcBase;

    int ip = 0, anchor = 0, op = 0;
    const int oend = cap;
    const int mflimit = n - kMfLimit;
    const int matchlimit = n - kLastLiterals;
    const int kOptimalMl = 15 - 1 + kMinMatch;                                                   // OPTIMAL_ML, lz4hc.c:75
    int start0 = 0, start2 = 0, start3 = 0;
    HcMatch m0 = {0, 0, 0}, m1 = {0, 0, 0}, m2 = {0, 0, 0}, m3 = {0, 0, 0};

    if (n < kMinLength) goto last_literals;                                                      // :1155
    while (ip <= mflimit) {
        m1 = hc_find_wider(s, ip, ip, matchlimit, kMinMatch - 1, maxNb, pa, false);              // :1159
        if (m1.len < kMinMatch) { ip++; continue; }
        start0 = ip; m0 = m1;
search2:
        if (ip + m1.len <= mflimit) {                                                            // :1167-1175
            start2 = ip + m1.len - 2;
            m2 = hc_find_wider(s, start2, ip, matchlimit, m1.len, maxNb, pa, false);
            start2 += m2.back;
        } else { m2.len = 0; m2.off = 0; m2.back = 0; }
        if (m2.len <= m1.len) {                                                                  // :1177-1184
            if (hc_encode_seq(src, &ip, dst, &op, &anchor, m1.len, m1.off, limited, oend)) return 0;
            continue;
        }
        if (start0 < ip) {                                                                       // :1186-1189
            if (start2 < ip + m0.len) { ip = start0; m1 = m0; }
        }
        if (start2 - ip < 3) { ip = start2; m1 = m2; goto search2; }                             // :1192-1196
search3:
        if (start2 - ip < kOptimalMl) {                                                          // :1199-1210
            int new_ml = m1.len;
            if (new_ml > kOptimalMl) new_ml = kOptimalMl;
            if (ip + new_ml > start2 + m2.len - kMinMatch) new_ml = (start2 - ip) + m2.len - kMinMatch;
            const int correction = new_ml - (start2 - ip);
            if (correction > 0) { start2 += correction; m2.len -= correction; }
        }
        if (start2 + m2.len <= mflimit) {                                                        // :1212-1220
            start3 = start2 + m2.len - 3;
            m3 = hc_find_wider(s, start3, start2, matchlimit, m2.len, maxNb, pa, false);
            start3 += m3.back;
        } else { m3.len = 0; m3.off = 0; m3.back = 0; }
        if (m3.len <= m2.len) {                                                                  // :1222-1240
            if (start2 < ip + m1.len) m1.len = start2 - ip;
            if (hc_encode_seq(src, &ip, dst, &op, &anchor, m1.len, m1.off, limited, oend)) return 0;
            ip = start2;
            if (hc_encode_seq(src, &ip, dst, &op, &anchor, m2.len, m2.off, limited, oend)) return 0;
            continue;
        }
        if (start3 < ip + m1.len + 3) {                                                          // :1242-1270
            if (start3 >= ip + m1.len) {
                if (start2 < ip + m1.len) {
                    const int correction = ip + m1.len - start2;
                    start2 += correction;
                    m2.len -= correction;
                    if (m2.len < kMinMatch) { start2 = start3; m2 = m3; }
                }
                if (hc_encode_seq(src, &ip, dst, &op, &anchor, m1.len, m1.off, limited, oend)) return 0;
                ip = start3; m1 = m3;
                start0 = start2; m0 = m2;
                goto search2;
            }
            start2 = start3; m2 = m3;
            goto search3;
        }
        if (start2 < ip + m1.len) {                                                              // :1277-1291
            if (start2 - ip < kOptimalMl) {
                if (m1.len > kOptimalMl) m1.len = kOptimalMl;
                if (ip + m1.len > start2 + m2.len - kMinMatch) m1.len = (start2 - ip) + m2.len - kMinMatch;
                const int correction = m1.len - (start2 - ip);
                if (correction > 0) { start2 += correction; m2.len -= correction; }
            } else m1.len = start2 - ip;
        }
        if (hc_encode_seq(src, &ip, dst, &op, &anchor, m1.len, m1.off, limited, oend)) return 0;
        ip = start2; m1 = m2;                                                                    // :1299-1306
        start2 = start3; m2 = m3;
        goto search3;
    }
last_literals:
    return hc_last_literals(src, n, anchor, dst, op, limited, oend);
}

// Level 2: LZ4MID_compress (lz4hc.c:521-775).  Two 16 K-entry tables inside the HC hash table: 4-byte hashes in the lower
// half, 7-of-8-byte hashes in the upper half (:141-151, :532-533).
DEV uint32_t mid_hash4(const uint8_t* p) { return (ld32u(p) * 2654435761u) >> (32 - 14); }
DEV uint32_t mid_hash8(const uint8_t* p) { return (uint32_t)(((ld64u(p) << 8) * 58295818150454627ull) >> (64 - 14)); }

DEV int hc_compress_mid(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap, HcWork w)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;
    const bool limited = cap < compress_bound(n);
    hc_reset_tables(w);
    uint32_t* const h4t = w.hash;
    uint32_t* const h8t = w.hash + 16384;
    int ip = 0, anchor = 0, op = 0;
    const int oend = cap;
    const int mflimit = n - kMfLimit;
    const int matchlimit = n - kLastLiterals;
    const uint32_t ilimitIdx = (uint32_t)(n - 8) + kHcBase;
    const uint8_t* const mlim = src + matchlimit;

    if (n >= kMinLength) while (ip <= mflimit) {
        const uint32_t ipIndex = (uint32_t)ip + kHcBase;
        int ml = 0; uint32_t dist = 0;
        {   // long match (:572-601); candidates below the prefix start do not exist for an independent block
            const uint32_t h8 = mid_hash8(src + ip);
            const uint32_t pos8 = h8t[h8];
            h8t[h8] = ipIndex;
            if (ipIndex - pos8 <= 65535u && pos8 >= kHcBase) {
                ml = hc_count(src + ip, src + (pos8 - kHcBase), mlim);
                if (ml >= kMinMatch) dist = ipIndex - pos8; else ml = 0;
            }
        }
        if (!ml) {   // short match, then one look at ip+1 for a longer one (:603-650)
            const uint32_t h4 = mid_hash4(src + ip);
            const uint32_t pos4 = h4t[h4];
            h4t[h4] = ipIndex;
            if (ipIndex - pos4 <= 65535u && pos4 >= kHcBase) {
                ml = hc_count(src + ip, src + (pos4 - kHcBase), mlim);
                if (ml >= kMinMatch) {
                    const uint32_t h8 = mid_hash8(src + ip + 1);
                    const uint32_t pos8 = h8t[h8];
                    const uint32_t m2d = ipIndex + 1 - pos8;
                    dist = ipIndex - pos4;
                    if (m2d <= 65535u && pos8 >= kHcBase && ip < mflimit) {
                        const int ml2 = hc_count(src + ip + 1, src + (pos8 - kHcBase), mlim);
                        if (ml2 > ml) { h8t[h8] = ipIndex + 1; ip++; ml = ml2; dist = m2d; }
                    }
                } else ml = 0;
            }
        }
        if (!ml) { ip += 1 + ((ip - anchor) >> 9); continue; }                                   // :668
        while (ip > anchor && (uint32_t)ip > dist && src[ip - 1] == src[ip - (int)dist - 1]) { ip--; ml++; }   // :673-675
        // "fill table with beginning of match": the index stays the loop-top ipIndex although ip may have moved (:678-680)
        h8t[mid_hash8(src + ip + 1)] = ipIndex + 1;
        h8t[mid_hash8(src + ip + 2)] = ipIndex + 2;
        h4t[mid_hash4(src + ip + 1)] = ipIndex + 1;
        if (hc_encode_seq(src, &ip, dst, &op, &anchor, ml, (int)dist, limited, oend)) return 0;
        {   const uint32_t endIdx = (uint32_t)ip + kHcBase;                                      // :695-706
            if (endIdx - 2 < ilimitIdx) {
                if (ip > 5) h8t[mid_hash8(src + ip - 5)] = endIdx - 5;
                h8t[mid_hash8(src + ip - 3)] = endIdx - 3;
                h8t[mid_hash8(src + ip - 2)] = endIdx - 2;
                h4t[mid_hash4(src + ip - 2)] = endIdx - 2;
                h4t[mid_hash4(src + ip - 1)] = endIdx - 1;
            }
        }
    }
    return hc_last_literals(src, n, anchor, dst, op, limited, oend);
}

// Level table rows 10..12 (lz4hc.c:92-106): {nbSearches, targetLength}; fullUpdate only at level 12 (:1406).
DEV int hc_compress_opt(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap, const int level, HcWork w)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;                                             // :1388
    const bool limited = cap < compress_bound(n);                                                // :1505-1508
    const int  nbSearches = level >= 12 ? 16384 : (level == 11 ? 512 : 96);
    size_t     sufficient = level >= 12 ? 4096 : (level == 11 ? 128 : 64);
    const bool fullUpdate = level >= 12;
    if (sufficient >= (size_t)kHcOptNum) sufficient = kHcOptNum - 1;                             // :1860
    HcOpt* const opt = w.opt;

    hc_reset_tables(w);
    HcState s; s.src = src; s.n = n; s.w = w; s.nextToUpdate = kHcBase;

    int ip = 0, anchor = 0, op = 0;
    const int oend = cap;
    const int mflimit = n - kMfLimit;
    const int matchlimit = n - kLastLiterals;

    while (ip <= mflimit) {                                                                      // :1863
        const int llen = ip - anchor;
        int best_mlen, best_off, cur, last_match_pos = 0;
        const HcMatch first = hc_find_longer(s, ip, matchlimit, kMinMatch - 1, nbSearches);
        if (first.len == 0) { ip++; continue; }
        if ((size_t)first.len > sufficient) {                                                    // :1871-1882
            if (hc_encode_seq(src, &ip, dst, &op, &anchor, first.len, first.off, limited, oend)) return 0;
            continue;
        }
        for (int r = 0; r < kMinMatch; ++r) {                                                    // :1885-1894
            opt[r].mlen = 1; opt[r].off = 0; opt[r].litlen = llen + r; opt[r].price = hc_lit_price(llen + r);
        }
        for (int ml = kMinMatch; ml <= first.len; ++ml) {                                        // :1896-1909
            opt[ml].mlen = ml; opt[ml].off = first.off; opt[ml].litlen = llen; opt[ml].price = hc_seq_price(llen, ml);
        }
        last_match_pos = first.len;
        for (int a = 1; a <= kHcTrailing; ++a) {                                                 // :1911-1919
            opt[last_match_pos + a].mlen = 1; opt[last_match_pos + a].off = 0; opt[last_match_pos + a].litlen = a;
            opt[last_match_pos + a].price = opt[last_match_pos].price + hc_lit_price(a);
        }
        bool direct = false;
        for (cur = 1; cur < last_match_pos; ++cur) {                                             // :1922-2019
            const int curPos = ip + cur;
            if (curPos > mflimit) break;
            if (fullUpdate) {
                if (opt[cur + 1].price <= opt[cur].price && opt[cur + kMinMatch].price < opt[cur].price + 3) continue;
            } else {
                if (opt[cur + 1].price <= opt[cur].price) continue;
            }
            const HcMatch nm = fullUpdate ? hc_find_longer(s, curPos, matchlimit, kMinMatch - 1, nbSearches)
                                          : hc_find_longer(s, curPos, matchlimit, last_match_pos - cur, nbSearches);
            if (!nm.len) continue;
            if ((size_t)nm.len > sufficient || nm.len + cur >= kHcOptNum) {                      // :1948-1956
                best_mlen = nm.len; best_off = nm.off; last_match_pos = cur + 1; direct = true;
                break;
            }
            {   const int baseLit = opt[cur].litlen;                                             // :1958-1972
                for (int l = 1; l < kMinMatch; ++l) {
                    const int price = opt[cur].price - hc_lit_price(baseLit) + hc_lit_price(baseLit + l);
                    const int pos = cur + l;
                    if (price < opt[pos].price) { opt[pos].mlen = 1; opt[pos].off = 0; opt[pos].litlen = baseLit + l; opt[pos].price = price; }
                }
            }
            for (int ml = kMinMatch; ml <= nm.len; ++ml) {                                       // :1974-2009
                const int pos = cur + ml;
                int price, ll;
                if (opt[cur].mlen == 1) { ll = opt[cur].litlen; price = ((cur > ll) ? opt[cur - ll].price : 0) + hc_seq_price(ll, ml); }
                else { ll = 0; price = opt[cur].price + hc_seq_price(0, ml); }
                if (pos > last_match_pos + kHcTrailing || price <= opt[pos].price) {
                    if (ml == nm.len && last_match_pos < pos) last_match_pos = pos;
                    opt[pos].mlen = ml; opt[pos].off = nm.off; opt[pos].litlen = ll; opt[pos].price = price;
                }
            }
            for (int a = 1; a <= kHcTrailing; ++a) {                                             // :2011-2018
                opt[last_match_pos + a].mlen = 1; opt[last_match_pos + a].off = 0; opt[last_match_pos + a].litlen = a;
                opt[last_match_pos + a].price = opt[last_match_pos].price + hc_lit_price(a);
            }
        }
        if (!direct) {                                                                           // :2022-2024
            best_mlen = opt[last_match_pos].mlen; best_off = opt[last_match_pos].off;
            cur = last_match_pos - best_mlen;
        }
        {   // reverse traversal: mark the chosen path (:2026-2046)
            int cand = cur, selML = best_mlen, selOff = best_off;
            for (;;) {
                const int nextML = opt[cand].mlen, nextOff = opt[cand].off;
                opt[cand].mlen = selML; opt[cand].off = selOff;
                selML = nextML; selOff = nextOff;
                if (nextML > cand) break;
                cand -= nextML;
            }
        }
        {   // emit the recorded sequences in order (:2048-2064)
            int r = 0;
            while (r < last_match_pos) {
                const int ml = opt[r].mlen, off = opt[r].off;
                if (ml == 1) { ip++; r++; continue; }
                r += ml;
                if (hc_encode_seq(src, &ip, dst, &op, &anchor, ml, off, limited, oend)) return 0;
            }
        }
    }
    return hc_last_literals(src, n, anchor, dst, op, limited, oend);                             // :2067-2098
}

// LZ4_compress_HC(level) for every level plz4 routes to the HC entry point (2..12); LZ4HC_getCLevelParams, lz4hc.c:108-117
DEV int hc_compress(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap, int level, HcWork w)
{
    if (level < 1) level = 9;
    if (level > 12) level = 12;
    if (level <= 2) return hc_compress_mid(src, n, dst, cap, w);
    if (level <= 9) return hc_compress_chain(src, n, dst, cap, level, w);
    return hc_compress_opt(src, n, dst, cap, level, w);
}


}  // namespace plz4
