// lz4hc_device.inl -- LZ4 HC "optimal parser" levels 10..12 (BASELINE config 4 = level 12) for one independent block.
//
//   hc_compress_opt  ==  LZ4_compress_HC(src, dst, n, cap, level)  for level in {10, 11, 12}
//      /root/reference/internal/pkg/clz4/lz4hc.c:1519-1535 -> :1500-1510 -> :242-259 (init, indices start at 64 KiB)
//      -> :1373-1415 -> LZ4HC_compress_optimal :1823-2123 with the level table :92-106
//      match finder  LZ4HC_FindLongerMatch :1802-1820 -> LZ4HC_InsertAndGetWiderMatch :884-1104 (patternAnalysis, chainSwap),
//      LZ4HC_Insert :781-802, pattern helpers :811-868, price model :1778-1800, LZ4HC_encodeSequence :268-354.
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

struct HcMatch { int len, off; };

// LZ4HC_FindLongerMatch(ip, iHighLimit, minLen, nbSearches) == LZ4HC_InsertAndGetWiderMatch(ip, iLowLimit = ip, ..., patternAnalysis,
// chainSwap) for a single prefix segment (lz4hc.c:1802-1820, :884-1104).
DEV HcMatch hc_find_longer(HcState& s, int pos, int highLimit, int minLen, int nbSearches)
{
    const uint8_t* const src = s.src;
    const uint8_t* const ip = src + pos;
    const uint8_t* const iHigh = src + highLimit;
    const uint32_t ipIndex = (uint32_t)pos + kHcBase;
    const uint32_t lowest = (kHcBase + 65536u > ipIndex) ? kHcBase : ipIndex - 65535u;          // :899-900
    int longest = minLen, offset = 0;
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
            if (ld16u(ip + longest - 1) == ld16u(mp + longest - 1)) {                            // :929
                if (ld32u(mp) == pattern) {
                    mlen = kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, iHigh);
                    if (mlen > longest) { longest = mlen; offset = (int)(ipIndex - mi); }
                }
            }
        }
        if (mlen == longest) {                                                                   // chain swap, :964-987
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
            if (dn == 1 && chainPos == 0) {                                                      // pattern analysis, :989-1062
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
                        continue;
                    }
                }
            }
        }
        mi -= s.w.chain[(mi + chainPos) & 0xFFFFu];                                              // :1065
    }
    HcMatch m; m.len = longest; m.off = offset;
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

    // LZ4_initStreamHC zeroes the whole state (hash AND chain), lz4hc.c:1582-1583
    for (int i = 0; i < kHcHashEntries; ++i) w.hash[i] = 0;
    for (int i = 0; i < kHcChainEntries; ++i) w.chain[i] = 0;
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
    {   // last literals (:2067-2098), limitedOutput / notLimited only
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
        op += last;
    }
    return op;
}

}  // namespace plz4
