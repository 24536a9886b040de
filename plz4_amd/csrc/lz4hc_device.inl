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
// State of this file (round 3).  Independent blocks without dictionary up to 4 MiB -- the frame path's blocks -- no longer run
// the parsers of this file at any level: levels 2..11 are lz4hc_lazy_device.inl (level 2: batches over the two tables; 3..11:
// segments walked at once, stitched; records through the emit stage of level 1), level 12 is lz4hc12_device.inl.  What still
// runs here, as ONE logical thread per block (all 64 lanes execute the same scalar program on uniform data, following the
// reference's control flow decision for decision): every level with a dictionary or linked blocks (hc_compress_mid,
// hc_compress_chain, hc_compress_opt: the tables -- 256 KiB hash + chain, 64 KiB prices -- built while parsing, in a per-wave HBM
// workspace) and raw-API blocks above 4 MiB.  What is wave-parallel in them and shared with the new paths: the match finder
// hc_find_wider_lists (up to 63 candidates of a chain per round: levels 10..11's searches, level 9's pattern-analysis cases),
// match counting, chainSwap link reads, the optimal parser's price update.
//
// Dictionaries and linked blocks (clz4.StreamCtxHC clz4.go:191-209, StreamLinkedCtxHC :250-283) come in the two
// shapes LZ4_compress_HC_continue can take under plz4 (lz4hc.c:1438-1461, :1626-1720):
//   kHcExt  the block has an external segment of <= 64 KiB in front of it: a linked block after LZ4_loadDictHC(previous
//           tail) + LZ4HC_setExternalDict, or a block > 4 KiB under an attached dictionary (the dictionary context's state is
//           copied, then setExternalDict).  Both leave the same state: indices 64 KiB.. over the segment, inserted up to its
//           last 3 bytes (hash chain) or LZ4MID_fillHTable (level 2), the block's prefix starting right behind.  The caller
//           lays the segment out IMMEDIATELY BEFORE the block in memory, so "position" below counts from the segment's first
//           byte and every two-segment count of the reference is one count over contiguous bytes; what the reference does
//           differently for candidates inside the segment is kept (no 2-byte pre-check, back-extension limits, the
//           protectDictEnd rule, level 2 not matching across the boundary).
//   kHcCtx  a block <= 4 KiB under an attached dictionary (usingDictCtxHc): own empty tables, and the dictionary context's
//           tables (read-only, built once per dictionary by hc_prime_dict) searched after the own chain ends.
// Compiled for the CPU as-is by tests/emu (checked there against the real liblz4 in oracle/_ref).
#pragma once
#include "wave.h"

namespace plz4 {

struct HcOpt { int price, off, mlen, litlen; };                       // LZ4HC_optimal_t, lz4hc.c:1770-1775
struct HcWork {
    uint32_t* hash;     // 32768 entries  (LZ4HC_HASHTABLESIZE, lz4hc.h:226-227)
    uint16_t* chain;    // 65536 entries  (LZ4HC_MAXD, lz4hc.h:222-223)
    HcOpt*    opt;      // LZ4_OPT_NUM + 3 entries (lz4hc.c:76, :1836-1841)
    // Independent blocks without dictionary: when a position is searched every position below it has been inserted
    // (LZ4HC_Insert runs up to it, :914), so the chain is a function of the data alone and can be built for the whole block up
    // front (hc12_build_lists, lz4hc12_device.inl): pre[q] = distance from q to the previous position with q's hash, 0 = none
    // within 65535.  With `pre` set the parsers neither reset nor fill the two tables above (4 M dependent table updates per
    // 4 MiB block): the head candidate of a search is pos - pre[pos], a chain link is pre[] again.
    const uint16_t* pre;
    // ... and with the positions of every hash laid out as one ascending run (lz4hc12_device.inl: list, rank[q] = where q sits
    // in it, the first position of a hash flagged): the candidates of a chain are consecutive entries, so a search looks at up to
    // 63 of them at once, one per lane (hc_find_wider_lists).  Null: one candidate after the other.
    const uint32_t* rank; const uint32_t* list;
};
enum : int { kHcHashEntries = 32768, kHcChainEntries = 65536, kHcOptNum = 4096, kHcTrailing = 3,
             kHcWorkBytes = kHcHashEntries * 4 + kHcChainEntries * 2 + (kHcOptNum + kHcTrailing + 1) * 16 };
static constexpr uint32_t kHcBase = 65536u;                           // LZ4HC_init_internal: first index (lz4hc.c:252-258)

enum : int { kHcNone = 0, kHcExt = 1, kHcCtx = 2 };
struct HcDict {                 // how a block is primed (see the header comment)
    int mode;                   // kHcNone / kHcExt / kHcCtx
    int len;                    // kHcExt: bytes of the segment lying right before the block; kHcCtx: dictionary length
    const uint8_t*  bytes;      // kHcCtx: the dictionary
    const uint32_t* hash;       // kHcCtx: the dictionary context's tables (hc_prime_dict over `bytes`)
    const uint16_t* chain;
};
struct HcState {
    const uint8_t* src;         // first byte of the segment (== the block when there is none); positions count from here
    int pfx;                    // bytes of the segment: the block's prefix starts at position pfx, index kHcBase + pfx
    HcWork w;
    uint32_t nextToUpdate;
    HcDict d;
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
DEV uint32_t hc_link(const HcState& s, uint32_t idx)       // chain table entry of index idx (lz4hc.c:228 DELTANEXTU16)
{
    if (s.w.pre) {
        // (an external segment's last three positions are never inserted, lz4hc.c:1660-1678: their entries of the freshly zeroed
        // table -- LZ4_loadDictHC starts from LZ4_initStreamHC, :1626-1653 -- stay 0; the chain swap may read them, :964-987)
        if (s.d.mode == kHcExt && (kHcBase + (uint32_t)s.pfx - 1u) - idx < 3u) return 0u;
        const uint32_t d = s.w.pre[idx - kHcBase]; return d ? d : 65535u;
    }
    return s.w.chain[idx & 0xFFFFu];
}
DEV void hc_insert(HcState& s, int pos)
{
    const uint32_t target = (uint32_t)pos + kHcBase;
    if (s.w.pre) { s.nextToUpdate = target; return; }
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

// LZ4HC_protectDictEnd (lz4hc.c:876-879): false for the last 3 indices of the external segment
DEV bool hc_protect(uint32_t prefixIdx, uint32_t mi) { return (uint32_t)((prefixIdx - 1u) - mi) >= 3u; }


// LZ4HC_InsertAndGetWiderMatch for the hash-chain levels (no chain swap) on an independent block, with the chain's candidates
// taken from `list`: up to 63 candidates per round, one per lane -- their 2-byte filter and 4-byte test (:925-932) and their
// chain links are evaluated together, and the first lane where something happens (the tests pass: the match is measured; the
// link is 1: pattern analysis, level 9) is handled exactly as the reference handles that candidate; the lanes below it were
// candidates that change nothing.  One dependent memory round trip per ROUND instead of one per candidate.
// kD: the block has an external segment of s.pfx bytes in front of it (kHcExt) whose positions are in the lists as well -- all but
// its last three, which the reference never inserts.  What the reference does differently for a candidate inside the segment is
// kept: no 2-byte pre-check (:940-960), the look-back may go down to the segment's first byte while a candidate in the block stops
// at the block's (:933 vs :953), and the pattern analysis never lands on the segment's last three positions (LZ4HC_protectDictEnd,
// :876-879, :1003-1060).
template <bool kD = false>
DEV HcMatch hc_find_wider_lists(HcState& s, int pos, int lowLimit, int highLimit, int longest, int nbSearches, bool patternAnalysis, bool chainSwap)
{
    const uint8_t* const src = s.src;
    const int pfxPos = kD ? s.pfx : 0;
    const uint32_t prefixIdx = kHcBase + (uint32_t)pfxPos;
    const uint8_t* const ip = src + pos;
    const uint8_t* const iLow = src + lowLimit;
    const uint8_t* const iHigh = src + highLimit;
    const uint32_t ipIndex = (uint32_t)pos + kHcBase;
    const bool     within = kHcBase + 65536u > ipIndex;
    const uint32_t lowest = within ? kHcBase : ipIndex - 65535u;
    const int lookBack = pos - lowLimit;
    const uint32_t pattern = UNI(ld32u(ip));
    int offset = 0, sBack = 0, attempts = nbSearches, repeat = 0;
    size_t srcPatternLength = 0;
    uint32_t chainPos = 0;       // chain swap (:964-987): the walk follows the chain of position candidate + chainPos
    // the 8 bytes behind a candidate's prefix come with the prefix: a candidate that differs from us within them has a known
    // forward length (4..11) -- no counting loop for it, and no event at all when that length (+ lookBack) cannot reach `longest`
    const bool useFw = iHigh - ip >= 12;
    const uint64_t ip64 = useFw ? ld64u(ip + 4) : 0ull;
    HcMatch out;
    const uint32_t head = UNI((uint32_t)s.w.pre[pos]);
    const uint32_t rank0 = UNI(s.w.rank[pos]);               // (requested with the head: one memory round trip)
    if (head) {
        uint32_t mi = ipIndex - head;                        // the current candidate ...
        int cursor = (int)rank0 - 1;                         // ... and where it sits in the list
        while (mi >= lowest && attempts > 0) {
            // lane k: the candidate k places below the current one in the chain (list[cursor - k]); 64 entries are fetched so
            // that every one of the 63 candidates knows its link (the distance to the next candidate)
            LV(uint32_t, q); LV(int, fl); LV(int, pat); LV(uint32_t, ent);
            {
                const int curL = cursor;
                LANES({ const uint32_t e = (curL - LANE >= -8) ? s.w.list[curL - LANE] : 0x80000000u; ent[I_] = e; q[I_] = e & 0x7FFFFFFFu; fl[I_] = (int)(e >> 31); })
            }
            LV(uint32_t, qn);
            LANES({ qn[I_] = SHFL(ent, LANE + 1) & 0x7FFFFFFFu; })
            // a candidate exists while the chain has not ended above it, it is inside the window, and attempts are left; its
            // link is what the chain table would hold (saturated at 65535, which also ends the walk, :1065 -> :918)
            const uint64_t ends = BALLOT(fl[I_] != 0);                          // the chain's first position sits in this lane
            const int lastOfChain = ends ? ctz64(ends) : 64;
            const int attL = attempts;
            const int cp = (int)chainPos;
            const uint64_t dead = BALLOT(LANE > lastOfChain || (int)q[I_] - cp + (int)kHcBase < (int)lowest || LANE >= attL || LANE >= 63);
            const int nvalid = dead ? ctz64(dead) : 63;                          // lane 0 is the current candidate: valid by the loop test
            LV(uint32_t, dn);
            LANES({
                const bool hasNext = LANE < lastOfChain;
                const uint32_t d = hasNext ? q[I_] - qn[I_] : 65535u;
                dn[I_] = d < 65535u ? d : 65535u;
            })
            // the candidates' 4-byte prefixes, once per round; the 2 bytes at `longest - 1` (:921) again whenever `longest` grows
            // (both requested together the first time: one memory round trip)
            LV(int, pfx); LV(int, pass); LV(int, fw);
            {
                const uint32_t ip16 = UNI((uint32_t)ld16u(iLow + longest - 1));
                const int nv = nvalid, L = longest;
                LANES({
                    pfx[I_] = 0; pass[I_] = 0; pat[I_] = 0; fw[I_] = 9;                   // 0..7: differs there; 8: the 8 bytes are equal; 9: not looked at
                    if (LANE < nv) {
                        const uint8_t* mp = src + q[I_] - cp;
                        const uint32_t m16 = ld16u(mp - lookBack + L - 1);
                        if (useFw) { const uint64_t x = ld64u(mp + 4) ^ ip64; fw[I_] = x ? (ctz64(x) >> 3) : 8; }
                        pfx[I_] = ld32u(mp) == pattern;
                        // (inside the segment: no 2-byte pre-check, but -- the chain swap can lead there -- not its last three positions, :943)
                        const int cq = (int)q[I_] - cp;
                        pass[I_] = pfx[I_] && ((kD && cq < pfxPos) ? cq <= pfxPos - 4 : m16 == ip16) && !(fw[I_] < 8 && 4 + fw[I_] + lookBack < L);
                        pat[I_] = patternAnalysis && dn[I_] == 1u && cp == 0;
                    }
                })
            }
            // the walk also ends inside the round where a link is saturated: the candidate behind it is never looked at
            const uint64_t sat = BALLOT(LANE < nvalid && dn[I_] >= 65535u);
            const int kSat = sat ? ctz64(sat) : 64;
            const int endLane = kSat < nvalid ? kSat + 1 : nvalid;                // lanes [0, endLane) are what the walk can reach
            int start = 0;                                                        // lanes below it are done
            int passFor = longest;                                                // the `longest` that pass[] was made for
            bool walkEnds = false, nextRound = false;
            while (!walkEnds && !nextRound) {
                if (passFor != longest) {
                    const uint32_t ip16 = UNI((uint32_t)ld16u(iLow + longest - 1));
                    const int L = longest, st = start, en = endLane;
                    LANES({
                        pass[I_] = 0;
                        if (LANE >= st && LANE < en && pfx[I_] && !(fw[I_] < 8 && 4 + fw[I_] + lookBack < L))
                            pass[I_] = (kD && (int)q[I_] - cp < pfxPos) ? ((int)q[I_] - cp <= pfxPos - 4) : (ld16u(src + q[I_] - cp - lookBack + L - 1) == ip16);
                    })
                    passFor = longest;
                }
                const int st0 = start;
                const uint64_t ev = BALLOT(LANE >= st0 && LANE < endLane && (pass[I_] || pat[I_]));
                const int k = ev ? ctz64(ev) : 64;
                if (k == 64) {
                    // every remaining candidate of the round changes nothing
                    attempts -= endLane - start;
                    if (kSat < nvalid || endLane > lastOfChain) { walkEnds = true; break; }   // saturated link / the chain's first position consumed
                    mi = RL(q, endLane - 1) - RL(dn, endLane - 1) - chainPos + kHcBase;       // the candidate below the last one looked at
                    cursor -= endLane;
                    nextRound = true;
                    break;
                }
                attempts -= k - start + 1;
                mi = RL(q, k) - chainPos + kHcBase;
                const uint32_t dnk = RL(dn, k);
                const uint8_t* const mp = src + (mi - kHcBase);
                if (RL(pass, k)) {                                                    // :933-939 / :940-960
                    const int back = lookBack ? hc_count_back(ip, mp, iLow, (kD && mi >= prefixIdx) ? src + pfxPos : src) : 0;
                    const int fwk = RL(fw, k);
                    int mlen = fwk < 8 ? kMinMatch + fwk
                             : fwk == 8 ? kMinMatch + 8 + hc_count(ip + kMinMatch + 8, mp + kMinMatch + 8, iHigh)
                             : kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, iHigh);
                    mlen -= back;
                    if (mlen > longest) { longest = mlen; offset = (int)(ipIndex - mi); sBack = back; }
                    if (chainSwap && mlen == longest && mi + (uint32_t)longest <= ipIndex) {               // :964-987, as in hc_find_wider
                        uint32_t distNext = 1;
                        const int end = longest - kMinMatch + 1;
                        int step = 1, accel = 1 << 4;
                        LV(uint32_t, links);
                        { const uint32_t mi0 = mi; LANES({ links[I_] = (LANE < end) ? hc_link(s, mi0 + (uint32_t)LANE) : 0u; }) }
                        for (int p2 = 0; p2 < end; p2 += step) {
                            const uint32_t cd = p2 < 64 ? RL(links, p2) : hc_link(s, mi + (uint32_t)p2);
                            step = (accel++ >> 4);
                            if (cd > distNext) { distNext = cd; chainPos = (uint32_t)p2; accel = 1 << 4; }
                        }
                        if (distNext > 1) {
                            if (distNext > mi || distNext >= 65535u) { walkEnds = true; break; }   // (a saturated link leads below the window: the walk ends)
                            cursor = (int)UNI(s.w.rank[mi - kHcBase + chainPos]) - 1;
                            mi -= distNext;
                            nextRound = true;
                            break;
                        }
                    }
                }
                if (RL(pat, k)) {                                                     // :989-1062 (chainPos is 0 here)
                    const uint32_t mci = mi - 1;
                    if (repeat == 0) {
                        if (((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24))) {
                            repeat = 2;
                            srcPatternLength = hc_count_pattern(ip + 4, iHigh, pattern) + 4;
                        } else repeat = 1;
                    }
                    if (repeat == 2 && mci >= lowest && (!kD || hc_protect(prefixIdx, mci))) {
                        const uint8_t* const mq = src + (mci - kHcBase);
                        if (UNI(ld32u(mq)) == pattern) {
                            const size_t fwd = hc_count_pattern(mq + 4, iHigh, pattern) + 4;
                            size_t back = hc_rcount_pattern(mq, src, pattern);
                            {   const uint32_t far = mci - (uint32_t)back;
                                back = mci - (far > lowest ? far : lowest); }
                            const size_t seg = back + fwd;
                            if (seg >= srcPatternLength && fwd <= srcPatternLength) {
                                mi = mci + (uint32_t)fwd - (uint32_t)srcPatternLength;     // :1027-1036: looked at next
                                if (kD && !hc_protect(prefixIdx, mi)) mi = prefixIdx;
                                cursor = (int)UNI(s.w.rank[mi - kHcBase]);
                            } else if (kD && !hc_protect(prefixIdx, mci - (uint32_t)back)) {
                                mi = prefixIdx;                                            // :1040-1042
                                cursor = (int)UNI(s.w.rank[mi - kHcBase]);
                            } else {
                                mi = mci - (uint32_t)back;                                 // :1038-1058
                                if (lookBack == 0) {
                                    const size_t maxML = seg < srcPatternLength ? seg : srcPatternLength;
                                    if ((size_t)longest < maxML) {
                                        if (ipIndex - mi > 65535u) { walkEnds = true; break; }
                                        longest = (int)maxML; offset = (int)(ipIndex - mi);
                                    }
                                    const uint32_t dp = hc_link(s, mi);
                                    if (dp > mi) { walkEnds = true; break; }
                                    const uint32_t at = mi - kHcBase;
                                    mi -= dp; cursor = (int)UNI(s.w.rank[at]) - 1;
                                    if (dp >= 65535u) { walkEnds = true; break; }
                                } else cursor = (int)UNI(s.w.rank[mi - kHcBase]);
                            }
                            nextRound = true;                                          // the walk goes on somewhere else
                            break;
                        }
                    }
                }
                if (dnk >= 65535u || k >= lastOfChain) { walkEnds = true; break; }    // the walk ends behind this candidate
                start = k + 1;                                                        // :1065: on to the next candidate ...
                if (start >= endLane) {                                               // ... which is beyond what this round fetched
                    if (start >= attL) { walkEnds = true; break; }                    // (no attempts left: attempts is 0 now)
                    mi -= dnk;
                    cursor -= k + 1;
                    nextRound = true;
                }
            }
            if (walkEnds) break;
        }
    }
    out.len = longest; out.off = offset; out.back = sBack;
    return out;
}

// LZ4HC_InsertAndGetWiderMatch (lz4hc.c:884-1104): `lowLimit` is iLowLimit (how far the match may be extended backwards),
// `longest` the length to beat, patternAnalysis / chainSwap as in the reference.  Positions count from s.src.
// kD = false: an independent block without a dictionary (no segment, no context): every branch for those folds away.
template <bool kD>
DEV HcMatch hc_find_wider(HcState& s, int pos, int lowLimit, int highLimit, int longest, int nbSearches,
                          bool patternAnalysis, bool chainSwap)
{
    if (!kD && s.w.list && nbSearches >= 8)                     // levels 4..12 on an independent block: 63 candidates per round
        return hc_find_wider_lists<false>(s, pos, lowLimit, highLimit, longest, nbSearches, patternAnalysis, chainSwap);
    if (kD && s.w.list && s.d.mode == kHcExt)                   // ... and behind an external segment whose positions are in the lists too
        return hc_find_wider_lists<true>(s, pos, lowLimit, highLimit, longest, nbSearches, patternAnalysis, chainSwap);
    const uint8_t* const src = s.src;
    const uint8_t* const ip = src + pos;
    const uint8_t* const iLow = src + lowLimit;
    const uint8_t* const iHigh = src + highLimit;
    const int pfx = kD ? s.pfx : 0;
    const uint8_t* const prefixPtr = src + pfx;
    const uint32_t prefixIdx = kHcBase + (uint32_t)pfx;
    const uint32_t ipIndex = (uint32_t)pos + kHcBase;
    const bool     within = kHcBase + 65536u > ipIndex;                                         // withinStartDistance, :898
    const uint32_t lowest = within ? kHcBase : ipIndex - 65535u;                                 // :899
    const int lookBack = pos - lowLimit;
    int offset = 0, sBack = 0;
    int attempts = nbSearches;
    uint32_t chainPos = 0;
    const uint32_t pattern = ld32u(ip);
    int repeat = 0;                       // 0 untested, 1 not, 2 confirmed
    size_t srcPatternLength = 0;

    hc_insert(s, pos);
    uint32_t mi = s.w.pre ? (s.w.pre[pos] ? ipIndex - s.w.pre[pos] : 0u) : s.w.hash[hc_hash(ip)];

    while (mi >= lowest && attempts > 0) {
        int mlen = 0;
        attempts--;
        // the chain link of this candidate is requested together with its bytes: one memory round trip per hop, not two
        const uint32_t dn0 = hc_link(s, mi);
        {
            const uint8_t* const mp = src + (mi - kHcBase);
            const uint32_t m32 = ld32u(mp);       // requested with the 2-byte probe and the chain link, whether or not it gets looked at
            if (!kD || mi >= prefixIdx) {                                                        // within the prefix, :925-939
                if (ld16u(iLow + longest - 1) == ld16u(mp - lookBack + longest - 1)) {
                    if (m32 == pattern) {
                        const int back = lookBack ? hc_count_back(ip, mp, iLow, prefixPtr) : 0;
                        mlen = kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, iHigh);
                        mlen -= back;
                        if (mlen > longest) { longest = mlen; offset = (int)(ipIndex - mi); sBack = back; }
                    }
                }
            } else if (mi <= prefixIdx - 4u && m32 == pattern) {                                 // within the segment, :940-960
                // count to the segment's end, then on into the prefix: one count over contiguous bytes
                mlen = kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, iHigh);
                const int back = lookBack ? hc_count_back(ip, mp, iLow, src) : 0;
                mlen -= back;
                if (mlen > longest) { longest = mlen; offset = (int)(ipIndex - mi); sBack = back; }
            }
        }
        if (chainSwap && mlen == longest) {                                                      // :964-987
            if (mi + (uint32_t)longest <= ipIndex) {
                uint32_t distNext = 1;
                const int end = longest - kMinMatch + 1;
                int step = 1, accel = 1 << 4;
                // the links of the first 64 positions of the match in one load (lane l: position l) instead of one memory
                // round trip per position; a longer match goes on with single loads
                LV(uint32_t, links);
                { const uint32_t mi0 = mi; LANES({ links[I_] = (LANE < end) ? hc_link(s, mi0 + (uint32_t)LANE) : 0u; }) }
                for (int p2 = 0; p2 < end; p2 += step) {
                    const uint32_t cd = p2 < 64 ? RL(links, p2) : hc_link(s, mi + (uint32_t)p2);
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
            const uint32_t dn = dn0;
            if (patternAnalysis && dn == 1 && chainPos == 0) {                                   // :989-1062
                const uint32_t mci = mi - 1;
                if (repeat == 0) {
                    if (((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24))) {
                        repeat = 2;
                        srcPatternLength = hc_count_pattern(ip + 4, iHigh, pattern) + 4;
                    } else repeat = 1;
                }
                if (repeat == 2 && mci >= lowest && (!kD || hc_protect(prefixIdx, mci))) {
                    const uint8_t* const mp = src + (mci - kHcBase);
                    if (ld32u(mp) == pattern) {
                        // forward to the segment's end and on into the prefix with the rotated pattern (:1009-1013), backwards
                        // from the prefix into the segment (:1017-1021): the same periodic run over contiguous bytes
                        const size_t fwd = hc_count_pattern(mp + 4, iHigh, pattern) + 4;
                        size_t back = hc_rcount_pattern(mp, src, pattern);
                        {   const uint32_t far = mci - (uint32_t)back;                           // limit to lowestMatchIndex, :1022-1023
                            back = mci - (far > lowest ? far : lowest); }
                        const size_t seg = back + fwd;
                        if (seg >= srcPatternLength && fwd <= srcPatternLength) {
                            const uint32_t nmi = mci + (uint32_t)fwd - (uint32_t)srcPatternLength;   // :1027-1036
                            mi = (!kD || hc_protect(prefixIdx, nmi)) ? nmi : prefixIdx;
                        } else {
                            const uint32_t nmi = mci - (uint32_t)back;                           // :1038-1058
                            if (kD && !hc_protect(prefixIdx, nmi)) mi = prefixIdx;
                            else {
                                mi = nmi;
                                if (lookBack == 0) {
                                    const size_t maxML = seg < srcPatternLength ? seg : srcPatternLength;
                                    if ((size_t)longest < maxML) {
                                        if (ipIndex - mi > 65535u) break;
                                        longest = (int)maxML;
                                        offset = (int)(ipIndex - mi);
                                    }
                                    const uint32_t dp = hc_link(s, mi);
                                    if (dp > mi) break;
                                    mi -= dp;
                                }
                            }
                        }
                        continue;
                    }
                }
            }
        }
        mi -= chainPos == 0 ? dn0 : hc_link(s, mi + chainPos);                                   // :1065
    }
    if (kD && s.d.mode == kHcCtx && attempts > 0 && within) {                                          // usingDictCtxHc, :1069-1098
        const uint32_t dictEnd = kHcBase + (uint32_t)s.d.len;        // the context's end index: its own indices start at 64 KiB
        uint32_t dmi = s.d.hash[hc_hash(ip)];
        mi = dmi + lowest - dictEnd;
        while (ipIndex - mi <= 65535u && attempts--) {
            const uint8_t* const mp = s.d.bytes + (dmi - kHcBase);
            if (ld32u(mp) == pattern) {
                const uint8_t* vLimit = ip + (dictEnd - dmi);
                if (vLimit > iHigh) vLimit = iHigh;
                int mlt = kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, vLimit);
                const int back = lookBack ? hc_count_back(ip, mp, iLow, s.d.bytes) : 0;
                mlt -= back;
                if (mlt > longest) { longest = mlt; offset = (int)(ipIndex - mi); sBack = back; }
            }
            const uint32_t nx = s.d.chain[dmi & 0xFFFFu];
            dmi -= nx; mi -= nx;
        }
    }
    HcMatch m; m.len = longest; m.off = offset; m.back = sBack;
    return m;
}

// LZ4HC_FindLongerMatch (lz4hc.c:1802-1820): forward-only search with pattern analysis and chain swap
template <bool kD>
DEV HcMatch hc_find_longer(HcState& s, int pos, int highLimit, int minLen, int nbSearches)
{
    HcMatch m = hc_find_wider<kD>(s, pos, pos, highLimit, minLen, nbSearches, true, true);
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

// Level 2 keeps two 16 K-entry tables inside the HC hash table: 4-byte hashes in the lower half, 7-of-8-byte hashes in the
// upper half (lz4hc.c:141-151, :532-533).
DEV uint32_t mid_hash4(const uint8_t* p) { return (ld32u(p) * 2654435761u) >> (32 - 14); }
DEV uint32_t mid_hash8(const uint8_t* p) { return (uint32_t)(((ld64u(p) << 8) * 58295818150454627ull) >> (64 - 14)); }

// Fresh tables, then what LZ4_loadDictHC leaves over the segment [0, pfx) (lz4hc.c:1626-1653): LZ4MID_fillHTable (:477-503) at
// level 2, LZ4HC_Insert up to the segment's last 3 bytes otherwise; LZ4HC_setExternalDict (:1660-1678) then resumes
// referencing at the prefix.  The same state is what a dictionary context carries (clz4.go:122-147), so this also builds the
// tables of HcDict for kHcCtx (pfx = dictionary length).
DEV void hc_prime(HcState& s, int level)
{
    if (s.w.pre && level > 2) { s.nextToUpdate = kHcBase + (uint32_t)s.pfx; return; }    // (independent block, no segment: nothing to prime)
    hc_reset_tables(s.w);
    s.nextToUpdate = kHcBase;
    const int size = s.pfx;
    if (level <= 2) {
        uint32_t* const h4t = s.w.hash;
        uint32_t* const h8t = s.w.hash + 16384;
        if (size > 8) {
            const uint32_t target = kHcBase + (uint32_t)size - 8u;
            uint32_t idx = kHcBase;
            for (; idx < target; idx += 3) {
                h4t[mid_hash4(s.src + (idx - kHcBase))] = idx;
                h8t[mid_hash8(s.src + (idx + 1 - kHcBase))] = idx + 1;
            }
            idx = (size > 32768 + 8) ? target - 32768u : kHcBase;
            for (; idx < target; ++idx) h8t[mid_hash8(s.src + (idx - kHcBase))] = idx;
        }
    } else if (size >= 4) {
        hc_insert(s, size - 3);
    }
    s.nextToUpdate = kHcBase + (uint32_t)size;
}

// Levels 3..9: LZ4HC_compress_hashChain (lz4hc.c:1121-1363), nbSearches = 1 << (level - 1) (table :92-106),
// patternAnalysis only above 128 attempts (level 9).  The goto structure of the reference is kept: it IS the algorithm.
template <bool kD>
DEV int hc_compress_chain(const uint8_t* src, const int pfxArg, const int n, uint8_t* __restrict__ dst, const int cap, const int level, HcWork w, const HcDict& d)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;                                             // :1388
    const bool limited = cap < compress_bound(n);
    const int  maxNb = 1 << (level - 1);
    const bool pa = maxNb > 128;
    const int  pfx = kD ? pfxArg : 0;
    HcState s; s.src = src; s.pfx = pfx; s.w = w; s.d = d;
    hc_prime(s, level);

    const int N = pfx + n;                                                                       // positions count from the segment
    int ip = pfx, anchor = pfx, op = 0;
    const int oend = cap;
    const int mflimit = N - kMfLimit;
    const int matchlimit = N - kLastLiterals;
    const int kOptimalMl = 15 - 1 + kMinMatch;                                                   // OPTIMAL_ML, lz4hc.c:75
    int start0 = 0, start2 = 0, start3 = 0;
    HcMatch m0 = {0, 0, 0}, m1 = {0, 0, 0}, m2 = {0, 0, 0}, m3 = {0, 0, 0};

    if (n < kMinLength) goto last_literals;                                                      // :1155
    while (ip <= mflimit) {
        m1 = hc_find_wider<kD>(s, ip, ip, matchlimit, kMinMatch - 1, maxNb, pa, false);              // :1159
        if (m1.len < kMinMatch) { ip++; continue; }
        start0 = ip; m0 = m1;
search2:
        if (ip + m1.len <= mflimit) {                                                            // :1167-1175
            start2 = ip + m1.len - 2;
            m2 = hc_find_wider<kD>(s, start2, ip, matchlimit, m1.len, maxNb, pa, false);
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
            m3 = hc_find_wider<kD>(s, start3, start2, matchlimit, m2.len, maxNb, pa, false);
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
    return hc_last_literals(src, N, anchor, dst, op, limited, oend);
}

// Level 2: LZ4MID_compress (lz4hc.c:521-775).

// One candidate of a level-2 table: length of the match at `ip` against index `pos` (0: none).  A candidate inside the
// segment is counted up to the segment's end only (safeLen, lz4hc.c:587-596, :637-646); `prefixOnly` is the ip+1 look (:616).
DEV int mid_try(const uint8_t* src, int ip, uint32_t ipIndex, uint32_t pos, uint32_t prefixIdx, const uint8_t* mlim, bool prefixOnly)
{
    if (ipIndex - pos > 65535u) return 0;
    if (pos >= prefixIdx) return hc_count(src + ip, src + (pos - kHcBase), mlim);
    if (prefixOnly || pos < kHcBase) return 0;
    const uint8_t* lim = src + ip + (prefixIdx - pos);
    if (lim > mlim) lim = mlim;
    return hc_count(src + ip, src + (pos - kHcBase), lim);
}

// LZ4MID_searchExtDict (lz4hc.c:420-470): the level-2 tables of a dictionary context, long hash first.  gDictEnd = kHcBase.
DEV HcMatch mid_search_ctx(const HcDict& d, const uint8_t* ipp, uint32_t ipIndex, const uint8_t* mlim)
{
    const uint32_t lEnd = kHcBase + (uint32_t)d.len;
    HcMatch m; m.len = 0; m.off = 0; m.back = 0;
    for (int k = 0; k < 2; ++k) {
        const uint32_t l = k == 0 ? d.hash[16384 + mid_hash8(ipp)] : d.hash[mid_hash4(ipp)];
        const uint32_t mIdx = l + kHcBase - lEnd;
        if (ipIndex - mIdx <= 65535u) {
            const uint8_t* const mp = d.bytes + (l - kHcBase);
            size_t safe = (size_t)(lEnd - l);
            if ((size_t)(mlim - ipp) < safe) safe = (size_t)(mlim - ipp);
            const int mlt = hc_count(ipp, mp, ipp + safe);
            if (mlt >= kMinMatch) { m.len = mlt; m.off = (int)(ipIndex - mIdx); return m; }
        }
    }
    return m;
}

template <bool kD>
DEV int hc_compress_mid(const uint8_t* src, const int pfxArg, const int n, uint8_t* __restrict__ dst, const int cap, HcWork w, const HcDict& d)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;
    const bool limited = cap < compress_bound(n);
    const int  pfx = kD ? pfxArg : 0;
    HcState s; s.src = src; s.pfx = pfx; s.w = w; s.d = d;
    hc_prime(s, 2);
    uint32_t* const h4t = w.hash;
    uint32_t* const h8t = w.hash + 16384;
    const int N = pfx + n;
    int ip = pfx, anchor = pfx, op = 0;
    const int oend = cap;
    const int mflimit = N - kMfLimit;
    const int matchlimit = N - kLastLiterals;
    const uint32_t prefixIdx = kHcBase + (uint32_t)pfx;
    const uint32_t ilimitIdx = (uint32_t)(N - 8) + kHcBase;
    const uint8_t* const mlim = src + matchlimit;

    if (n >= kMinLength) while (ip <= mflimit) {
        const uint32_t ipIndex = (uint32_t)ip + kHcBase;
        int ml = 0; uint32_t dist = 0;
        const uint32_t h4 = mid_hash4(src + ip);
        const uint32_t pos4 = h4t[h4];        // read together with the long table's entry (its slot is only written further down)
        {   // long match (:572-601)
            const uint32_t h8 = mid_hash8(src + ip);
            const uint32_t pos8 = h8t[h8];
            h8t[h8] = ipIndex;
            ml = mid_try(src, ip, ipIndex, pos8, prefixIdx, mlim, false);
            if (ml >= kMinMatch) dist = ipIndex - pos8; else ml = 0;
        }
        if (!ml) {   // short match, then one look at ip+1 for a longer one inside the prefix (:603-650)
            h4t[h4] = ipIndex;
            ml = mid_try(src, ip, ipIndex, pos4, prefixIdx, mlim, false);
            if (ml >= kMinMatch) {
                dist = ipIndex - pos4;
                if (pos4 >= prefixIdx) {
                    const uint32_t h8 = mid_hash8(src + ip + 1);
                    const uint32_t pos8 = h8t[h8];
                    const uint32_t m2d = ipIndex + 1 - pos8;
                    if (m2d <= 65535u && pos8 >= prefixIdx && ip < mflimit) {
                        const int ml2 = hc_count(src + ip + 1, src + (pos8 - kHcBase), mlim);
                        if (ml2 > ml) { h8t[h8] = ipIndex + 1; ip++; ml = ml2; dist = m2d; }
                    }
                }
            } else ml = 0;
        }
        if (kD && !ml && d.mode == kHcCtx && ipIndex - kHcBase < 65535u - 8u) {                        // :652-665
            const HcMatch dm = mid_search_ctx(d, src + ip, ipIndex, mlim);
            if (dm.len >= kMinMatch) { ml = dm.len; dist = (uint32_t)dm.off; }
        }
        if (!ml) { ip += 1 + ((ip - anchor) >> 9); continue; }                                   // :668
        while (ip > anchor && (uint32_t)(ip - pfx) > dist && src[ip - 1] == src[ip - (int)dist - 1]) { ip--; ml++; }   // :673-675
        // "fill table with beginning of match": the index stays the loop-top ipIndex although ip may have moved (:678-680)
        h8t[mid_hash8(src + ip + 1)] = ipIndex + 1;
        h8t[mid_hash8(src + ip + 2)] = ipIndex + 2;
        h4t[mid_hash4(src + ip + 1)] = ipIndex + 1;
        if (hc_encode_seq(src, &ip, dst, &op, &anchor, ml, (int)dist, limited, oend)) return 0;
        {   const uint32_t endIdx = (uint32_t)ip + kHcBase;                                      // :695-706
            if (endIdx - 2 < ilimitIdx) {
                if (ip - pfx > 5) h8t[mid_hash8(src + ip - 5)] = endIdx - 5;
                h8t[mid_hash8(src + ip - 3)] = endIdx - 3;
                h8t[mid_hash8(src + ip - 2)] = endIdx - 2;
                h4t[mid_hash4(src + ip - 2)] = endIdx - 2;
                h4t[mid_hash4(src + ip - 1)] = endIdx - 1;
            }
        }
    }
    return hc_last_literals(src, N, anchor, dst, op, limited, oend);
}

// Level table rows 10..12 (lz4hc.c:92-106): {nbSearches, targetLength}; fullUpdate only at level 12 (:1406).
template <bool kD>
DEV int hc_compress_opt(const uint8_t* src, const int pfxArg, const int n, uint8_t* __restrict__ dst, const int cap, const int level, HcWork w, const HcDict& d)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;                                             // :1388
    const bool limited = cap < compress_bound(n);                                                // :1505-1508
    const int  nbSearches = level >= 12 ? 16384 : (level == 11 ? 512 : 96);
    size_t     sufficient = level >= 12 ? 4096 : (level == 11 ? 128 : 64);
    const bool fullUpdate = level >= 12;
    if (sufficient >= (size_t)kHcOptNum) sufficient = kHcOptNum - 1;                             // :1860
    HcOpt* const opt = w.opt;

    const int pfx = kD ? pfxArg : 0;
    HcState s; s.src = src; s.pfx = pfx; s.w = w; s.d = d;
    hc_prime(s, level);

    const int N = pfx + n;
    int ip = pfx, anchor = pfx, op = 0;
    const int oend = cap;
    const int mflimit = N - kMfLimit;
    const int matchlimit = N - kLastLiterals;

    while (ip <= mflimit) {                                                                      // :1863
        const int llen = ip - anchor;
        int best_mlen, best_off, cur, last_match_pos = 0;
        const HcMatch first = hc_find_longer<kD>(s, ip, matchlimit, kMinMatch - 1, nbSearches);
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
            const HcMatch nm = fullUpdate ? hc_find_longer<kD>(s, curPos, matchlimit, kMinMatch - 1, nbSearches)
                                          : hc_find_longer<kD>(s, curPos, matchlimit, last_match_pos - cur, nbSearches);
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
            {   // :1974-2009.  The match lengths are independent of each other (each one reads and writes its own opt[cur + ml];
                // last_match_pos only moves at the last one), so the lanes take one length each: one memory round trip for
                // the whole loop instead of one per length.
                const int ll = (opt[cur].mlen == 1) ? opt[cur].litlen : 0;
                const int basePrice = (opt[cur].mlen == 1) ? ((cur > ll) ? opt[cur - ll].price : 0) : opt[cur].price;
                const int lastOld = last_match_pos;
                for (int ml0 = kMinMatch; ml0 <= nm.len; ml0 += 64) {
                    LV(int, took);
                    LANES({
                        const int ml = ml0 + LANE;
                        took[I_] = 0;
                        if (ml <= nm.len) {
                            const int pos = cur + ml;
                            const int price = basePrice + hc_seq_price(ll, ml);
                            if (pos > lastOld + kHcTrailing || price <= opt[pos].price) {
                                took[I_] = 1;
                                opt[pos].mlen = ml; opt[pos].off = nm.off; opt[pos].litlen = ll; opt[pos].price = price;
                            }
                        }
                    })
                    if (nm.len - ml0 < 64) {                                                     // the lane of the last length is in this chunk
                        const int lastLane = nm.len - ml0;
                        if (RL(took, lastLane) && lastOld < cur + nm.len) last_match_pos = cur + nm.len;
                    }
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
    return hc_last_literals(src, N, anchor, dst, op, limited, oend);                             // :2067-2098
}

// LZ4_compress_HC(level) / LZ4_compress_HC_continue for every level plz4 routes to the HC entry points (2..12);
// LZ4HC_getCLevelParams, lz4hc.c:108-117.  kHcExt: the d.len bytes before `blk` are the external segment.
DEV int hc_compress(const uint8_t* blk, const int n, uint8_t* __restrict__ dst, const int cap, int level, HcWork w, const HcDict& d)
{
    if (level < 1) level = 9;
    if (level > 12) level = 12;
    const int pfx = d.mode == kHcExt ? d.len : 0;
    const uint8_t* const src = blk - pfx;
    if (level <= 2) return hc_compress_mid<true>(src, pfx, n, dst, cap, w, d);
    if (level <= 9) return hc_compress_chain<true>(src, pfx, n, dst, cap, level, w, d);
    return hc_compress_opt<true>(src, pfx, n, dst, cap, level, w, d);
}
// LZ4_compress_HC proper: an independent block, no dictionary (the hot path of BASELINE config 4)
DEV int hc_compress(const uint8_t* __restrict__ blk, const int n, uint8_t* __restrict__ dst, const int cap, int level, HcWork w)
{
    if (level < 1) level = 9;
    if (level > 12) level = 12;
    HcDict d; d.mode = kHcNone; d.len = 0; d.bytes = nullptr; d.hash = nullptr; d.chain = nullptr;
    if (level <= 2) return hc_compress_mid<false>(blk, 0, n, dst, cap, w, d);
    if (level <= 9) return hc_compress_chain<false>(blk, 0, n, dst, cap, level, w, d);
    return hc_compress_opt<false>(blk, 0, n, dst, cap, level, w, d);
}

// The tables a dictionary context carries (clz4.NewDictCtxHC, clz4.go:122-147 == LZ4_loadDictHC): `w` <- tables over dict[0, len).
DEV void hc_prime_dict(const uint8_t* dict, int len, int level, HcWork w)
{
    HcState s; s.src = dict; s.pfx = len; s.w = w;
    s.d.mode = kHcNone; s.d.len = 0; s.d.bytes = nullptr; s.d.hash = nullptr; s.d.chain = nullptr;
    hc_prime(s, level);
}

}  // namespace plz4
