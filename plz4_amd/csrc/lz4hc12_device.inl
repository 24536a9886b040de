// lz4hc12_device.inl -- HC level 12 (LZ4HC_compress_optimal, BASELINE config 4) for independent blocks, re-cut for a GPU.
//
// What liblz4 does (/root/reference/internal/pkg/clz4/lz4hc.c): one thread walks the block; at (nearly) every position it
// inserts the positions behind it into a hash table + a chain of 16-bit deltas (LZ4HC_Insert :781-802) and then walks that
// chain for the longest match (LZ4HC_FindLongerMatch :1802-1820 -> LZ4HC_InsertAndGetWiderMatch :884-1104 with pattern
// analysis and chain swap, up to 16384 candidates); a price DP over windows of <= 4096 positions picks the sequences
// (LZ4HC_compress_optimal :1823-2123).
//
// Two facts make that parallel without changing one decision:
//   (1) when position p is searched, EVERY position below p has been inserted (LZ4HC_Insert runs up to p, :914), whatever the
//       parser did before -- so the chain is a function of the data alone: chain[q] = distance from q to the previous
//       position with the same hash.  It is built for the whole block up front (hc12_build_lists), 64 positions per step --
//       and not as a linked list only: the positions of every hash are also laid out as one ascending run (`list`, with
//       rank[q] = where q sits in it), so that "the next candidates of this chain" are CONSECUTIVE entries a lane fetches
//       eight at a time, instead of one dependent memory access per candidate.
//   (2) level 12 always searches with the same parameters (minLen 3, forward only, :1936), so the search result of a position
//       is a function of the data alone as well: F(p) = (length, offset).  It is computed for every position the parser may
//       ask for, one position per LANE (Hc12Walk), ahead of the parser; the 64 KiB of source behind the positions in flight
//       sit in LDS, so the 2-byte / 4-byte tests of a candidate (95 % of all candidates end there) never leave the CU.
//   The price DP touches no match finder any more (hc12_walk): it reads F, scans the skip test (:1929-1934) for 64 positions at
//   once and updates the prices of a match's lengths one length per lane.  Up to 4 MiB a block is walked in segments at once
//   (lz4hc_lazy_device.inl: at the top of its loop the parser's state is (ip, anchor), so a wave may start there anywhere and
//   the walks are stitched where they meet), its sequences are records and level 1's emit stage writes them; larger raw blocks
//   keep one wave per block that hands the sequences to a writer of its own, 64 at a time.
//
// Everything here is also compiled by g++ for tests/emu (PLZ4_EMU), where the three phases run back to back on the CPU and are
// checked against the real LZ4_compress_HC of oracle/_ref.
#pragma once
#include "wave.h"
#include "lz4_device.inl"
#include "lz4hc_device.inl"

// diagnostics build: the parser's and the search kernel's timers share one counter array, one of the two is compiled in
#if defined(PLZ4_STATS_SEARCH)
#define PSTAT(i, v)    do {} while (0)
#define PSTAT_FLUSH()  do {} while (0)
#define SSTAT(i, v)    STAT(i, v)
#define SSTAT_FLUSH()  STAT_FLUSH()
#else
#define PSTAT(i, v)    STAT(i, v)
#define PSTAT_FLUSH()  STAT_FLUSH()
#define SSTAT(i, v)    do {} while (0)
#define SSTAT_FLUSH()  do {} while (0)
#endif

namespace plz4 {

enum : int { kHc12Sufficient = 4095,           // level 12: sufficient_len 4096, capped to LZ4_OPT_NUM - 1 (lz4hc.c:92-106, :1860)
             kHc12Searches   = 16384,          // level 12: nbSearches
             kHc12NotComputed = -1,            // F.len of a position the search phase skipped: the parser searches it itself
             kHc12OptEntries = kHcOptNum + kHcTrailing + 1 };
struct Hc12F { int32_t len; int32_t off; };    // LZ4HC_FindLongerMatch's answer for one position (len 0: none)

// ------------------------------------------------------------------------------------------ phase 1: chains and lists
// chain[q] (uint16): distance from q to the previous position with the same LZ4HC hash (lz4hc.c:120-122), 1..65535;
// 0 = none within 65535.  liblz4's table holds min(distance, 65535) instead (:793-796); what differs is only that a head
// candidate exactly 65535 back (valid, :913-916 read the hash table's full index) can be told from one further back.
// hc12_raw gives liblz4's value back for the places that compare or subtract chain values.
// list: the positions of hash 0 in ascending order, then those of hash 1, ...; offsets[h] = where hash h starts (a histogram
// + prefix sum, made before); an entry carries kHc12First when it is the first position of its hash.  rank[q] = index of q
// in list.  The chain walk "q, q - chain[q], ..." is list[rank[q]], list[rank[q] - 1], ... down to the entry with kHc12First.
DEV uint32_t hc12_raw(uint32_t d) { return d ? d : 65535u; }
static constexpr uint32_t kHc12First = 0x80000000u;
DEV uint32_t hc12_hash(uint32_t v) { return (v * 2654435761u) >> 17; }

// One wave.  Two passes over the block, one per half of the hash range, so that the two LDS tables (last position + 1 and
// list cursor per hash: 2 x 64 KiB) fit.  Per lane one atomic max (commits the position, returns the previous holder) and one
// atomic add (takes the list slot); same-slot lanes are expected to resolve in ascending lane order, any other order is
// detected and put right in place.  chain[0, nPad) is written (nPad % 64 == 0), rank / list for positions below n - 3.
// [skipLo, skipHi): positions that are never inserted (an external segment's last three, lz4hc.c:1660-1678: LZ4HC_setExternalDict
// references the segment up to its end - 3 and resumes at the prefix); they get no list entry and nobody's chain leads to them.
DEV void hc12_build_lists(const uint8_t* __restrict__ src, const int n, const uint32_t* __restrict__ offsets,
                          uint16_t* __restrict__ chain, uint32_t* __restrict__ rank, uint32_t* __restrict__ list, const int nPad,
                          uint32_t* lastT, uint32_t* curT, const int skipLo = 0, const int skipHi = 0)
{
    const int nIns = n >= 4 ? n - 3 : 0;                       // positions with 4 bytes to hash
    for (int half = 0; half < 2; ++half) {
        LANES({ for (int i = LANE; i < kHcHashEntries / 2; i += 64) { lastT[i] = 0u; curT[i] = offsets[half * (kHcHashEntries / 2) + i]; } })
        LDS_FENCE();
        // the four source bytes of a step's 64 positions are requested steps ahead: a step then costs its two LDS atomics and
        // its stores, not a memory round trip (one wave per CU runs this: nothing else would hide it)
        auto words = [&](LVREF(uint32_t, w), int b) { LANES({ const int p = b + LANE; w[I_] = p < nIns ? ld32u(src + p) : 0u; }) };
        auto step = [&](const int base, LVREF(uint32_t, w)) {
            LV(uint32_t, h); LV(uint32_t, prev); LV(uint32_t, slot); LV(int, act);
            LANES({
                const int p = base + LANE;
                act[I_] = 0; h[I_] = 0; prev[I_] = 0; slot[I_] = 0;
                if (p < nIns && !(p >= skipLo && p < skipHi)) {
                    const uint32_t hv = hc12_hash(w[I_]);
                    if ((int)(hv >> 14) == half) {
                        act[I_] = 1; h[I_] = hv & 16383u;
                        prev[I_] = lds_max_rtn(&lastT[h[I_]], (uint32_t)p + 1u);
                        slot[I_] = lds_add_rtn(&curT[h[I_]], 1u);
                    }
                }
            })
            LDS_ORDER();             // (the atomics' answers are waited for where they are used; no fence: it would wait for the loads ahead too)
            {   // in lane order, a lane whose predecessor is in this batch sits right behind it in the list
                LV(uint32_t, ps);
                LANES({ ps[I_] = SHFL(slot, (prev[I_] > (uint32_t)base) ? (int)(prev[I_] - 1u - (uint32_t)base) : LANE); })
                if (BALLOT(act[I_] && (prev[I_] > (uint32_t)(base + LANE) || (prev[I_] > (uint32_t)base && slot[I_] != ps[I_] + 1u)))) {
                    // some other resolution order: a hash's pre-batch values are the smallest its lanes got back
                    LV(uint32_t, gmin); LV(uint32_t, smin); LV(uint32_t, near); LV(uint32_t, idx);
                    LANES({ gmin[I_] = prev[I_]; smin[I_] = slot[I_]; near[I_] = 0; idx[I_] = 0; })
                    for (int l = 0; l < 64; ++l) {
                        const uint32_t hl = RL(h, l), pl = RL(prev, l), sl = RL(slot, l); const int al = RL(act, l);
                        LANES({
                            if (al && act[I_] && h[I_] == hl) {
                                if (pl < gmin[I_]) gmin[I_] = pl;
                                if (sl < smin[I_]) smin[I_] = sl;
                                if (l < LANE) { near[I_] = (uint32_t)(base + l) + 1u; idx[I_]++; }
                            }
                        })
                    }
                    LANES({ prev[I_] = near[I_] ? near[I_] : gmin[I_]; slot[I_] = smin[I_] + idx[I_]; })
                }
            }
            LANES({
                const int p = base + LANE;
                if (act[I_]) {
                    uint32_t d = 0;
                    if (prev[I_]) { const uint32_t dist = (uint32_t)p + 1u - prev[I_]; d = dist <= 65535u ? dist : 0u; }
                    chain[p] = (uint16_t)d;
                    rank[p] = slot[I_];
                    list[slot[I_]] = (uint32_t)p | (prev[I_] ? 0u : kHc12First);
                } else if ((p >= nIns || (p >= skipLo && p < skipHi)) && half == 0) chain[p] = 0;
            })
        };
        // Sixteen steps' words are in flight: loads and stores of a wave complete in order, so the wait for a word is also a wait
        // for every store issued before its load -- the scattered list stores of the steps before, an L2 round trip each.  With
        // four steps per group the group took one such round trip (877 cycles per step); with sixteen it is shared by sixteen.
#define HC12_W16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define HC12_DECL(k)  LV(uint32_t, w##k); LV(uint32_t, n##k);
#define HC12_FIRST(k) words(w##k, 64 * k);
#define HC12_NEXT(k)  words(n##k, base + 1024 + 64 * k);
#define HC12_STEP(k)  if (base + 64 * k < nPad) step(base + 64 * k, w##k);
#define HC12_MOVE(k)  LANES({ w##k[I_] = n##k[I_]; })
        HC12_W16(HC12_DECL)
        HC12_W16(HC12_FIRST)
        for (int base = 0; base < nPad; base += 1024) {
            HC12_W16(HC12_NEXT)
            HC12_W16(HC12_STEP)
            HC12_W16(HC12_MOVE)
        }
#undef HC12_W16
#undef HC12_DECL
#undef HC12_FIRST
#undef HC12_NEXT
#undef HC12_STEP
#undef HC12_MOVE
        LDS_FENCE();
    }
}

// The chain alone (HcWork::pre for levels 3..11): one pass, `tab` = 32768 x uint32 of LDS (128 KiB): last position + 1 per hash.
DEV void hc12_build_chain(const uint8_t* __restrict__ src, const int n, uint16_t* __restrict__ chain, const int nPad, uint32_t* tab)
{
    LANES({ for (int i = LANE; i < kHcHashEntries; i += 64) tab[i] = 0u; })
    LDS_FENCE();
    const int nIns = n >= 4 ? n - 3 : 0;
    for (int base = 0; base < nPad; base += 64) {
        LV(uint32_t, h); LV(uint32_t, prev); LV(int, act);
        LANES({
            const int p = base + LANE;
            act[I_] = p < nIns; h[I_] = 0; prev[I_] = 0;
            if (act[I_]) { h[I_] = hc12_hash(ld32u(src + p)); prev[I_] = lds_max_rtn(&tab[h[I_]], (uint32_t)p + 1u); }
        })
        LDS_FENCE();
        if (BALLOT(act[I_] && prev[I_] > (uint32_t)(base + LANE))) {      // another resolution order: see hc12_build_lists
            LV(uint32_t, gmin); LV(uint32_t, near);
            LANES({ gmin[I_] = prev[I_]; near[I_] = 0; })
            for (int l = 0; l < 64; ++l) {
                const uint32_t hl = RL(h, l), pl = RL(prev, l); const int al = RL(act, l);
                LANES({ if (al && act[I_] && h[I_] == hl) { if (pl < gmin[I_]) gmin[I_] = pl; if (l < LANE) near[I_] = (uint32_t)(base + l) + 1u; } })
            }
            LANES({ prev[I_] = near[I_] ? near[I_] : gmin[I_]; })
        }
        LANES({
            const int p = base + LANE;
            uint32_t d = 0;
            if (act[I_] && prev[I_]) { const uint32_t dist = (uint32_t)p + 1u - prev[I_]; d = dist <= 65535u ? dist : 0u; }
            chain[p] = (uint16_t)d;
        })
    }
}

// ------------------------------------------------------------------------------------------ phase 2: F(p), one lane per position
// Chain accessors: ch(q) = the stored value of position q.
struct Hc12Flat { const uint16_t* c; DEVM uint32_t operator()(uint32_t q) const { return c[q]; } };
// LZ4HC_FindLongerMatch(ip = src + pos, minLen 3, nbSearches 16384) == LZ4HC_InsertAndGetWiderMatch(lookBack 0,
// patternAnalysis, chainSwap) on an independent block without dictionary (lz4hc.c:884-1065).  Plain per-lane code: every
// lane of a wave runs its own position.  step() evaluates ONE candidate of the chain walk and returns true when the search
// has ended; the state between steps is what the reference keeps in its loop variables.
template <class Chain>
struct Hc12Lane {
    const uint8_t* src; const uint8_t* ip; const uint8_t* iHigh;
    uint32_t ipIndex, lowest, mi, chainPos, pattern, ip16, srcPatLen;
    int longest, offset, attempts, repeat;

    DEVM void init(const uint8_t* s, int n, int pos, uint32_t head)
    {
        src = s; ip = s + pos; iHigh = s + (n - kLastLiterals);
        ipIndex = (uint32_t)pos + kHcBase;
        lowest = (kHcBase + 65536u > ipIndex) ? kHcBase : ipIndex - 65535u;               // :898-899
        mi = head ? ipIndex - head : 0u;                                                   // the hash table's entry for ip (:916)
        chainPos = 0; pattern = ld32u(ip); longest = kMinMatch - 1; offset = 0;
        attempts = kHc12Searches; repeat = 0; srcPatLen = 0;
        ip16 = ld16u(ip + longest - 1);
    }
    DEVM Hc12F result() const { Hc12F f; f.len = longest > kMinMatch - 1 ? longest : 0; f.off = f.len ? offset : 0; return f; }   // :1815

    DEVM bool step(const Chain& ch)
    {
        if (!(mi >= lowest && attempts > 0)) return true;                                  // :918
        attempts--;
        const uint32_t mpos = mi - kHcBase;
        const uint32_t dn0 = hc12_raw(ch(mpos));
        const uint8_t* mp = src + mpos;
        const uint32_t m32 = ld32u(mp);
        int mlen = 0;
        if (ld16u(mp + longest - 1) == ip16 && m32 == pattern) {                           // :925-939 (lookBack 0)
            mlen = kMinMatch + hc_count(ip + kMinMatch, mp + kMinMatch, iHigh);
            if (mlen > longest) { longest = mlen; offset = (int)(ipIndex - mi); ip16 = ld16u(ip + longest - 1); }
        }
        if (mlen == longest) {                                                             // chain swap, :964-987
            if (mi + (uint32_t)longest <= ipIndex) {
                uint32_t distNext = 1;
                const int end = longest - kMinMatch + 1;
                int stepw = 1, accel = 1 << 4;
                for (int p2 = 0; p2 < end; p2 += stepw) {
                    const uint32_t cd = hc12_raw(ch(mpos + (uint32_t)p2));
                    stepw = (accel++ >> 4);
                    if (cd > distNext) { distNext = cd; chainPos = (uint32_t)p2; accel = 1 << 4; }
                }
                if (distNext > 1) {
                    if (distNext > mi) return true;
                    mi -= distNext;
                    return false;
                }
            }
        }
        if (dn0 == 1 && chainPos == 0) {                                                   // pattern analysis, :989-1062
            const uint32_t mci = mi - 1;
            if (repeat == 0) {
                if (((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24))) {
                    repeat = 2;
                    srcPatLen = hc_count_pattern(ip + 4, iHigh, pattern) + 4;
                } else repeat = 1;
            }
            if (repeat == 2 && mci >= lowest) {
                const uint8_t* const mq = src + (mci - kHcBase);
                if (ld32u(mq) == pattern) {
                    const uint32_t fwd = hc_count_pattern(mq + 4, iHigh, pattern) + 4;
                    uint32_t back = hc_rcount_pattern(mq, src, pattern);
                    {   const uint32_t far = mci - back;                                   // :1022-1023
                        back = mci - (far > lowest ? far : lowest); }
                    const uint32_t seg = back + fwd;
                    if (seg >= srcPatLen && fwd <= srcPatLen) {
                        mi = mci + fwd - srcPatLen;                                        // :1027-1036
                    } else {
                        mi = mci - back;                                                   // :1038-1058 (lookBack == 0)
                        const uint32_t maxML = seg < srcPatLen ? seg : srcPatLen;
                        if ((uint32_t)longest < maxML) {
                            if (ipIndex - mi > 65535u) return true;
                            longest = (int)maxML; offset = (int)(ipIndex - mi); ip16 = ld16u(ip + longest - 1);
                        }
                        const uint32_t dp = hc12_raw(ch(mi - kHcBase));
                        if (dp > mi) return true;
                        mi -= dp;
                    }
                    return false;
                }
            }
        }
        mi -= chainPos == 0 ? dn0 : hc12_raw(ch(mpos + chainPos));                         // :1065
        return false;
    }
};


// The same search as the kernel runs it (k_hc12_search): the candidates of a chain come from `list`, eight entries per
// fetch; the bytes of a candidate come from a window of the source in LDS; and the search is cut into PHASES -- a loop trip of
// a wave runs one phase's code for the lanes that are in it, so no lane waits for another lane's inner loop:
//   kPhFilter  up to FOUR candidates: their 2-byte filter and 4-byte test (:925-932) are read together; the first one that
//              passes goes on to kPhCount, one whose chain link is 1 to kPhPattern, the others are behind us.  The next
//              eight entries of the chain are requested a trip before they are needed.
//   kPhCount   LZ4_count, 16 bytes per trip
//   kPhScan    chain swap after an improvement (:964-987): the links of the match's positions
//   kPhRank    after a swap or a pattern jump the walk goes on in another chain: where that position sits in `list`, and
//              the entries below it
//   kPhPattern pattern analysis (:989-1062)
// Positions within 32 bytes of the block's end are not run here (no wide loads there): the kernel leaves them to the parser.
// Decisions and their order are those of Hc12Lane::step; tests/emu checks every position of every input against it.
enum : int { kPhIdle = 0, kPhFilter = 1, kPhCount = 2, kPhScan = 3, kPhPattern = 4, kPhSlow = 5, kPhWait = 6, kPhDone = 7,
             kPhRank = 8 };

DEV uint32_t hc12_bytes16(uint64_t lo, uint64_t hi, int o)      // the two bytes at offset o (0..14) of a 16-byte value
{
    if (o < 7) return (uint32_t)(lo >> (8 * o)) & 0xFFFFu;
    if (o == 7) return ((uint32_t)(lo >> 56) | ((uint32_t)hi << 8)) & 0xFFFFu;
    return (uint32_t)(hi >> (8 * (o - 8))) & 0xFFFFu;
}

// Source bytes: the whole block (tests, and what the LDS window falls back to) ...
struct Hc12SrcFlat {
    const uint8_t* g;
    DEVM uint32_t r16(uint32_t q) const { return ld16u(g + q); }
    DEVM uint32_t r32(uint32_t q) const { return ld32u(g + q); }
    DEVM uint64_t r64(uint32_t q) const { return ld64u(g + q); }
    DEVM uint32_t q16(uint32_t q) const { return ld16u(g + q); }
    DEVM uint32_t q32(uint32_t q) const { return ld32u(g + q); }
};
// ... or a 128 KiB ring in LDS that holds [hi - 131072, hi): position q at q & 0x1FFFF.  r16/r32/r64: reads that would run past
// `hi` (long matches) or across the ring's end go to memory.  q16/q32: for reads known to lie inside the window (the ring's
// first 16 bytes are mirrored behind its end, so a read may straddle it).
enum : int { kHc12SrcRing = 131072, kHc12SrcRingPad = 16 };
struct Hc12SrcRing {
    const uint8_t* ring; const uint8_t* g; uint32_t hi;
    DEVM bool in(uint32_t q, uint32_t len) const { return q + len <= hi && (q & (kHc12SrcRing - 1)) + len <= (uint32_t)kHc12SrcRing; }
    DEVM uint32_t r16(uint32_t q) const { return in(q, 2) ? ld16u(ring + (q & (kHc12SrcRing - 1))) : ld16u(g + q); }
    DEVM uint32_t r32(uint32_t q) const { return in(q, 4) ? ld32u(ring + (q & (kHc12SrcRing - 1))) : ld32u(g + q); }
    DEVM uint64_t r64(uint32_t q) const { return in(q, 8) ? ld64u(ring + (q & (kHc12SrcRing - 1))) : ld64u(g + q); }
    DEVM uint32_t q16(uint32_t q) const { return ld16u(ring + (q & (kHc12SrcRing - 1))); }
    DEVM uint32_t q32(uint32_t q) const { return ld32u(ring + (q & (kHc12SrcRing - 1))); }
};

struct Hc12Tabs { const uint8_t* src; const uint16_t* chain; const uint32_t* rank; const uint32_t* list; };   // global memory
// eight consecutive list entries / sixteen consecutive chain links as two 16-byte loads (one request per lane each, not eight)
DEV void hc12_ld8(const uint32_t* p, uint32_t (&e)[8])
{
    const v16u_t a = *(const v16u_t*)p, b = *(const v16u_t*)(p + 4);
    e[0] = a.w[0]; e[1] = a.w[1]; e[2] = a.w[2]; e[3] = a.w[3]; e[4] = b.w[0]; e[5] = b.w[1]; e[6] = b.w[2]; e[7] = b.w[3];
}

template <class Src>
struct Hc12Walk {
    uint64_t ip0, ip1;                      // [ip, ip + 16)
    uint32_t pos, ipIndex, lowest, mi, chainPos, pattern, ip16, srcPatLen, dn0;
    uint32_t b[8];                          // the entries below the current candidate in its chain's list, nearest first
    uint32_t pe[8];                         // the eight entries below those, requested ahead (pePending)
    int nb, listEnded, pePending;           // entries in b; b ends with the chain's first position: nothing more to fetch
    uint32_t cursor;                        // list index of the entry right above the next fetch
    int lim, longest, offset, attempts, repeat, cnt, phase;

    DEVM void init(const Hc12Tabs& t, const Src& sw, int n, int p, uint32_t head)
    {
        pos = (uint32_t)p; lim = (n - kLastLiterals) - p;    // p + 32 <= n: the caller leaves the block's last positions to Hc12Lane
        ip0 = sw.r64(pos); ip1 = sw.r64(pos + 8);
        ipIndex = pos + kHcBase;
        lowest = (kHcBase + 65536u > ipIndex) ? kHcBase : ipIndex - 65535u;
        mi = head ? ipIndex - head : 0u;
        chainPos = 0; pattern = (uint32_t)ip0; longest = kMinMatch - 1; offset = 0;
        attempts = kHc12Searches; repeat = 0; srcPatLen = 0; cnt = 0; dn0 = 0;
        ip16 = hc12_bytes16(ip0, ip1, longest - 1);
        nb = 0; listEnded = 1; pePending = 0; cursor = 0; after_rank_down = 0;
        phase = kPhFilter;
        if (!head) return;                               // no candidate: the filter phase ends the search
        cursor = pos; after_rank_down = 1;               // the first candidate is the entry below pos in pos's own chain
        rank_trip(t);
    }
    DEVM Hc12F result() const
    {
        Hc12F f; f.len = longest > kMinMatch - 1 ? longest : 0; f.off = f.len ? offset : 0; return f;
    }
    DEVM void pop(int k)                     // drop the k nearest entries
    {
        if (k == 4) { b[0] = b[4]; b[1] = b[5]; b[2] = b[6]; b[3] = b[7]; }
        else for (int i = 0; i < k; ++i) { for (int j = 0; j < 7; ++j) b[j] = b[j + 1]; }
        nb -= k;
    }
    // e[0..8): the entries list[lo .. lo + 8), i.e. the eight below `cursor`; nearest first into b.  An entry that is the first of
    // its hash ends the chain; what lies below it in the list belongs to another hash.
    DEVM void take(const uint32_t (&e)[8])
    {
        uint32_t firsts = 0;                              // bit j: e[j] is the first position of its hash
        for (int j = 0; j < 8; ++j) firsts |= (e[j] >> 31) << j;
        listEnded = firsts != 0;
        nb = firsts ? 8 - (31 - __builtin_clz(firsts)) : 8;   // the entries down to the highest such one
        for (int k = 0; k < 8; ++k) b[k] = e[7 - k] & ~kHc12First;
        cursor -= 8u;
    }
    // kPhRank: cursor holds a POSITION of the chain the walk goes on in; the current candidate is that position or
    // (after_rank_down) the entry below it.  Its place in the list, whether the chain ends with it, and the entries below it.
    // (`list` has 8 readable entries in front of index 0.)
    int after_rank_down;
    DEVM void rank_trip(const Hc12Tabs& t)
    {
        cursor = t.rank[cursor] - (uint32_t)after_rank_down;
        after_rank_down = 0; pePending = 0;
        const int lo = (int)cursor - 8;
        uint32_t e[8];
        hc12_ld8(t.list + lo, e);
        phase = kPhFilter;
        if (t.list[(int)cursor] & kHc12First) { nb = 0; listEnded = 1; return; }
        take(e);
    }
    // successor of the current candidate in the followed chain -> dn0 as liblz4's table would give it (:988, :1065)
    DEVM uint32_t link_of(uint32_t m, uint32_t succ, bool has) const
    {
        const uint32_t q2 = m - kHcBase + chainPos;
        const uint32_t d = has ? q2 - succ : 65535u;
        return d < 65535u ? d : 65535u;
    }
    DEVM void advance()                                                                     // :1065
    {
        mi -= dn0;
        if (nb > 0) pop(1); else listEnded = 1;
        phase = kPhFilter;
    }
    DEVM void after() { if (dn0 == 1 && chainPos == 0) phase = kPhPattern; else advance(); }
    DEVM void conclude(const Src& sw, int mlen)                                              // :934-939, then what comes next
    {
        if (mlen > longest) {
            longest = mlen; offset = (int)(ipIndex - mi);
            ip16 = longest - 1 <= 14 ? hc12_bytes16(ip0, ip1, longest - 1) : sw.r16(pos + (uint32_t)longest - 1u);
        }
        if (mlen == longest && mi + (uint32_t)longest <= ipIndex) { phase = kPhScan; return; }
        after();
    }
    // Up to four candidates at once.  kNear: every byte looked at lies inside the source window (longest <= 60): plain LDS
    // reads, all eight issued together.  Returns true when the search has ended.
    template <bool kNear>
    DEVM bool filter_trip(const Hc12Tabs& t, const Src& sw)
    {
        if (nb == 0 && !listEnded) {                      // the entries requested a trip ago (or right now, if none were)
            if (!pePending) hc12_ld8(t.list + ((int)cursor - 8), pe);
            pePending = 0;
            take(pe);
        }
        if (!listEnded && !pePending) {                   // ask for the next eight as soon as the last request has been taken
            hc12_ld8(t.list + ((int)cursor - 8), pe);
            pePending = 1;
        }
        // Two rounds of up to four candidates per trip (a fetch brings eight entries): the second round costs the lanes that are
        // still walking only the candidate tests themselves, not another trip through the wave's phase scheduling.
        for (int round = 0; round < 2; ++round) {
        // the next candidates and their links, as far as the fetched entries go: candidate k needs entry k
        const int can = (listEnded || nb >= 4) ? 4 : nb;
        uint32_t m[5], d[4], t16[4], m32[4];
        m[0] = mi;
        for (int k = 0; k < 4; ++k) { d[k] = link_of(m[k], k < nb ? b[k] : 0u, k < nb); m[k + 1] = m[k] - d[k]; }
        const uint32_t o = (uint32_t)longest - 1u;
        for (int k = 0; k < 4; ++k) {
            const bool live = k < can && m[k] >= lowest && m[k] < ipIndex;   // (a dead one is never looked at: the search ends before)
            const uint32_t q = live ? m[k] - kHcBase : pos;
            if (kNear) { t16[k] = sw.q16(q + o); m32[k] = sw.q32(q); }
            else       { t16[k] = sw.r16(q + o); m32[k] = sw.r32(q); }
        }
        // what happens at candidate k, in the order of :918-:1065: 1 the walk is over; 2 it passes the tests -> count;
        // 3 its link is 1 -> pattern analysis; 0 it is behind us.  The first k with an event decides.
        int ev = 0, at = can;
        for (int k = 3; k >= 0; --k) {
            const int dead = !(m[k] >= lowest && attempts > k);
            const int pass = t16[k] == ip16 && m32[k] == pattern;
            const int pat  = d[k] == 1u && chainPos == 0u;
            const int e = dead ? 1 : (pass ? 2 : (pat ? 3 : 0));
            if (k < can && e) { ev = e; at = k; }
        }
        if (ev == 1) { attempts -= at; phase = kPhIdle; return true; }
        if (ev) {
            attempts -= at + 1;
            mi = at == 0 ? m[0] : (at == 1 ? m[1] : (at == 2 ? m[2] : m[3]));
            dn0 = at == 0 ? d[0] : (at == 1 ? d[1] : (at == 2 ? d[2] : d[3]));
            pop(at);
            if (ev == 3) { phase = kPhPattern; return false; }
            // the first 16 bytes of the count right here (most matches end inside them); longer ones go on in kPhCount
            const uint32_t mq = mi - kHcBase;
            uint64_t c0, c1;
            if (kNear) { c0 = (uint64_t)sw.q32(mq + 4) | ((uint64_t)sw.q32(mq + 8) << 32); c1 = sw.q32(mq + 12); }
            else       { c0 = (uint64_t)sw.r32(mq + 4) | ((uint64_t)sw.r32(mq + 8) << 32); c1 = sw.r32(mq + 12); }
            const uint64_t x0 = c0 ^ ((ip0 >> 32) | (ip1 << 32));
            const uint32_t x1 = (uint32_t)c1 ^ (uint32_t)(ip1 >> 32);
            const int c = x0 ? 4 + (ctz64(x0) >> 3) : (x1 ? 12 + (__builtin_ctz(x1) >> 3) : 16);
            if (c == 16 && lim > 16) { cnt = 16; phase = kPhCount; return false; }
            phase = kPhFilter;
            conclude(sw, c < lim ? c : lim);
            return false;
        }
        // all of them are behind us
        attempts -= can;
        mi = can == 1 ? m[1] : (can == 2 ? m[2] : (can == 3 ? m[3] : m[4]));
        if (nb >= can) pop(can); else { nb = 0; listEnded = 1; }
        if (nb == 0 && !listEnded) break;                 // the next entries are on their way
        }
        return false;
    }
    DEVM bool is_near() const { return longest <= 60; }
    DEVM void count_trip(const Hc12Tabs& t, const Src& sw)
    {
        const uint32_t mpos = mi - kHcBase;
        if (cnt + 16 > lim) {                            // the last bytes before the limit: the plain count
            conclude(sw, cnt + hc_count(t.src + pos + cnt, t.src + mpos + cnt, t.src + pos + lim));
            return;
        }
        const uint64_t x0 = sw.r64(pos + (uint32_t)cnt) ^ sw.r64(mpos + (uint32_t)cnt);
        const uint64_t x1 = sw.r64(pos + (uint32_t)cnt + 8u) ^ sw.r64(mpos + (uint32_t)cnt + 8u);
        if (x0) { conclude(sw, cnt + (ctz64(x0) >> 3)); return; }
        if (x1) { conclude(sw, cnt + 8 + (ctz64(x1) >> 3)); return; }
        cnt += 16;
        if (cnt >= lim) conclude(sw, lim);
    }
    DEVM bool scan_trip(const Hc12Tabs& t)                                                   // :964-987
    {
        const uint32_t mpos = mi - kHcBase;
        const int end = longest - kMinMatch + 1;
        uint32_t distNext = 1;
        if (end <= 16) {                                 // the step stays 1: every link of the match, then first-maximum
            uint32_t w[8];                               // (chain[] is padded: 16 entries from any position of the block are readable)
            hc12_ld8((const uint32_t*)(t.chain + mpos), w);
            for (int j = 0; j < 16; ++j) {
                const uint32_t e = j < end ? hc12_raw((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) : 0u;
                if (e > distNext) { distNext = e; chainPos = (uint32_t)j; }
            }
        } else {
            int stepw = 1, accel = 1 << 4;
            for (int p2 = 0; p2 < end; p2 += stepw) {
                const uint32_t cd = hc12_raw(t.chain[mpos + (uint32_t)p2]);
                stepw = (accel++ >> 4);
                if (cd > distNext) { distNext = cd; chainPos = (uint32_t)p2; accel = 1 << 4; }
            }
        }
        if (distNext > 1) {
            if (distNext > mi) { phase = kPhIdle; return true; }
            const uint32_t q2 = mpos + chainPos;         // the walk goes on in this position's chain
            mi -= distNext;
            nb = 0;
            if (distNext >= 65535u) { listEnded = 1; pePending = 0; phase = kPhFilter; }   // (a saturated link: the next test ends the search)
            else { listEnded = 0; pePending = 0; cursor = q2; phase = kPhRank; after_rank_down = 1; }
            return false;
        }
        after();
        return false;
    }
    DEVM bool pattern_trip(const Hc12Tabs& t, const Src& sw)                                 // :989-1062, dn0 == 1 && chainPos == 0
    {
        const uint8_t* const ip = t.src + pos;
        const uint8_t* const iHigh = ip + lim;
        const uint32_t mci = mi - 1;
        if (repeat == 0) {
            if (((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24))) {
                repeat = 2;
                srcPatLen = hc_count_pattern(ip + 4, iHigh, pattern) + 4;
            } else repeat = 1;
        }
        if (repeat == 2 && mci >= lowest) {
            const uint8_t* const mq = t.src + (mci - kHcBase);
            if (ld32u(mq) == pattern) {
                const uint32_t fwd = hc_count_pattern(mq + 4, iHigh, pattern) + 4;
                uint32_t back = hc_rcount_pattern(mq, t.src, pattern);
                {   const uint32_t far = mci - back;
                    back = mci - (far > lowest ? far : lowest); }
                const uint32_t seg = back + fwd;
                nb = 0; listEnded = 0; after_rank_down = 0; pePending = 0;
                if (seg >= srcPatLen && fwd <= srcPatLen) {
                    mi = mci + fwd - srcPatLen;                                        // this candidate is looked at next
                    cursor = mi - kHcBase; phase = kPhRank;
                } else {
                    mi = mci - back;
                    const uint32_t maxML = seg < srcPatLen ? seg : srcPatLen;
                    if ((uint32_t)longest < maxML) {
                        if (ipIndex - mi > 65535u) { phase = kPhIdle; return true; }
                        longest = (int)maxML; offset = (int)(ipIndex - mi);
                        ip16 = longest - 1 <= 14 ? hc12_bytes16(ip0, ip1, longest - 1) : sw.r16(pos + (uint32_t)longest - 1u);
                    }
                    const uint32_t at = mi - kHcBase;
                    const uint32_t dp = hc12_raw(t.chain[at]);
                    if (dp > mi) { phase = kPhIdle; return true; }
                    mi -= dp;
                    if (dp >= 65535u) { listEnded = 1; phase = kPhFilter; }
                    else { cursor = at; after_rank_down = 1; phase = kPhRank; }
                }
                return false;
            }
        }
        advance();
        return false;
    }
};

// ------------------------------------------------------------------------------------------ phase 3: the parser
// The price table (LZ4HC_optimal_t opt[LZ4_OPT_NUM + TRAILING_LITERALS], :1770-1775, :1836) as three arrays; entries below
// `nl` live in LDS, the rest (windows that long are rare) in a per-wave global workspace.  mloff = mlen << 16 | off
// (mlen <= 4095 inside the table, :1948-1956).
struct __attribute__((aligned(16))) Hc12Ent { int price; int litlen; uint32_t mloff; uint32_t pad; };   // one 16-byte LDS access per entry
struct Hc12Ws {
    Hc12Ent* ent; int nl;                                      // LDS
    int* gprice; int* glitlen; uint32_t* gmloff;               // global, index - nl
    uint64_t* seq;                                             // LDS: 64 pending sequences, pos | ml << 23 | off << 46
};
enum : int { kHc12WsGlobalBytes = kHc12OptEntries * 12 };

#define HC12_PRICE(i)  ((i) < w.nl ? w.ent[(i)].price  : w.gprice[(i) - w.nl])
#define HC12_LITLEN(i) ((i) < w.nl ? w.ent[(i)].litlen : w.glitlen[(i) - w.nl])
#define HC12_MLOFF(i)  ((i) < w.nl ? w.ent[(i)].mloff  : w.gmloff[(i) - w.nl])
// K (a compile-time flag): every index of this step lies below w.nl, i.e. in LDS
#define HC12K_PRICE(i)  (K ? w.ent[(i)].price  : HC12_PRICE(i))
#define HC12K_LITLEN(i) (K ? w.ent[(i)].litlen : HC12_LITLEN(i))
#define HC12K_MLOFF(i)  (K ? w.ent[(i)].mloff  : HC12_MLOFF(i))
#define HC12K_SET(i, pr, ll, mo) do { if (K) { Hc12Ent e_; e_.price = (pr); e_.litlen = (ll); e_.mloff = (mo); e_.pad = 0; w.ent[(i)] = e_; } else HC12_SET(i, pr, ll, mo); } while (0)
struct Hc12InLds { static constexpr bool value = true; };
struct Hc12Anywhere { static constexpr bool value = false; };
#define HC12_SET(i, pr, ll, mo) do { const int i_ = (i); \
        if (i_ < w.nl) { Hc12Ent e_; e_.price = (pr); e_.litlen = (ll); e_.mloff = (mo); e_.pad = 0; w.ent[i_] = e_; } \
        else { w.gprice[i_ - w.nl] = (pr); w.glitlen[i_ - w.nl] = (ll); w.gmloff[i_ - w.nl] = (mo); } } while (0)

// What a walk over part of a block leaves behind (the segments of lz4hc_lazy_device.inl)
struct LzRun {
    int cnt;        // records written
    int endIp;      // where it stopped, at the top of the parser's loop (>= the stop position), unless ...
    int finished;   // ... it reached the block's end: anchor = where the last literals start
    int anchor;
};
struct Hc12NoHook { DEVM bool operator()(int, int, int) const { return false; } };

// == LZ4HC_compress_optimal(nbSearches 16384, sufficient_len 4095, fullUpdate) (:1823-2123) over the search results
// F[0 .. n-12] (chain: for positions F does not cover).
//   kRec = false: the whole block, written to dst with the last literals; returns the compressed size or 0 (limitedOutput).
//   kRec = true : a walk from the top of the parser's loop (:1863) at (ipStart, anchorStart) until that point is reached again at
//                 or behind ipStop, its sequences as records in seqOut (lz4_seq_device.inl's format); hook(q, anchor, records)
//                 is called when a window's first match has been found at q and may end the walk there; *run = how it ended.
//                 The state at the top of the loop is (ip, anchor): the literal run in front of a window enters its prices.
#if defined(PLZ4_EMU)
static uint8_t* plz4_emu_f_need = nullptr;       // test diagnostics: the positions whose search result the parser actually used
#define EMU_F_NEED(p) do { if (plz4_emu_f_need) plz4_emu_f_need[(p)] = 1; } while (0)
#else
#define EMU_F_NEED(p) do {} while (0)
#endif
template <bool kRec, class Hook>
DEV int hc12_walk(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap,
                  const Hc12F* __restrict__ F, const uint16_t* __restrict__ chain, const Hc12Ws& w,
                  uint64_t* seqOut, const int ipStart, const int anchorStart, const int ipStop, Hook& hook, LzRun* run)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;                                       // :1388
    const bool limited = cap < compress_bound(n);                                          // :1505-1508
    const int mflimit = n - kMfLimit;
    int ip = ipStart, anchor = anchorStart, op = 0, nseq = 0, outN = 0;
    bool finished = true;
    STAT_DECL;                              // diagnostics build only: cycles per part of the parser (scripts/stats_probe12.py)
    const unsigned long long tP0 = STAT_NOW(); (void)tP0;

    // F of 64 consecutive positions, one per lane (the next window is requested ahead)
    LV(int, fLen); LV(int, fOff); LV(int, gLen); LV(int, gOff);
    int fBase = -(1 << 30);
    LANES({ fLen[I_] = 0; fOff[I_] = 0; gLen[I_] = 0; gOff[I_] = 0; })
    auto f_fetch = [&](LVREF(int, l), LVREF(int, o), int base) {
        LANES({ const int p = base + LANE; Hc12F f; f.len = 0; f.off = 0; if (p <= mflimit) f = F[p]; l[I_] = f.len; o[I_] = f.off; })
    };
    auto f_window = [&](int pos) {                     // make [fBase, fBase + 64) contain pos
        if (pos >= fBase && pos < fBase + 64) return;
        if (pos >= fBase + 64 && pos < fBase + 128) { LANES({ fLen[I_] = gLen[I_]; fOff[I_] = gOff[I_]; }) fBase += 64; }
        else { fBase = pos; f_fetch(fLen, fOff, fBase); }
        f_fetch(gLen, gOff, fBase + 64);
    };
    auto search_now = [&](int p) -> Hc12F {            // a position the search phase left out
        Hc12Flat ch; ch.c = chain;
        Hc12Lane<Hc12Flat> L; L.init(src, n, p, ch((uint32_t)p));
        while (!L.step(ch)) {}
        return L.result();
    };

    // ---- the writer: LZ4HC_encodeSequence (:268-354) for up to 64 recorded sequences at once, one per lane
    auto flush = [&]() -> bool {
        if (!nseq) return true;
        const unsigned long long tf0 = STAT_NOW(); (void)tf0;
        LDS_ORDER();
        const int cnt = nseq, anchor0 = anchor, op0 = op;
        nseq = 0;
        if (kRec) {                                    // records: position, length - 4, offset (22 | 22 | 16 bits)
            const int base = outN;
            LANES({
                if (LANE < cnt) {
                    const uint64_t e = w.seq[LANE];
                    seqOut[base + LANE] = (e & 0x3FFFFFu) | ((((e >> 23) & 0x7FFFFFu) - kMinMatch) << 22) | ((e >> 46) << 44);
                }
            })
            outN += cnt;
            const uint64_t le = UNI(w.seq[cnt - 1]);
            anchor = (int)(le & 0x7FFFFFu) + (int)((le >> 23) & 0x7FFFFFu);
            return true;
        }
        LV(int, sp); LV(int, sm); LV(int, so); LV(int, an); LV(int, lit); LV(int, xl); LV(int, xm); LV(int, acc); LV(int, sz);
        LANES({
            sp[I_] = 0; sm[I_] = 0; so[I_] = 0; an[I_] = 0; lit[I_] = 0; xl[I_] = 0; xm[I_] = 0; sz[I_] = 0;
            if (LANE < cnt) {
                const uint64_t e = w.seq[LANE];
                sp[I_] = (int)(e & 0x7FFFFFu); sm[I_] = (int)((e >> 23) & 0x7FFFFFu); so[I_] = (int)(e >> 46);
                if (LANE) { const uint64_t pe = w.seq[LANE - 1]; an[I_] = (int)(pe & 0x7FFFFFu) + (int)((pe >> 23) & 0x7FFFFFu); }
                else an[I_] = anchor0;
                lit[I_] = sp[I_] - an[I_];
                const int r = sm[I_] - kMinMatch;
                xl[I_] = lit[I_] >= 15 ? (lit[I_] - 15) / 255 + 1 : 0;
                xm[I_] = r >= 15 ? (r - 15) / 255 + 1 : 0;
                sz[I_] = 1 + xl[I_] + lit[I_] + 2 + xm[I_];
            }
            acc[I_] = sz[I_];
        })
        SCAN_INCL(acc);
        const int opEnd = op0 + RL(acc, 63);
        if (limited) {                                 // :303-305, :328-332, per sequence
            const uint64_t over = BALLOT(LANE < cnt && (
                (int64_t)(op0 + acc[I_] - sz[I_]) + 1 + lit[I_] / 255 + lit[I_] + (2 + 1 + kLastLiterals) > cap ||
                (int64_t)(op0 + acc[I_] - sz[I_]) + 1 + xl[I_] + lit[I_] + 2 + (sm[I_] - kMinMatch) / 255 + (1 + kLastLiterals) > cap));
            if (over) return false;
        }
        // sequences with long literal runs or long length codes are written by the whole wave, one after the other
        const uint64_t big = BALLOT(LANE < cnt && (lit[I_] > 64 || xm[I_] > 8));
        LANES({
            if (LANE < cnt && !((big >> LANE) & 1)) {
                int o = op0 + acc[I_] - sz[I_];
                const int r = sm[I_] - kMinMatch;
                dst[o++] = (uint8_t)((min_(lit[I_], 15) << 4) | min_(r, 15));
                if (xl[I_]) { int rest = lit[I_] - 15; for (; rest >= 255; rest -= 255) dst[o++] = 255; dst[o++] = (uint8_t)rest; }
                {   const uint8_t* s = src + an[I_]; uint8_t* d = dst + o; int rem = lit[I_];
                    for (; rem >= 16; rem -= 16) { *(v16u_t*)d = *(const v16u_t*)s; d += 16; s += 16; }
                    if (rem & 8) { st64u(d, ld64u(s)); d += 8; s += 8; }
                    if (rem & 4) { st32u(d, ld32u(s)); d += 4; s += 4; }
                    if (rem & 2) { st16u(d, ld16u(s)); d += 2; s += 2; }
                    if (rem & 1) { *d = *s; } }
                o += lit[I_];
                st16u(dst + o, (uint16_t)so[I_]); o += 2;
                if (xm[I_]) { int rest = r - 15; for (; rest >= 255; rest -= 255) dst[o++] = 255; dst[o++] = (uint8_t)rest; }
            }
        })
        for (uint64_t b = big; b; b &= b - 1) {
            const int k = ctz64(b);
            int o = op0 + RL(acc, k) - RL(sz, k);
            const int l = RL(lit, k), r = RL(sm, k) - kMinMatch, a = RL(an, k), off = RL(so, k);
            LANES({ if (LANE == 0) dst[o] = (uint8_t)((min_(l, 15) << 4) | min_(r, 15)); })
            o++;
            if (l >= 15) o = emit_len_ext(dst, o, l - 15);
            wave_copy(dst + o, src + a, l);
            o += l;
            LANES({ if (LANE == 0) st16u(dst + o, (uint16_t)off); })
            o += 2;
            if (r >= 15) o = emit_len_ext(dst, o, r - 15);
        }
        anchor = RL(sp, cnt - 1) + RL(sm, cnt - 1);
        op = opEnd;
        PSTAT(5, STAT_NOW() - tf0); PSTAT(9, cnt);
        return true;
    };
    auto push_seq = [&](int pos, int ml, int off) -> bool {
        const uint64_t e = (uint64_t)(uint32_t)pos | ((uint64_t)(uint32_t)ml << 23) | ((uint64_t)(uint32_t)off << 46);
        const int k = nseq;
        LANES({ if (LANE == 0) w.seq[k] = e; })
        nseq++;
        return nseq < 64 ? true : flush();
    };
    // the position parsing has reached but not yet written up to (== `anchor` once the pending sequences are out)
    int pendEnd = anchorStart;

    while (ip <= mflimit) {                                                                // :1863
        if (kRec && ip >= ipStop) { finished = false; break; }
        // ---- the next position with a match: F.len != 0 (positions without one only move ip, :1868)
        const unsigned long long tA = STAT_NOW(); (void)tA;
        f_window(ip);
        uint64_t any;
        {
            const int ipL = ip, fb = fBase;
            any = BALLOT(fb + LANE >= ipL && fb + LANE <= mflimit && fLen[I_] != 0);
        }
#if defined(PLZ4_EMU)
        { const int to = any ? fBase + ctz64(any) : min_(fBase + 63, mflimit); for (int q_ = ip; q_ <= to; ++q_) EMU_F_NEED(q_); }
#endif
        if (!any) { ip = fBase + 64; continue; }
        ip = fBase + ctz64(any);
        Hc12F first; first.len = RL(fLen, ip - fBase); first.off = RL(fOff, ip - fBase);
        if (first.len == kHc12NotComputed) first = search_now(ip);
        if (first.len == 0) { ip++; continue; }
        if (kRec && hook(ip, pendEnd, outN + nseq)) { finished = false; break; }
        const int llen = ip - pendEnd;
        PSTAT(0, STAT_NOW() - tA);
        const unsigned long long tB = STAT_NOW(); (void)tB;
        if (first.len > kHc12Sufficient) {                                                 // :1871-1882
            if (!push_seq(ip, first.len, first.off)) return 0;
            ip += first.len; pendEnd = ip;
            continue;
        }
        // ---- initialise the table from the first match (:1885-1919)
        {
            const int fl = first.len, fo = first.off;
            const int pm = hc_seq_price(llen, fl);
            for (int i0 = 0; i0 <= fl + kHcTrailing; i0 += 64) {
                LANES({
                    const int i = i0 + LANE;
                    if (i < kMinMatch)   HC12_SET(i, hc_lit_price(llen + i), llen + i, 1u << 16);
                    else if (i <= fl)    HC12_SET(i, hc_seq_price(llen, i), llen, ((uint32_t)i << 16) | (uint32_t)fo);
                    else if (i <= fl + kHcTrailing) HC12_SET(i, pm + hc_lit_price(i - fl), i - fl, 1u << 16);
                })
            }
        }
        LDS_ORDER();
        PSTAT(1, STAT_NOW() - tB); PSTAT(7, 1);
        int last = first.len, cur = 1;
        int bestMl = 0, bestOff = 0; bool direct = false;
        // ---- the DP (:1922-2019)
        for (;;) {
            if (cur >= last || ip + cur > mflimit) break;
            const unsigned long long tC = STAT_NOW(); (void)tC;
            f_window(ip + cur);
            LV(int, p0); LV(int, ll0); LV(uint32_t, mo0);
            uint64_t need;
            auto scan = [&](auto tag) {
                constexpr bool K = decltype(tag)::value;
                const int ipL = ip, fb = fBase, curL = cur, lastL = last;
                LANES({
                    const int pos = fb + LANE, c = pos - ipL;
                    p0[I_] = 0; ll0[I_] = 0; mo0[I_] = 0;
                    int want = 0;
                    if (c >= curL && c < lastL && pos <= mflimit && fLen[I_] != 0) {
                        if (K) { const Hc12Ent e = w.ent[c]; p0[I_] = e.price; ll0[I_] = e.litlen; mo0[I_] = e.mloff; }
                        else { p0[I_] = HC12_PRICE(c); ll0[I_] = HC12_LITLEN(c); mo0[I_] = HC12_MLOFF(c); }
                        const int p1 = HC12K_PRICE(c + 1), p4 = HC12K_PRICE(c + kMinMatch);
                        want = !(p1 <= p0[I_] && p4 < p0[I_] + 3);                          // fullUpdate skip test, :1929-1931
                    }
                    ll0[I_] = (int)((uint32_t)ll0[I_] | ((uint32_t)want << 31));
                })
                need = BALLOT(ll0[I_] < 0);
            };
            if (last + 8 < w.nl) scan(Hc12InLds{}); else scan(Hc12Anywhere{});
            PSTAT(2, STAT_NOW() - tC); PSTAT(10, 1);
            const unsigned long long tD = STAT_NOW(); (void)tD;
            if (!need) { cur = fBase + 64 - ip; continue; }
            const int s = ctz64(need);
            const int c = fBase + s - ip;
            EMU_F_NEED(fBase + s);
            Hc12F nm; nm.len = RL(fLen, s); nm.off = RL(fOff, s);
            if (nm.len == kHc12NotComputed) { nm = search_now(ip + c); if (!nm.len) { cur = c + 1; continue; } }
            if (nm.len > kHc12Sufficient || nm.len + c >= kHcOptNum) {                     // :1948-1956
                bestMl = nm.len; bestOff = nm.off; last = c + 1; cur = c; direct = true;
                break;
            }
            const int priceC = RL(p0, s), baseLit = RL(ll0, s) & 0x7FFFFFFF;
            const int mlenC = (int)(RL(mo0, s) >> 16);
            const int ll = (mlenC == 1) ? baseLit : 0;                                     // :1976-1982
            const int lastOld = last;
            PSTAT(12, STAT_NOW() - tD);
            auto update = [&](auto tag) {
                constexpr bool K = decltype(tag)::value;
                const int nmLen = nm.len; const uint32_t nmOff = (uint32_t)nm.off;
                const int litBase = priceC - hc_lit_price(baseLit);
                const bool needBase = (mlenC == 1) && (c > ll);
                // ONE round of LDS reads for the whole step: the prices the first 64 lengths compare with (one per lane), the
                // base price and the price at `last` (same address in every lane); everything after it is arithmetic and stores
                LV(int, old); LV(int, bpl); LV(int, newp); LV(int, took);
                {
                    const int bpAt = needBase ? c - ll : c;
                    LANES({
                        const int t = 1 + LANE, pos = c + t;
                        old[I_] = (t <= nmLen && pos <= lastOld + kHcTrailing) ? HC12K_PRICE(pos) : 0x7FFFFFFF;
                        bpl[I_] = LANE == 0 ? HC12K_PRICE(bpAt) : (LANE == 1 ? HC12K_PRICE(lastOld) : 0);
                    })
                }
                const int basePrice = (mlenC == 1) ? (needBase ? RL(bpl, 0) : 0) : priceC;
                const int plOld = RL(bpl, 1);
                PSTAT(13, STAT_NOW() - tD);
                // the part of price(ll, ml) that does not depend on the lane (below 274 the length code is one extra byte from 19 on:
                // no division per lane)
                const int seq0 = basePrice + 1 + 2 + hc_lit_price(ll);
                LANES({
                    const int t = 1 + LANE, pos = c + t;
                    took[I_] = 0; newp[I_] = old[I_];
                    if (t <= nmLen) {
                        if (t < kMinMatch) {                                               // literals after cur, :1958-1972
                            const int pr = litBase + hc_lit_price(baseLit + t);
                            if (pr < old[I_]) { newp[I_] = pr; HC12K_SET(pos, pr, baseLit + t, 1u << 16); }
                        } else {                                                           // every length of the match, :1984-2008
                            const int pr = seq0 + (t >= 19 ? 1 : 0);                       // == basePrice + hc_seq_price(ll, t), t <= 64
                            if (pr <= old[I_]) {                                           // (an entry past last + 3 reads as +inf: always taken)
                                took[I_] = 1; newp[I_] = pr;
                                HC12K_SET(pos, pr, ll, ((uint32_t)t << 16) | nmOff);
                            }
                        }
                    }
                })
                bool tookLast = nmLen <= 64 ? RL(took, (nmLen - 1) & 63) != 0 : false;
                PSTAT(14, STAT_NOW() - tD);
                for (int t0 = 65; t0 <= nmLen; t0 += 64) {                                 // lengths past 64: rare
                    LANES({
                        const int t = t0 + LANE;
                        took[I_] = 0;
                        if (t <= nmLen) {
                            const int pos = c + t;
                            const int pr = basePrice + hc_seq_price(ll, t);
                            if (pos > lastOld + kHcTrailing || pr <= HC12K_PRICE(pos)) {
                                took[I_] = 1;
                                HC12K_SET(pos, pr, ll, ((uint32_t)t << 16) | nmOff);
                            }
                        }
                    })
                    if (nmLen - t0 < 64) tookLast = RL(took, nmLen - t0) != 0;
                }
                int pl;                                                                    // the price at `last` after this step
                if (tookLast && lastOld < c + nmLen) { last = c + nmLen; pl = basePrice + hc_seq_price(ll, nmLen); }
                else if (lastOld > c && lastOld - c <= 64 && lastOld - c <= nmLen) pl = RL(newp, lastOld - c - 1);
                else if (lastOld > c && lastOld - c <= nmLen) { LDS_ORDER(); pl = UNI(HC12K_PRICE(lastOld)); }
                else pl = plOld;
                LDS_ORDER();
                {   const int lastL = last;                                                // :2011-2018
                    LANES({ if (LANE >= 1 && LANE <= kHcTrailing) HC12K_SET(lastL + LANE, pl + hc_lit_price(LANE), LANE, 1u << 16); })
                }
                LDS_ORDER();
            };
            if ((lastOld > c + nm.len ? lastOld : c + nm.len) + 8 < w.nl) update(Hc12InLds{}); else update(Hc12Anywhere{});
            PSTAT(3, STAT_NOW() - tD); PSTAT(6, 1);
            cur = c + 1;
        }
        const unsigned long long tE = STAT_NOW(); (void)tE;
        if (!direct) {                                                                     // :2022-2024
            const uint32_t mo = UNI(HC12_MLOFF(last));
            bestMl = (int)(mo >> 16); bestOff = (int)(mo & 0xFFFFu);
            cur = last - bestMl;
        }
        if (last <= 63 && last + 8 < w.nl) {
            // A window that fits the wave (about half of them): the table's (mlen, offset) column in one LDS read, one entry per
            // lane.  The reverse traversal (:2026-2046) is a walk over lanes (readlane per hop); it hands every node of the path
            // the old entry of the node above it -- here: each path lane takes the entry of the next path lane above it -- and
            // the forward pass (:2048-2064) visits exactly the path's nodes, so the sequences are the path lanes whose new
            // length is not 1, at position ip + lane, in lane order.
            LV(uint32_t, mo);
            { const int top = cur; LANES({ mo[I_] = (LANE <= top) ? w.ent[LANE].mloff : 0u; }) }
            uint64_t path = 0;
            for (int cand = cur;;) {
                path |= 1ull << cand;
                const int nml = (int)(RL(mo, cand) >> 16);
                if (nml > cand) break;
                cand -= nml;
            }
            LV(uint32_t, nv); LV(int, isSeq);
            {
                const uint64_t pathL = path; const int curL = cur; const bool directL = direct;
                const uint32_t best = ((uint32_t)bestMl << 16) | (uint32_t)(bestOff & 0xFFFF);
                LANES({
                    const uint64_t above = (LANE < 63) ? (pathL >> (LANE + 1)) : 0;
                    const int from = above ? LANE + 1 + ctz64(above) : LANE;
                    nv[I_] = SHFL(mo, from);
                    if (LANE == curL) nv[I_] = best;
                    const int ml = (directL && LANE == curL) ? 2 : (int)(nv[I_] >> 16);      // (a direct match is a match whatever its 16 bits say)
                    isSeq[I_] = ((pathL >> LANE) & 1) && ml != 1;
                })
            }
            const uint64_t seqMask = BALLOT(isSeq[I_]);
            const int cnt = __builtin_popcountll(seqMask);
            if (cnt) {
                if (nseq + cnt > 64) { if (!flush()) return 0; }
                const int base = nseq, ip0 = ip; const bool directL = direct; const int curL = cur, bestL = bestMl;
                LANES({
                    if (isSeq[I_]) {
                        const int k = base + __builtin_popcountll(seqMask & ((1ull << LANE) - 1));
                        const int ml = (directL && LANE == curL) ? bestL : (int)(nv[I_] >> 16);
                        w.seq[k] = (uint64_t)(uint32_t)(ip0 + LANE) | ((uint64_t)(uint32_t)ml << 23) | ((uint64_t)(nv[I_] & 0xFFFFu) << 46);
                    }
                })
                nseq += cnt;
                const int hi = 63 - __builtin_clzll(seqMask);
                const int mlHi = (direct && hi == cur) ? bestMl : (int)(RL(nv, hi) >> 16);
                pendEnd = ip + hi + mlHi;
                if (nseq == 64) { if (!flush()) return 0; }
            }
            ip += direct ? cur + bestMl : last;
        } else {
        {   // reverse traversal: mark the chosen path (:2026-2046)
                // (a direct long match may not fit the 16 bits of length: it is the path's last sequence, bestMl is kept aside)
                int cand = cur; uint32_t sel = ((uint32_t)bestMl << 16) | (uint32_t)(bestOff & 0xFFFF);
                for (;;) {
                    const uint32_t nx = UNI(HC12_MLOFF(cand));
                    { const int candL = cand; const uint32_t selL = sel; LANES({ if (LANE == 0) { if (candL < w.nl) w.ent[candL].mloff = selL; else w.gmloff[candL - w.nl] = selL; } }) }
                    LDS_ORDER();
                    sel = nx;
                    const int nml = (int)(nx >> 16);
                    if (nml > cand) break;
                    cand -= nml;
                }
            }
            {   // record the sequences in order (:2048-2064)
                int r = 0;
                while (r < last) {
                    const uint32_t mo = UNI(HC12_MLOFF(r));
                    int ml = (int)(mo >> 16); const int off = (int)(mo & 0xFFFFu);
                    if (direct && r == cur) ml = bestMl;                                    // may not fit the table's 16 bits
                    if (ml == 1) { ip++; r++; continue; }
                    r += ml;
                    if (!push_seq(ip, ml, off)) return 0;
                    ip += ml; pendEnd = ip;
                }
            }
        }
        PSTAT(4, STAT_NOW() - tE);
    }
    if (!flush()) return 0;
    if (kRec) {
        run->cnt = outN; run->endIp = ip; run->finished = finished ? 1 : 0; run->anchor = pendEnd;
        return outN;
    }
    // last literals (:2067-2098, limitedOutput / notLimited)
    {
        const int lastRun = n - anchor;
        const int llAdd = (lastRun + 255 - 15) / 255;
        if (limited && (int64_t)op + 1 + llAdd + lastRun > cap) return 0;
        if (lastRun >= 15) { LANES({ if (LANE == 0) dst[op] = 0xF0; }) op = emit_len_ext(dst, op + 1, lastRun - 15); }
        else { LANES({ if (LANE == 0) dst[op] = (uint8_t)(lastRun << 4); }) op++; }
        wave_copy(dst + op, src + anchor, lastRun);
        op += lastRun;
    }
    PSTAT(8, STAT_NOW() - tP0); PSTAT(11, 1);
    PSTAT_FLUSH();
    return op;
}
DEV int hc12_parse(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap,
                   const Hc12F* __restrict__ F, const uint16_t* __restrict__ chain, const Hc12Ws& w)
{
    Hc12NoHook none;
    return hc12_walk<false>(src, n, dst, cap, F, chain, w, nullptr, 0, 0, n + 1, none, nullptr);
}

}  // namespace plz4
