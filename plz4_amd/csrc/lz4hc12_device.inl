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
//       position with the same hash.  It is built for the whole block up front (hc12_build_chain), 64 positions per step.
//   (2) level 12 always searches with the same parameters (minLen 3, forward only, :1936), so the search result of a position
//       is a function of the data alone as well: F(p) = (length, offset).  It is computed for every position the parser may
//       ask for, one position per LANE (Hc12Lane: one chain step per call, so that a kernel can refill finished lanes from a
//       queue), ahead of the parser.
//   The price DP stays serial per block but touches no match finder any more (hc12_parse): it reads F, scans the skip test
//   (:1929-1934) for 64 positions at once, updates the prices of a match's lengths one length per lane, and hands the chosen
//   sequences to a writer that lays out 64 sequences at a time.
//
// Everything here is also compiled by g++ for tests/emu (PLZ4_EMU), where the three phases run back to back on the CPU and are
// checked against the real LZ4_compress_HC of oracle/_ref.
#pragma once
#include "wave.h"
#include "lz4_device.inl"
#include "lz4hc_device.inl"

namespace plz4 {

enum : int { kHc12Sufficient = 4095,           // level 12: sufficient_len 4096, capped to LZ4_OPT_NUM - 1 (lz4hc.c:92-106, :1860)
             kHc12Searches   = 16384,          // level 12: nbSearches
             kHc12NotComputed = -1,            // F.len of a position the search phase skipped: the parser searches it itself
             kHc12OptEntries = kHcOptNum + kHcTrailing + 1 };
struct Hc12F { int32_t len; int32_t off; };    // LZ4HC_FindLongerMatch's answer for one position (len 0: none)

// ------------------------------------------------------------------------------------------ phase 1: the chain
// chain[q] (uint16): distance from q to the previous position with the same LZ4HC hash (lz4hc.c:120-122), 1..65535;
// 0 = none within 65535.  liblz4's table holds min(distance, 65535) instead (:793-796); what differs is only that a head
// candidate exactly 65535 back (valid, :913-916 read the hash table's full index) can be told from one further back.
// hc12_raw gives liblz4's value back for the places that compare or subtract chain values.
DEV uint32_t hc12_raw(uint32_t d) { return d ? d : 65535u; }

// One wave, `tab` = 32768 x uint32 of LDS (128 KiB): last position + 1 per hash.  One atomic max per lane commits the lane's
// position and returns the previous holder; same-slot lanes are expected to resolve in ascending lane order, any other order
// shows up as a returned position >= the lane's own and is put right in place.  chain[0, nPad) is written (nPad % 64 == 0).
DEV void hc12_build_chain(const uint8_t* __restrict__ src, const int n, uint16_t* __restrict__ chain, const int nPad, uint32_t* tab)
{
    LANES({ for (int i = LANE; i < kHcHashEntries; i += 64) tab[i] = 0u; })
    LDS_FENCE();
    const int nIns = n >= 4 ? n - 3 : 0;                       // positions with 4 bytes to hash
    for (int base = 0; base < nPad; base += 64) {
        LV(uint32_t, h); LV(uint32_t, prev); LV(int, act);
        LANES({
            const int p = base + LANE;
            act[I_] = p < nIns; h[I_] = 0; prev[I_] = 0;
            if (act[I_]) {
                h[I_] = (ld32u(src + p) * 2654435761u) >> 17;
                prev[I_] = lds_max_rtn(&tab[h[I_]], (uint32_t)p + 1u);
            }
        })
        LDS_FENCE();
        if (BALLOT(act[I_] && prev[I_] > (uint32_t)(base + LANE))) {
            // some other resolution order: a slot's pre-batch value is the smallest value its lanes got back; every lane
            // takes the nearest lower lane of its slot, or that pre-batch value
            LV(uint32_t, gmin); LV(uint32_t, near);
            LANES({ gmin[I_] = prev[I_]; near[I_] = 0; })
            for (int l = 0; l < 64; ++l) {
                const uint32_t hl = RL(h, l), pl = RL(prev, l); const int al = RL(act, l);
                LANES({
                    if (al && act[I_] && h[I_] == hl) {
                        if (pl < gmin[I_]) gmin[I_] = pl;
                        if (l < LANE) near[I_] = (uint32_t)(base + l) + 1u;
                    }
                })
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
// A window of the chain in LDS: 72 chunks of 1024 entries, position q at chunk (q / 1024) % 72.  It holds [loaded - 73728, loaded).
enum : int { kHc12RingChunk = 1024, kHc12RingChunks = 72, kHc12RingEntries = kHc12RingChunk * kHc12RingChunks };
struct Hc12Ring {
    const uint16_t* r;
    DEVM uint32_t operator()(uint32_t q) const { return r[(((q >> 10) % (uint32_t)kHc12RingChunks) << 10) | (q & 1023u)]; }
};

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

// ------------------------------------------------------------------------------------------ phase 3: the parser
// The price table (LZ4HC_optimal_t opt[LZ4_OPT_NUM + TRAILING_LITERALS], :1770-1775, :1836) as three arrays; entries below
// `nl` live in LDS, the rest (windows that long are rare) in a per-wave global workspace.  mloff = mlen << 16 | off
// (mlen <= 4095 inside the table, :1948-1956).
struct Hc12Ws {
    int* price; int* litlen; uint32_t* mloff; int nl;          // LDS
    int* gprice; int* glitlen; uint32_t* gmloff;               // global, index - nl
    uint64_t* seq;                                             // LDS: 64 pending sequences, pos | ml << 23 | off << 46
};
enum : int { kHc12WsGlobalBytes = kHc12OptEntries * 12 };

#define HC12_PRICE(i)  ((i) < w.nl ? w.price[(i)]  : w.gprice[(i) - w.nl])
#define HC12_LITLEN(i) ((i) < w.nl ? w.litlen[(i)] : w.glitlen[(i) - w.nl])
#define HC12_MLOFF(i)  ((i) < w.nl ? w.mloff[(i)]  : w.gmloff[(i) - w.nl])
#define HC12_SET(i, pr, ll, mo) do { const int i_ = (i); \
        if (i_ < w.nl) { w.price[i_] = (pr); w.litlen[i_] = (ll); w.mloff[i_] = (mo); } \
        else { w.gprice[i_ - w.nl] = (pr); w.glitlen[i_ - w.nl] = (ll); w.gmloff[i_ - w.nl] = (mo); } } while (0)

// == LZ4HC_compress_optimal(nbSearches 16384, sufficient_len 4095, fullUpdate) + the last literals (:1823-2123) over the
// search results F[0 .. n-12] (chain: for positions F does not cover).  Returns the compressed size or 0 (limitedOutput).
DEV int hc12_parse(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap,
                   const Hc12F* __restrict__ F, const uint16_t* __restrict__ chain, const Hc12Ws& w)
{
    if ((uint32_t)n > (uint32_t)kMaxInput) return 0;                                       // :1388
    const bool limited = cap < compress_bound(n);                                          // :1505-1508
    const int mflimit = n - kMfLimit;
    int ip = 0, anchor = 0, op = 0, nseq = 0;

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
        LDS_FENCE();
        const int cnt = nseq, anchor0 = anchor, op0 = op;
        nseq = 0;
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
    int pendEnd = 0;

    while (ip <= mflimit) {                                                                // :1863
        // ---- the next position with a match: F.len != 0 (positions without one only move ip, :1868)
        f_window(ip);
        uint64_t any;
        {
            const int ipL = ip, fb = fBase;
            any = BALLOT(fb + LANE >= ipL && fb + LANE <= mflimit && fLen[I_] != 0);
        }
        if (!any) { ip = fBase + 64; continue; }
        ip = fBase + ctz64(any);
        Hc12F first; first.len = RL(fLen, ip - fBase); first.off = RL(fOff, ip - fBase);
        if (first.len == kHc12NotComputed) first = search_now(ip);
        if (first.len == 0) { ip++; continue; }
        const int llen = ip - pendEnd;
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
        LDS_FENCE();
        int last = first.len, cur = 1;
        int bestMl = 0, bestOff = 0; bool direct = false;
        // ---- the DP (:1922-2019)
        for (;;) {
            if (cur >= last || ip + cur > mflimit) break;
            f_window(ip + cur);
            LV(int, p0); LV(int, ll0); LV(uint32_t, mo0);
            uint64_t need;
            {
                const int ipL = ip, fb = fBase, curL = cur, lastL = last;
                LANES({
                    const int pos = fb + LANE, c = pos - ipL;
                    p0[I_] = 0; ll0[I_] = 0; mo0[I_] = 0;
                    int want = 0;
                    if (c >= curL && c < lastL && pos <= mflimit && fLen[I_] != 0) {
                        p0[I_] = HC12_PRICE(c); ll0[I_] = HC12_LITLEN(c); mo0[I_] = HC12_MLOFF(c);
                        const int p1 = HC12_PRICE(c + 1), p4 = HC12_PRICE(c + kMinMatch);
                        want = !(p1 <= p0[I_] && p4 < p0[I_] + 3);                          // fullUpdate skip test, :1929-1931
                    }
                    ll0[I_] = (int)((uint32_t)ll0[I_] | ((uint32_t)want << 31));
                })
                need = BALLOT(ll0[I_] < 0);
            }
            if (!need) { cur = fBase + 64 - ip; continue; }
            const int s = ctz64(need);
            const int c = fBase + s - ip;
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
            bool tookLast = false;
            {
                const int basePrice = (mlenC == 1) ? ((c > ll) ? UNI(HC12_PRICE(c - ll)) : 0) : priceC;
                const int nmLen = nm.len; const uint32_t nmOff = (uint32_t)nm.off;
                const int litBase = priceC - hc_lit_price(baseLit);
                for (int t0 = 1; t0 <= nmLen; t0 += 64) {
                    LV(int, took);
                    LANES({
                        const int t = t0 + LANE;
                        took[I_] = 0;
                        if (t <= nmLen) {
                            const int pos = c + t;
                            if (t < kMinMatch) {                                           // literals after cur, :1958-1972
                                const int pr = litBase + hc_lit_price(baseLit + t);
                                if (pr < HC12_PRICE(pos)) HC12_SET(pos, pr, baseLit + t, 1u << 16);
                            } else {                                                       // every length of the match, :1984-2008
                                const int pr = basePrice + hc_seq_price(ll, t);
                                if (pos > lastOld + kHcTrailing || pr <= HC12_PRICE(pos)) {
                                    took[I_] = 1;
                                    HC12_SET(pos, pr, ll, ((uint32_t)t << 16) | nmOff);
                                }
                            }
                        }
                    })
                    if (nmLen - t0 < 64) tookLast = RL(took, nmLen - t0) != 0;
                }
                if (tookLast && lastOld < c + nmLen) last = c + nmLen;
            }
            LDS_FENCE();
            {   const int pl = UNI(HC12_PRICE(last)); const int lastL = last;              // :2011-2018
                LANES({ if (LANE >= 1 && LANE <= kHcTrailing) HC12_SET(lastL + LANE, pl + hc_lit_price(LANE), LANE, 1u << 16); })
            }
            LDS_FENCE();
            cur = c + 1;
        }
        if (!direct) {                                                                     // :2022-2024
            const uint32_t mo = UNI(HC12_MLOFF(last));
            bestMl = (int)(mo >> 16); bestOff = (int)(mo & 0xFFFFu);
            cur = last - bestMl;
        }
        {   // reverse traversal: mark the chosen path (:2026-2046)
            // (a direct long match may not fit the 16 bits of length: it is the path's last sequence, bestMl is kept aside)
            int cand = cur; uint32_t sel = ((uint32_t)bestMl << 16) | (uint32_t)(bestOff & 0xFFFF);
            for (;;) {
                const uint32_t nx = UNI(HC12_MLOFF(cand));
                { const int candL = cand; const uint32_t selL = sel; LANES({ if (LANE == 0) { if (candL < w.nl) w.mloff[candL] = selL; else w.gmloff[candL - w.nl] = selL; } }) }
                LDS_FENCE();
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
    if (!flush()) return 0;
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
    return op;
}

}  // namespace plz4
