// lz4hc_lazy_device.inl -- HC level 2 (LZ4MID), levels 3..9 (the lazy hash-chain parser) and 10..11 (the optimal parser without
// the full update) for independent blocks, cut for a GPU:
// the waves that walk a block only DECIDE -- searches with all 64 lanes, one 8-byte record per sequence; the block's bytes are
// written by the data-parallel emit stage level 1 uses (lz4_seq_device.inl: one lane per sequence).
//
//   reference: /root/reference/internal/pkg/clz4/lz4hc.c
//     levels 3..9  LZ4HC_compress_hashChain :1121-1363 (nbSearches = 1 << (level-1), table :92-106; pattern analysis above 128 attempts)
//                  LZ4HC_InsertAndGetWiderMatch :884-1104
//     level 2      LZ4MID_compress :521-775 (hc_mid_parse, at the end of this file)
//     levels 10..11 LZ4HC_compress_optimal :1823-2123 (hc_opt_run)
//     output       LZ4HC_encodeSequence :268-354, last literals :1330-1362 / :745-772
//
// Levels 3..9 -- what the reference's loop does, and where each part went:
//   (1) Searches.  On an independent block every position below ip is in the chain when ip is searched (LZ4HC_Insert runs up to
//       ip, :914), so the chain is a function of the data; it is built up front together with the per-hash lists
//       (lz4hc12_device.inl: the candidates of a chain are consecutive list entries).  The searches are hc_find_few: one
//       candidate per lane, and for the FIRST search of a sequence (:1159 -- it runs at every position of a literal run until
//       one finds something) 64 / attempts consecutive positions at once: three memory round trips serve 16 positions at
//       level 3; levels 7..9 take one / two / four rounds of 64 candidates for one position.  Level 9's pattern analysis can only
//       step in at a candidate whose chain link is 1: a search that meets one is handed to hc_find_wider_lists (lz4hc_device.inl).
//       [Computing the first search of EVERY position ahead of the walk, one position per lane over the whole chip -- level
//       12's scheme -- was built and measured here as well and lost at every level: the positions inside matches are searched
//       for nothing.]
//   (2) The sequence itself is not written here.  A record (position, length, offset) per sequence; lz4_seq_device.inl's emit
//       kernels size, place and write them, 64 sequences per wave step, and hash the payload.  LZ4HC's output checks
//       (:283-288, :323-327, :1340-1352) are conservative estimates of bytes that the rest of the block needs in any case (a token,
//       the last five literals), so, as at level 1, "a check fails somewhere" == "the complete block is larger than the
//       capacity": seq_emit_scan's verdict is LZ4_compress_HC's return value.  (Check 1 at a sequence: op + ll + ll/255 + 8 >
//       oend, while the block needs at least op + 1 + ext(ll) + ll + 2 + ext(ml) + 6 with ext(ll) >= ll/255; check 2:
//       op' + (ml-4)/255 + 6 > oend with ext(ml) >= (ml-4)/255; the last one, :1340, is exact.)
//   (3) The decisions (:1177-1306: which of up to three overlapping matches goes first and how they are trimmed) are the
//       reference's, one for one -- they are the format of the output -- but the loop is a three-state machine over
//       (ip, m1, m2, m3) without output pointers, tables or inserts.
//   (4) And the walk is cut into SEGMENTS that are walked at the same time.  Whenever the loop is back at its top (:1157, "no
//       match in hand": the state kLzFirst below) everything it will do from there on is a function of ip and the data -- and
//       already of q, the first position at or behind ip whose first search is not empty.  So a wave may start in that state at
//       any position B_j: its records are the block's records from the first q on that the true walk -- the one from position
//       0 -- also reaches in that state.  hc_lazy_segment walks [B_j, B_j+1) for every j at once and notes the first
//       kLzStarts such q of each segment; hc_lazy_stitch then walks from where segment j-1 truly ended until it meets one
//       of segment j's q (typically after a handful of sequences: a literal run re-synchronises the two walks), and the
//       block's records are: segment 0, bridge 1, segment 1 from the meeting point on, bridge 2, ...  A bridge that meets
//       nothing within the noted starts simply walks on to B_j+1 and replaces the segment.  The pieces are gathered into one
//       array for the emit stage.  The serial work of one wave per block becomes S waves per block; the output is the one
//       walk's, bit for bit.
#pragma once
#include "lz4hc12_device.inl"
#include "lz4_seq_device.inl"

namespace plz4 {

enum : int { kLzFirst = 0, kLzSecond = 1, kLzThird = 2 };

// Sequence records of one block, 64 at a time (one coalesced store)
struct SeqSink {
    uint64_t* seq; int n;
    int bias = 0;                 // (a walk behind an external segment counts positions from the segment's first byte: records count from the block's)
    LV(uint64_t, buf);
    DEVM void put(int pos, int ml, int off)
    {
        const uint64_t r = seq_pack((uint32_t)(pos - bias), (uint32_t)(ml - kMinMatch), (uint32_t)off);
        const int slot = n & 63;
        LANES({ if (LANE == slot) buf[I_] = r; })
        ++n;
        if ((n & 63) == 0) flush64(n - 64, 64);
    }
    DEVM void flush64(int base, int cnt) { LANES({ if (LANE < cnt) seq[base + LANE] = buf[I_]; }) }
    DEVM void finish() { if (n & 63) flush64(n & ~63, n & 63); }
};

// LZ4HC_InsertAndGetWiderMatch without pattern analysis and chain swap for nb = 4..32 attempts, one candidate per lane.
//   multi = false: the search at `pos` with look-back down to `low` and `longest` to beat (lanes 0..nb-1).
//   multi = true : the first search (no look-back, nothing to beat) of the 64 / nb positions pos, pos+1, ... at once, lanes
//                  [g*nb, (g+1)*nb) for position pos + g; the answer is the first position that finds a match.
// A candidate's total is counted exactly; the reference's walk keeps the first candidate with the largest total above
// `longest` (:934-939: a later one replaces it only when longer) -- its 2-byte filter (:921) never rejects a candidate that
// would improve (the two bytes lie inside such a candidate's match because lookBack < longest at every call), so the answer
// is the first maximum.  Candidates: the entries below the position in its chain's list, while the chain has not ended, the
// distance is <= 65535 (:918; a saturated link, :1065, leads below that window as well) and attempts are left.
struct LzFound { int pos, len, off, back; };          // pos < 0: nothing
// one round of at most 64 candidates: those ciBase.. below the position in its list; *more: every lane had a candidate (the chain
// goes on below them).  len = the largest total of the round (0: none), first candidate first.
// *link1: a candidate of the round has a chain link of 1 (where level 9's pattern analysis would step in, :989).
// kD: an external segment of s.pfx bytes lies in front of the block and its positions (all but its last three) are in the lists:
// a candidate in the block is not extended backwards into the segment (lz4hc.c:933: LZ4HC_countBack down to prefixPtr), one in the
// segment down to the segment's first byte (:953); the segment's candidates have no 2-byte pre-check (:940-960), which changes
// nothing here: every candidate is counted exactly.
template <bool kD = false>
DEV LzFound hc_find_round(const HcState& s, const int pos, const int low, const int highLimit, const int nb,
                          const bool multi, const int mflimit, const int ciBase, bool* more, bool* link1)
{
    const uint8_t* const src = s.src;
    const uint8_t* const iHigh = src + highLimit;
    const int lookBack = pos - low;
    LV(int, p); LV(int, ci); LV(int, q); LV(uint32_t, key); LV(int, bk); LV(int, qn);
    LANES({
        const int g = multi ? LANE / nb : 0;
        ci[I_] = multi ? LANE % nb : LANE;
        p[I_] = pos + g;
        const bool act = multi ? p[I_] <= mflimit : LANE < nb;
        uint32_t head = 0, rk = 0;
        if (act) { head = s.w.pre[p[I_]]; rk = s.w.rank[p[I_]]; }
        const int at = (int)rk - 1 - ciBase - ci[I_];
        uint32_t e = 0x80000000u;
        if (act && head && at >= -8) e = s.w.list[at];
        qn[I_] = (link1 && act && head && at >= -7) ? (int)(s.w.list[at - 1] & 0x7FFFFFFFu) : -2;   // the entry below: this candidate's link
        q[I_] = (int)(e & 0x7FFFFFFFu);
        key[I_] = (act && head) ? (e >> 31) : 2u;             // for now: 1 = the chain's first position, 2 = no candidate here
    })
    const uint64_t firsts = BALLOT(key[I_] == 1u);
    LV(bool, ok);
    LANES({
        const int g0 = multi ? LANE - ci[I_] : 0;               // my group's first lane
        const uint64_t below = firsts & (((uint64_t)1 << LANE) - 1) & ~(((uint64_t)1 << g0) - 1);   // chain ended at a nearer candidate
        const bool valid = key[I_] != 2u && !below && p[I_] - q[I_] <= 65535 && q[I_] < p[I_];
        ok[I_] = valid && !((firsts >> LANE) & 1);                                                  // ... and the chain goes on below it
        qn[I_] = (ok[I_] && q[I_] - qn[I_] == 1) ? 1 : 0;
        int total = 0; bk[I_] = 0;
        if (valid) {
            const uint8_t* const ipp = src + p[I_];
            const uint8_t* const mp = src + q[I_];
            if (ld32u(mp) == ld32u(ipp)) {                                                          // :930
                const int back = lookBack ? hc_count_back(ipp, mp, src + low, (kD && q[I_] >= s.pfx) ? src + s.pfx : src) : 0;   // :933 / :953 (<= 0)
                total = kMinMatch + hc_count(ipp + kMinMatch, mp + kMinMatch, iHigh) - back;
                bk[I_] = back;
            }
        }
        key[I_] = ((uint32_t)total << 6) | (uint32_t)(63 - ci[I_]);                                  // largest total, nearest candidate first
    })
    for (int m = 1; m < nb; m <<= 1) {
        LV(uint32_t, o);
        LANES({ o[I_] = SHFL(key, LANE ^ m); })
        LANES({ key[I_] = key[I_] > o[I_] ? key[I_] : o[I_]; })
    }
    *more = BALLOT(ok[I_]) == ~0ull;
    if (link1) *link1 = BALLOT(qn[I_] == 1) != 0;
    LzFound f; f.pos = -1; f.len = 0; f.off = 0; f.back = 0;
    const uint64_t hit = BALLOT(ci[I_] == 0 && (key[I_] >> 6) != 0u);
    if (hit) {
        const int l0 = multi ? ctz64(hit) : 0;
        const uint32_t kk = RL(key, l0);
        const int c = l0 + 63 - (int)(kk & 63u);
        f.pos = RL(p, l0); f.len = (int)(kk >> 6); f.off = f.pos - RL(q, c); f.back = RL(bk, c);
    }
    return f;
}
// pa: level 9 (pattern analysis on).  Its walk is this one as long as no candidate it looks at has a link of 1; when one has,
// pos = -2 comes back and the caller asks hc_find_wider_lists, which plays the pattern analysis.
template <bool kD = false>
DEV LzFound hc_find_few(const HcState& s, const int pos, const int low, const int highLimit, const int longest, const int nb,
                        const bool multi, const int mflimit, const bool pa = false)
{
    bool more = false, l1 = false, any1 = false;
    LzFound best = hc_find_round<kD>(s, pos, low, highLimit, nb < 64 ? nb : 64, multi, mflimit, 0, &more, pa ? &l1 : nullptr);
    any1 = l1;
    for (int base = 64; base < nb && more && !any1; base += 64) {           // (128 / 256 attempts: more rounds, for the one position)
        const LzFound f = hc_find_round<kD>(s, pos, low, highLimit, nb - base < 64 ? nb - base : 64, false, mflimit, base, &more, pa ? &l1 : nullptr);
        any1 |= l1;
        if (f.len > best.len) best = f;                                     // a later candidate replaces only when longer (:934)
    }
    if (any1) { best.pos = -2; best.len = longest; best.off = 0; best.back = 0; return best; }
    if (best.len <= longest) { best.pos = -1; best.len = longest; best.off = 0; best.back = 0; }
    return best;
}

enum : int { kLzStarts = 64 };      // meeting points noted per segment
struct LzNoHook { DEVM bool operator()(int, int, int) const { return false; } };

// Levels 3..9 of an independent block, from state kLzFirst at ipStart until that state is reached again at or behind ipStop (or
// the block ends).  w.pre / w.rank / w.list = the block's chain and lists.  hook(q, records so far) is called whenever a first
// match has been found at q, before anything is decided about it; returning true ends the walk there (endIp = q).  Records go
// to seq[0..).
// kPa: level 9 (pattern analysis).  A template parameter so that the code of levels 3..8 does not carry the general finder (three
// inlined copies of it cost levels 3..6 a quarter of their speed: instruction cache).
// kD: the block (`blk`, n bytes) has an external segment of pfx bytes right in front of it in memory (a linked block after
// LZ4_loadDictHC(previous tail), or a block > 4 KiB under an attached dictionary: lz4hc.c:1438-1461, :1626-1720) and w's chain and
// lists were built over segment + block (positions from the segment's first byte; the segment's last three positions left out,
// :1660-1678).  The interface stays in positions of the BLOCK -- ipStart / anchorStart / ipStop, the hook's arguments, the records,
// the LzRun that comes back -- so segments, stitching and the emit stage do not know.
template <bool kPa, bool kD, class Hook>
DEV LzRun hc_lazy_run_t(const uint8_t* __restrict__ blk, const int pfxArg, const int n, const int level, HcWork w, uint64_t* seq,
                        const int ipStart, const int anchorStart, const int ipStopRel, Hook& hook)
{
    const int  maxNb = 1 << (level - 1);
    const bool pa = kPa;
    const int  pfx = kD ? pfxArg : 0;
    const uint8_t* const src = blk - pfx;
    HcState s; s.src = src; s.pfx = pfx; s.w = w; s.nextToUpdate = 0; s.d.mode = kD ? kHcExt : kHcNone; s.d.len = pfx; s.d.bytes = nullptr; s.d.hash = nullptr; s.d.chain = nullptr;
    const int mflimit = pfx + n - kMfLimit, matchlimit = pfx + n - kLastLiterals;
    const int kOptimalMl = 15 - 1 + kMinMatch;                                                   // OPTIMAL_ML, lz4hc.c:75
    SeqSink out; out.seq = seq; out.n = 0; out.bias = pfx;
    LANES({ out.buf[I_] = 0; })
    int ip = pfx + ipStart, anchor = pfx + anchorStart;
    const int ipStop = pfx + ipStopRel;
    LzRun run; run.cnt = 0; run.endIp = ipStart; run.finished = 1; run.anchor = anchorStart;
    if (n < kMinLength) return run;                                                              // :1155

    // one candidate per lane (hc_find_few), first searches in groups of 64 / attempts; level 9 falls back to the general finder
    // for the searches that meet a link of 1
    auto wider = [&](int pos, int low, int longest) {
        const LzFound f = hc_find_few<kD>(s, pos, low, matchlimit, longest, maxNb, false, mflimit, pa);
        if (kPa && f.pos == -2) return hc_find_wider_lists<kD>(s, pos, low, matchlimit, longest, maxNb, pa, false);
        HcMatch m; m.len = f.len; m.off = f.off; m.back = f.back;
        return m;
    };

    int st = kLzFirst;
    int start0 = 0, start2 = 0, start3 = 0;
    HcMatch m0 = {0, 0, 0}, m1 = {0, 0, 0}, m2 = {0, 0, 0}, m3 = {0, 0, 0};
    for (;;) {
        if (st == kLzFirst) {
            // the literal run: the first position at or behind ip whose first search found something (:1157-1162)
            bool found = false;
            if (ip >= ipStop && ip <= mflimit) { run.finished = 0; break; }
            while (ip <= mflimit) {
                const LzFound f = hc_find_few<kD>(s, ip, ip, matchlimit, kMinMatch - 1, maxNb, true, mflimit, pa);
                if (kPa && f.pos == -2) {                            // (level 9, a link of 1 among ip's candidates)
                    m1 = hc_find_wider_lists<kD>(s, ip, ip, matchlimit, kMinMatch - 1, maxNb, pa, false);
                    if (m1.len < kMinMatch) { ip++; continue; }
                } else {
                    if (f.pos < 0) { ip += maxNb < 64 ? 64 / maxNb : 1; continue; }
                    ip = f.pos; m1.len = f.len; m1.off = f.off; m1.back = 0;
                }
                found = true;
                break;
            }
            if (!found) break;
            if (hook(ip - pfx, 0, out.n)) { run.finished = 0; break; }
            start0 = ip; m0 = m1;
            st = kLzSecond;
        }
        if (st == kLzSecond) {
            // one match in hand: is there a longer one that starts inside it? (:1167-1196)
            m2.len = 0; m2.off = 0; m2.back = 0;
            if (ip + m1.len <= mflimit) {
                start2 = ip + m1.len - 2;
                m2 = wider(start2, ip, m1.len);
                start2 += m2.back;
            }
            if (m2.len <= m1.len) {                                                              // no: m1 goes out
                out.put(ip, m1.len, m1.off);
                ip += m1.len; anchor = ip;
                st = kLzFirst;
                continue;
            }
            if (start0 < ip && start2 < ip + m0.len) { ip = start0; m1 = m0; }                   // :1186-1189
            if (start2 - ip < 3) { ip = start2; m1 = m2; continue; }                             // m1 too short to keep: m2 takes its place
            st = kLzThird;
        }
        // kLzThird -- two overlapping matches in hand: a third one? (:1198-1306)
        if (start2 - ip < kOptimalMl) {                                                          // :1199-1210
            int newMl = m1.len;
            if (newMl > kOptimalMl) newMl = kOptimalMl;
            if (ip + newMl > start2 + m2.len - kMinMatch) newMl = (start2 - ip) + m2.len - kMinMatch;
            const int correction = newMl - (start2 - ip);
            if (correction > 0) { start2 += correction; m2.len -= correction; }
        }
        m3.len = 0; m3.off = 0; m3.back = 0;
        if (start2 + m2.len <= mflimit) {                                                        // :1212-1220
            start3 = start2 + m2.len - 3;
            m3 = wider(start3, start2, m2.len);
            start3 += m3.back;
        }
        if (m3.len <= m2.len) {                                                                  // no: m1 (cut at m2's start) and m2 go out, :1222-1240
            if (start2 < ip + m1.len) m1.len = start2 - ip;
            out.put(ip, m1.len, m1.off);
            out.put(start2, m2.len, m2.off);
            ip = start2 + m2.len; anchor = ip;
            st = kLzFirst;
            continue;
        }
        if (start3 < ip + m1.len + 3) {                                                          // :1242-1270
            if (start3 >= ip + m1.len) {                                                         // m3 leaves no room for m2: m1 goes out, m3 is the match in hand
                if (start2 < ip + m1.len) {
                    const int correction = ip + m1.len - start2;
                    start2 += correction;
                    m2.len -= correction;
                    if (m2.len < kMinMatch) { start2 = start3; m2 = m3; }
                }
                out.put(ip, m1.len, m1.off);
                anchor = ip + m1.len;
                ip = start3; m1 = m3;
                start0 = start2; m0 = m2;
                st = kLzSecond;
                continue;
            }
            start2 = start3; m2 = m3;                                                            // m3 swallows m2
            continue;
        }
        if (start2 < ip + m1.len) {                                                              // three in a row: m1 goes out, :1277-1306
            if (start2 - ip < kOptimalMl) {
                if (m1.len > kOptimalMl) m1.len = kOptimalMl;
                if (ip + m1.len > start2 + m2.len - kMinMatch) m1.len = (start2 - ip) + m2.len - kMinMatch;
                const int correction = m1.len - (start2 - ip);
                if (correction > 0) { start2 += correction; m2.len -= correction; }
            } else m1.len = start2 - ip;
        }
        out.put(ip, m1.len, m1.off);
        anchor = ip + m1.len;
        ip = start2; m1 = m2;
        start2 = start3; m2 = m3;
    }
    out.finish();
    run.cnt = out.n; run.endIp = ip - pfx; run.anchor = anchor - pfx;
    return run;
}

// ---- levels 10..11: LZ4HC_compress_optimal (lz4hc.c:1823-2123) without the full update (:1406: level 12 only, which has its own
// design, lz4hc12_device.inl).  The price DP over a window of up to 4096 positions is the reference's; its searches are wave-wide
// (hc_find_wider_lists with the chain swap, 96 / 512 attempts: their length to beat, last_match_pos - cur, is walk state, so they
// cannot be computed ahead like level 12's), its sequences are records, and it is walked in segments like the lazy levels.  One
// difference: at the top of its loop (:1863) this parser's state is (ip, anchor) -- the literal run in front of a window enters
// its prices (:1885-1894) -- so two walks have met when they find their next first match at the same position WITH the same anchor.
// kD: behind an external segment of pfx bytes (see hc_lazy_run_t).  There level 12 takes this route as well (its own design,
// lz4hc12_device.inl, computes every position's search up front on the independent block): 16384 attempts, sufficient_len 4095
// and the full update (:1406, :1929-1936: the skip test looks four positions ahead and every search starts from MINMATCH - 1).
template <bool kD, class Hook>
DEV LzRun hc_opt_run(const uint8_t* __restrict__ blk, const int pfxArg, const int n, const int level, HcWork w, uint64_t* seq,
                     const int ipStart, const int anchorStart, const int ipStopRel, Hook& hook)
{
    const int  nbSearches = level >= 12 ? 16384 : (level == 11 ? 512 : 96);                      // table :92-106
    const int  sufficient = level >= 12 ? kHcOptNum - 1 : (level == 11 ? 128 : 64);              // (:1860: capped to LZ4_OPT_NUM - 1)
    const bool fullUpdate = level >= 12;
    HcOpt* const opt = w.opt;
    const int pfx = kD ? pfxArg : 0;
    const uint8_t* const src = blk - pfx;
    HcState s; s.src = src; s.pfx = pfx; s.w = w; s.nextToUpdate = 0; s.d.mode = kD ? kHcExt : kHcNone; s.d.len = pfx; s.d.bytes = nullptr; s.d.hash = nullptr; s.d.chain = nullptr;
    const int mflimit = pfx + n - kMfLimit, matchlimit = pfx + n - kLastLiterals;
    SeqSink out; out.seq = seq; out.n = 0; out.bias = pfx;
    LANES({ out.buf[I_] = 0; })
    int ip = pfx + ipStart, anchor = pfx + anchorStart;
    const int ipStop = pfx + ipStopRel;
    LzRun run; run.finished = 1;
    while (ip <= mflimit) {                                                                      // :1863
        if (ip >= ipStop) { run.finished = 0; break; }
        const int llen = ip - anchor;
        int bestMl = 0, bestOff = 0, cur, last = 0;
        const HcMatch first = hc_find_longer<kD>(s, ip, matchlimit, kMinMatch - 1, nbSearches);
        if (first.len == 0) { ip++; continue; }
        if (hook(ip - pfx, anchor - pfx, out.n)) { run.finished = 0; break; }
        if (first.len > sufficient) {                                                            // :1871-1882
            out.put(ip, first.len, first.off);
            ip += first.len; anchor = ip;
            continue;
        }
        // the window's table from the first match (:1885-1919), one entry per lane
        for (int i0 = 0; i0 <= first.len + kHcTrailing; i0 += 64) {
            const int pm = hc_seq_price(llen, first.len);
            LANES({
                const int i = i0 + LANE;
                HcOpt e;
                if (i < kMinMatch) { e.mlen = 1; e.off = 0; e.litlen = llen + i; e.price = hc_lit_price(llen + i); opt[i] = e; }
                else if (i <= first.len) { e.mlen = i; e.off = first.off; e.litlen = llen; e.price = hc_seq_price(llen, i); opt[i] = e; }
                else if (i <= first.len + kHcTrailing) { e.mlen = 1; e.off = 0; e.litlen = i - first.len; e.price = pm + hc_lit_price(i - first.len); opt[i] = e; }
            })
        }
        WAVE_FENCE();
        last = first.len;
        bool direct = false;
        for (cur = 1; cur < last; ++cur) {                                                       // :1922-2019
            const int curPos = ip + cur;
            if (curPos > mflimit) break;
            if (fullUpdate) { if (opt[cur + 1].price <= opt[cur].price && opt[cur + kMinMatch].price < opt[cur].price + 3) continue; }   // :1929-1931
            else if (opt[cur + 1].price <= opt[cur].price) continue;                             // :1932-1934
            const HcMatch nm = hc_find_longer<kD>(s, curPos, matchlimit, fullUpdate ? kMinMatch - 1 : last - cur, nbSearches);
            if (!nm.len) continue;
            if (nm.len > sufficient || nm.len + cur >= kHcOptNum) {                              // :1948-1956
                bestMl = nm.len; bestOff = nm.off; last = cur + 1; direct = true;
                break;
            }
            {   const int baseLit = opt[cur].litlen;                                             // :1958-1972
                for (int l = 1; l < kMinMatch; ++l) {
                    const int price = opt[cur].price - hc_lit_price(baseLit) + hc_lit_price(baseLit + l);
                    const int pos = cur + l;
                    if (price < opt[pos].price) { opt[pos].mlen = 1; opt[pos].off = 0; opt[pos].litlen = baseLit + l; opt[pos].price = price; }
                }
            }
            {   // every length of the match, one per lane (:1974-2009: each reads and writes its own entry; `last` moves at the last one)
                const int ll = (opt[cur].mlen == 1) ? opt[cur].litlen : 0;
                const int basePrice = (opt[cur].mlen == 1) ? ((cur > ll) ? opt[cur - ll].price : 0) : opt[cur].price;
                const int lastOld = last;
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
                    if (nm.len - ml0 < 64) {
                        const int lastLane = nm.len - ml0;
                        if (RL(took, lastLane) && lastOld < cur + nm.len) last = cur + nm.len;
                    }
                }
            }
            for (int a = 1; a <= kHcTrailing; ++a) {                                             // :2011-2018
                opt[last + a].mlen = 1; opt[last + a].off = 0; opt[last + a].litlen = a;
                opt[last + a].price = opt[last].price + hc_lit_price(a);
            }
        }
        if (!direct) { bestMl = opt[last].mlen; bestOff = opt[last].off; cur = last - bestMl; }   // :2022-2024
        {   // the chosen path, marked backwards (:2026-2046) ...
            int cand = cur, selML = bestMl, selOff = bestOff;
            for (;;) {
                const int nextML = opt[cand].mlen, nextOff = opt[cand].off;
                opt[cand].mlen = selML; opt[cand].off = selOff;
                selML = nextML; selOff = nextOff;
                if (nextML > cand) break;
                cand -= nextML;
            }
        }
        {   // ... and its sequences in order (:2048-2064)
            int r = 0;
            while (r < last) {
                const int ml = opt[r].mlen, off = opt[r].off;
                if (ml == 1) { ip++; r++; continue; }
                r += ml;
                out.put(ip, ml, off);
                ip += ml; anchor = ip;
            }
        }
    }
    out.finish();
    run.cnt = out.n; run.endIp = ip - pfx; run.anchor = anchor - pfx;
    return run;
}

// pfx < 0: an independent block without a segment (the instantiations the frame path's blocks have always run); pfx >= 0: kD
template <bool kD, class Hook>
DEV LzRun hc_lazy_run(const uint8_t* __restrict__ blk, const int pfx, const int n, const int level, HcWork w, uint64_t* seq,
                      const int ipStart, const int anchorStart, const int ipStop, Hook& hook)
{
    if (level >= 10) return hc_opt_run<kD>(blk, pfx, n, level, w, seq, ipStart, anchorStart, ipStop, hook);
    if ((1 << (level - 1)) > 128) return hc_lazy_run_t<true, kD>(blk, pfx, n, level, w, seq, ipStart, anchorStart, ipStop, hook);      // pattern analysis above 128 attempts
    return hc_lazy_run_t<false, kD>(blk, pfx, n, level, w, seq, ipStart, anchorStart, ipStop, hook);
}

// ---- segments.  Layout of a block's workspace (entries of 8 bytes): rec[j * segCap ..): the records of segment j;
// bridge[j * segCap ..): those of the walk that leads into it.  meta[j]: what the two walks left behind.
struct LzSegMeta {
    LzRun seg, bridge;
    int nStarts;        // entries of starts[]: (q << 32) | records before it
    int skip;           // stitched: the segment's records count from here on (== seg.cnt: none of them)
    int pad0, pad1;
};
struct LzPiece { const uint64_t* src; int cnt; int dst; };
DEV int lz_segments(int n, int maxSegs, int minSeg) { int s = n / (minSeg > 0 ? minSeg : 1); if (s > maxSegs) s = maxSegs; return s < 1 ? 1 : s; }
DEV int lz_seg_len(int n, int segs) { return (n + segs - 1) / segs; }
DEV int lz_seg_cap(int segLen) { return ((segLen / kMinMatch + 8) + 63) & ~63; }

// pass 1, one wave per (block, segment)
// a noted start: q (22 bits) | the anchor it was found with (22; 0 at the lazy levels, whose state does not hold it) | records before it (20)
DEV uint64_t lz_start(int q, int anchor, int recs) { return ((uint64_t)(uint32_t)q << 42) | ((uint64_t)(uint32_t)anchor << 20) | (uint32_t)recs; }
DEV int lz_start_q(uint64_t v) { return (int)(v >> 42); }
DEV int lz_start_anchor(uint64_t v) { return (int)((v >> 20) & 0x3FFFFFu); }
DEV int lz_start_recs(uint64_t v) { return (int)(v & 0xFFFFFu); }
struct LzStartNote {
    uint64_t* starts; int n;
    DEVM bool operator()(int q, int anchor, int recs)
    {
        if (n < kLzStarts) { const uint64_t v = lz_start(q, anchor, recs); const int at = n; LANES({ if (LANE == 0) starts[at] = v; }) ++n; }
        return false;
    }
};
// run(seq, ipStart, anchorStart, ipStop, hook) -> LzRun: the level's walk (hc_lazy_run; level 12: hc12_walk<true>)
template <class Run>
DEV void lz_segment(Run& run, int n, int segs, int j, uint64_t* rec, LzSegMeta* meta, uint64_t* starts)
{
    const int segLen = lz_seg_len(n, segs), cap = lz_seg_cap(segLen);
    LzStartNote note; note.starts = starts + (size_t)j * kLzStarts; note.n = 0;
    const int b0 = j * segLen, b1 = j + 1 == segs ? n + 1 : (j + 1) * segLen;
    const LzRun r = run(rec + (size_t)j * cap, b0, b0, b1, note);
    const int ns = note.n;
    LANES({ if (LANE == 0) { meta[j].seg = r; meta[j].nStarts = ns; meta[j].skip = 0; meta[j].bridge.cnt = 0; } })
}
template <bool kD = false>
DEV void hc_lazy_segment(const uint8_t* __restrict__ src, int n, int level, HcWork w, int segs, int j,
                         uint64_t* rec, LzSegMeta* meta, uint64_t* starts, int pfx = 0)
{
    auto run = [&](uint64_t* seq, int ipStart, int anchorStart, int ipStop, auto& hook) { return hc_lazy_run<kD>(src, pfx, n, level, w, seq, ipStart, anchorStart, ipStop, hook); };
    lz_segment(run, n, segs, j, rec, meta, starts);
}

// pass 2, one wave per block: from where the walk before segment j truly ended to a meeting point with segment j
struct LzMeet {
    const uint64_t* starts; int nStarts, at, skip, met;
    DEVM bool operator()(int q, int anchor, int)
    {
        while (at < nStarts && lz_start_q(starts[at]) < q) ++at;
        if (at < nStarts && lz_start_q(starts[at]) == q && lz_start_anchor(starts[at]) == anchor) { met = 1; skip = lz_start_recs(starts[at]); return true; }
        return false;
    }
};
// pieces[0 .. 2*segs): what the block's records are, in order; returns their total, *lastAnchor as the one walk's
template <class Run>
DEV int lz_stitch(Run& run, int n, int segs, uint64_t* rec, uint64_t* bridge, LzSegMeta* meta, const uint64_t* starts, LzPiece* pieces, int* lastAnchor)
{
    const int segLen = lz_seg_len(n, segs), cap = lz_seg_cap(segLen);
    LzRun cur = meta[0].seg;                      // the true walk so far ends like this
    int total = cur.cnt;
    { const LzPiece p0 = { rec, cur.cnt, 0 }; const LzPiece none = { rec, 0, 0 }; LANES({ if (LANE == 0) { pieces[0] = none; pieces[1] = p0; } }) }
    for (int j = 1; j < segs; ++j) {
        LzPiece pb = { bridge + (size_t)j * cap, 0, total }, ps = { rec + (size_t)j * cap, 0, total };
        const int b1 = j + 1 == segs ? n + 1 : (j + 1) * segLen;
        if (!cur.finished && cur.endIp < b1) {
            LzMeet meet; meet.starts = starts + (size_t)j * kLzStarts; meet.nStarts = meta[j].nStarts; meet.at = 0; meet.skip = 0; meet.met = 0;
            const LzRun br = run(bridge + (size_t)j * cap, cur.endIp, cur.anchor, b1, meet);
            pb.cnt = br.cnt;
            total += br.cnt;
            if (meet.met) {                          // from here on segment j's walk is the true one
                ps.src += meet.skip; ps.cnt = meta[j].seg.cnt - meet.skip; ps.dst = total;
                total += ps.cnt;
                cur = meta[j].seg;
            } else cur = br;                         // (ran into the next segment, or to the block's end)
        }
        LANES({ if (LANE == 0) { pieces[2 * j] = pb; pieces[2 * j + 1] = ps; } })
    }
    *lastAnchor = cur.anchor;
    return total;
}
template <bool kD = false>
DEV int hc_lazy_stitch(const uint8_t* __restrict__ src, int n, int level, HcWork w, int segs,
                       uint64_t* rec, uint64_t* bridge, LzSegMeta* meta, const uint64_t* starts, LzPiece* pieces, int* lastAnchor, int pfx = 0)
{
    auto run = [&](uint64_t* seq, int ipStart, int anchorStart, int ipStop, auto& hook) { return hc_lazy_run<kD>(src, pfx, n, level, w, seq, ipStart, anchorStart, ipStop, hook); };
    return lz_stitch(run, n, segs, rec, bridge, meta, starts, pieces, lastAnchor);
}
// level 12 (lz4hc12_device.inl): its price DP over the search results F, walked in segments the same way
DEV void hc12_segment(const uint8_t* __restrict__ src, int n, const Hc12F* F, const uint16_t* chain, const Hc12Ws& w, int segs, int j,
                      uint64_t* rec, LzSegMeta* meta, uint64_t* starts)
{
    auto run = [&](uint64_t* seq, int ipStart, int anchorStart, int ipStop, auto& hook) {
        LzRun r; r.cnt = 0; r.endIp = ipStart; r.finished = 1; r.anchor = anchorStart;
        hc12_walk<true>(src, n, nullptr, 0, F, chain, w, seq, ipStart, anchorStart, ipStop, hook, &r);
        return r;
    };
    lz_segment(run, n, segs, j, rec, meta, starts);
}
DEV int hc12_stitch(const uint8_t* __restrict__ src, int n, const Hc12F* F, const uint16_t* chain, const Hc12Ws& w, int segs,
                    uint64_t* rec, uint64_t* bridge, LzSegMeta* meta, const uint64_t* starts, LzPiece* pieces, int* lastAnchor)
{
    auto run = [&](uint64_t* seq, int ipStart, int anchorStart, int ipStop, auto& hook) {
        LzRun r; r.cnt = 0; r.endIp = ipStart; r.finished = 1; r.anchor = anchorStart;
        hc12_walk<true>(src, n, nullptr, 0, F, chain, w, seq, ipStart, anchorStart, ipStop, hook, &r);
        return r;
    };
    return lz_stitch(run, n, segs, rec, bridge, meta, starts, pieces, lastAnchor);
}

// ---------------------------------------------------------------------------------------------------------------- level 2
// LZ4MID_compress (lz4hc.c:521-775) keeps two direct-mapped tables -- the last position inserted per hash of 4 bytes and of 7
// bytes (:141-151) -- and inserts only the positions it searches (every position of a literal run, then two after a match's
// start and four before its end, :678-706).  What a search finds therefore depends on the walk so far; but while the walk is in
// a literal run it is easy to say what the tables WOULD hold if the run went on: every position searched so far inserted in
// both.  hc_mid_parse takes the next positions of the run one per lane on exactly that assumption:
//   * the three table entries a position may look at (its long hash, its short hash, and the long hash of the position behind
//     it, :616) are read for all lanes at once -- one memory round trip instead of one per position;
//   * a lane's candidate is the nearest lower lane of the batch with the same hash, else what the table held before the batch;
//   * the first lane whose candidate matches ends the run: the lanes below it were literals, as assumed, so its own candidates
//     are the true ones; the lanes above it were never searched and leave no trace.
//   The lanes up to that one commit their inserts (the last lane of a hash stores), then the one sequence is handled as the
//   reference does (the look at ip + 1, the catch-up, the table fills around the match), and the next batch starts behind it.
// Batches are ONE lane wide behind a match (measured: 1 / 2 / 3 / 4 / 6 / 16 lanes = 7831 / 7684 / 7454 / 7245 / 6771 / 5395 MiB/s on
// the T text, where a match starts every 10.6 bytes; measuring the one position's three candidates in the round trip of their
// 4-byte test instead of after it: 7548, dropped), then 16, then 64 in a literal run; the step between positions (:668: 1 + run length / 512)
// is constant inside a batch.  The tables (128 KiB) live in device memory, per wave.  Output: records, as for levels 3..9.
DEV uint32_t mid_hash4v(uint32_t v) { return (v * 2654435761u) >> (32 - 14); }
DEV uint32_t mid_hash8v(uint64_t v) { return (uint32_t)(((v << 8) * 58295818150454627ull) >> (64 - 14)); }

// Everything the sequence at a batch's matching lane needs measured, in ONE memory round trip: the match at P1 (distance D1)
// forwards, the candidate of the look at ip + 1 (P2 = P1 + 1, distance D2, when alt) forwards, and both backwards (the catch-up,
// :673-675, at most down to the anchor and to the block's start).  Lanes 0..23 / 24..47: eight bytes each of the two forward
// counts; 48..55 / 56..63: eight bytes each backwards.  Longer ones go on with the whole wave.
struct MidMeasure { int ml1, ml2, bk1, bk2; };
// lim1: where the count of the match at P1 ends (matchlimit; the segment's end for a candidate inside an external segment: safeLen,
// lz4hc.c:587-596, :637-646); lowPos: the catch-up does not take the match below this position (the block's first byte, :673:
// `(ip - prefixPtr) > matchDistance`; 0 without a segment).
DEV MidMeasure mid_measure(const uint8_t* __restrict__ src, int P1, int D1, bool alt, int P2, int D2, int anchor, int matchlimit, int lim1, int lowPos)
{
    LV(int, cnt);
    LANES({
        const bool fwd = LANE < 48, second = fwd ? LANE >= 24 : LANE >= 56;
        const int j = fwd ? (second ? LANE - 24 : LANE) : (second ? LANE - 56 : LANE - 48);
        const int P = second ? P2 : P1, D = second ? D2 : D1;
        int c = 0;
        if (!second || alt) {
            if (fwd) {
                const int off = 8 * j, valid = ((second ? matchlimit : lim1) - P) - off;
                if (valid > 0) {
                    uint64_t x = ld64p_guard(src + P + off, valid) ^ ld64p_guard(src + P - D + off, valid);
                    if (valid < 8) x |= ~0ull << (8 * valid);
                    c = x ? (ctz64(x) >> 3) : 8;
                }
            } else {
                const int maxBack = min_(P - anchor, max_(P - D - lowPos, 0)), usable = maxBack - 8 * j;
                if (usable >= 8) {
                    const uint64_t x = ld64u(src + P - 8 * (j + 1)) ^ ld64u(src + P - D - 8 * (j + 1));
                    c = x ? (int)(__builtin_clzll(x) >> 3) : 8;
                } else for (; c < usable && src[P - 8 * j - 1 - c] == src[P - D - 8 * j - 1 - c]; ++c) {}
            }
        }
        cnt[I_] = c;
    })
    const uint64_t stop = BALLOT(cnt[I_] < 8);
    MidMeasure r;
    auto part = [&](int lo, int nl) -> int {                     // 8 * full chunks + the first short one; -1: every chunk full
        const uint64_t sm = (stop >> lo) & (((uint64_t)1 << nl) - 1);
        if (!sm) return -1;
        const int j0 = ctz64(sm);
        return 8 * j0 + RL(cnt, lo + j0);
    };
    r.ml1 = part(0, 24);
    if (r.ml1 < 0) r.ml1 = 192 + wave_count_ptr(src + P1 + 192, src + P1 - D1 + 192, (lim1 - P1) - 192);
    r.ml2 = 0; r.bk2 = 0;
    if (alt) {
        r.ml2 = part(24, 24);
        if (r.ml2 < 0) r.ml2 = 192 + wave_count_ptr(src + P2 + 192, src + P2 - D2 + 192, (matchlimit - P2) - 192);
        r.bk2 = part(56, 8);
        if (r.bk2 < 0) r.bk2 = 64 + wave_back_ptr(src + P2 - 64, src + P2 - D2 - 64, min_(P2 - anchor, max_(P2 - D2 - lowPos, 0)) - 64);
    }
    r.bk1 = part(48, 8);
    if (r.bk1 < 0) r.bk1 = 64 + wave_back_ptr(src + P1 - 64, src + P1 - D1 - 64, min_(P1 - anchor, max_(P1 - D1 - lowPos, 0)) - 64);
    return r;
}

#ifndef PLZ4_MID_WIDTH0
#define PLZ4_MID_WIDTH0 1
#endif
enum : int { kMidWidth0 = PLZ4_MID_WIDTH0 };       // lanes of the batch right behind a match (text: a match every other position); then 16, then 64
#if defined(PLZ4_EMU)
static inline void gmem_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
#else
__device__ __forceinline__ void gmem_max(uint32_t* p, uint32_t v) { atomicMax(p, v); }
#endif
// LZ4MID_fillHTable (lz4hc.c:477-503) over the external segment [0, size) by the whole wave: what LZ4_loadDictHC leaves in the two
// (zeroed) tables.  The reference's two loops store in ascending order, the later store wins: a maximum per slot and loop; the
// second loop (every position of the last 32 KiB into the long table) overrides the first (every third position into both), so its
// values carry a mark until everything is in.
DEV void hc_mid_prime(const uint8_t* __restrict__ src, const int size, uint32_t* h4t, uint32_t* h8t)
{
    if (size <= 8) return;
    const int target = size - 8;
    LANES({
        for (int k = LANE; 3 * k < target; k += 64) {
            const int p = 3 * k;
            gmem_max(&h4t[mid_hash4v(ld32u(src + p))], (uint32_t)p + kHcBase);
            gmem_max(&h8t[mid_hash8v(ld64u(src + p + 1))], (uint32_t)p + 1u + kHcBase);
        }
    })
    WAVE_FENCE();
    const int from = size > 32768 + 8 ? target - 32768 : 0;
    LANES({ for (int p = from + LANE; p < target; p += 64) gmem_max(&h8t[mid_hash8v(ld64u(src + p))], ((uint32_t)p + kHcBase) | 0x80000000u); })
    WAVE_FENCE();
    LANES({ for (int i = LANE; i < 16384; i += 64) h8t[i] &= 0x7FFFFFFFu; })
    WAVE_FENCE();
}

// kD: the block (`blk`, n bytes) has an external segment of pfxArg bytes right in front of it (a linked block after LZ4_loadDictHC, a
// block > 4 KiB under an attached dictionary); positions count from the segment's first byte inside, records from the block's.  What
// the reference does differently there: the tables start from LZ4MID_fillHTable over the segment; a candidate inside the segment
// matches up to the segment's end only (safeLen, :587-596, :637-646) and gets neither the look at ip + 1 (:616) nor a catch-up (:673).
template <bool kD = false>
DEV int hc_mid_parse(const uint8_t* __restrict__ blk, const int n, uint32_t* h4t, uint32_t* h8t, uint64_t* seq, int* lastAnchor, const int pfxArg = 0)
{
    const int pfx = kD ? pfxArg : 0;
    const uint8_t* const src = blk - pfx;
    const int N = pfx + n;
    const uint32_t prefixIdx = kHcBase + (uint32_t)pfx;
    *lastAnchor = 0;
    LANES({ for (int i = LANE; i < 16384; i += 64) { h4t[i] = 0u; h8t[i] = 0u; } })            // LZ4_initStreamHC: everything zero, lz4hc.c:1582-1583
    WAVE_FENCE();
    if (kD) hc_mid_prime(src, pfx, h4t, h8t);
    SeqSink out; out.seq = seq; out.n = 0; out.bias = pfx;
    LANES({ out.buf[I_] = 0; })
    const int mflimit = N - kMfLimit, matchlimit = N - kLastLiterals;
    const uint32_t ilimitIdx = (uint32_t)(N - 8) + kHcBase;
    int ip = pfx, anchor = pfx, width = kMidWidth0;
    if (n < kMinLength || ip > mflimit) { return 0; }

    // a batch: the lanes' positions and their bytes (requested as soon as the batch's start is known: before the table fills of
    // the sequence in front of it are made)
    LV(int, pk); LV(bool, act); LV(uint64_t, a64); LV(uint64_t, b64);
    int L = 0, step = 1;
    auto fetch = [&]() {
        const int run9 = (ip - anchor) >> 9;                                                   // :668: the step grows with the run
        step = 1 + run9;
        const int ip0 = ip, an = anchor, wd = width, st = step;
        LANES({
            pk[I_] = ip0 + LANE * st;
            act[I_] = LANE < wd && pk[I_] <= mflimit && ((pk[I_] - an) >> 9) == run9;
            a64[I_] = 0; b64[I_] = 0;
            if (act[I_]) { a64[I_] = ld64u(src + pk[I_]); b64[I_] = ld64u(src + pk[I_] + 1); }
        })
        L = __builtin_popcountll(BALLOT(act[I_]));                                             // (lane 0 is in while ip <= mflimit)
    };
    fetch();
    while (ip <= mflimit) {
        // hashes and table entries
        LV(uint32_t, h8); LV(uint32_t, h4); LV(uint32_t, h8x); LV(uint32_t, t8); LV(uint32_t, t4); LV(uint32_t, x8);
        LANES({
            h8[I_] = 0; h4[I_] = 0; h8x[I_] = 0; t8[I_] = 0; t4[I_] = 0; x8[I_] = 0;
            if (act[I_]) {
                h8[I_] = mid_hash8v(a64[I_]); h4[I_] = mid_hash4v((uint32_t)a64[I_]); h8x[I_] = mid_hash8v(b64[I_]);
                t8[I_] = h8t[h8[I_]]; t4[I_] = h4t[h4[I_]]; x8[I_] = h8t[h8x[I_]];
            }
        })
        // inside the batch: the nearest lower lane with the same hash takes the table's place; n8 / n4: the nearest HIGHER
        // lane with a lane's hash (who stores, further down)
        LV(int, n8); LV(int, n4);
        LANES({ n8[I_] = 64; n4[I_] = 64; })
        for (int i = 0; i < L; ++i) {
            const uint32_t hi8 = RL(h8, i), hi4 = RL(h4, i);
            const uint32_t idx = (uint32_t)(ip + i * step) + kHcBase;
            const uint64_t above = ~(((uint64_t)2 << i) - 1);                                 // lanes > i (i == 63: none)
            const uint64_t e8 = BALLOT(act[I_] && h8[I_] == hi8) & above;
            const uint64_t e4 = BALLOT(act[I_] && h4[I_] == hi4) & above;
            const int nx8 = e8 ? ctz64(e8) : 64, nx4 = e4 ? ctz64(e4) : 64;
            LANES({
                if ((e8 >> LANE) & 1) t8[I_] = idx;
                if ((e4 >> LANE) & 1) t4[I_] = idx;
                if (act[I_] && LANE >= i && h8x[I_] == hi8) x8[I_] = idx;                     // (a lane's own long insert comes before its look at ip + 1)
                if (LANE == i) { n8[I_] = nx8; n4[I_] = nx4; }
            })
        }
        // whose candidate matches (>= MINMATCH: four equal bytes, within 65535)
        LV(bool, has8); LV(bool, has4);
        LANES({
            has8[I_] = false; has4[I_] = false;
            if (act[I_]) {
                const uint32_t idx = (uint32_t)pk[I_] + kHcBase;
                // (a candidate inside the segment is counted up to the segment's end: four bytes of it have to be there)
                if (idx - t8[I_] <= 65535u && (!kD || t8[I_] >= prefixIdx || prefixIdx - t8[I_] >= 4u)) has8[I_] = ld32u(src + (t8[I_] - kHcBase)) == (uint32_t)a64[I_];
                if (idx - t4[I_] <= 65535u && (!kD || t4[I_] >= prefixIdx || prefixIdx - t4[I_] >= 4u)) has4[I_] = ld32u(src + (t4[I_] - kHcBase)) == (uint32_t)a64[I_];
            }
        })
        const uint64_t hit = BALLOT(act[I_] && (has8[I_] || has4[I_]));
        const int m = hit ? ctz64(hit) : L;
        // the inserts of the lanes that were searched: the long table always (:575), the short one unless the long one matched (:604)
        {
            const int last8 = m < L ? m : L - 1;
            const int last4 = (m < L && !RL(has8, m)) ? m : m - 1;
            LANES({
                const uint32_t idx = (uint32_t)pk[I_] + kHcBase;
                if (LANE <= last8 && n8[I_] > last8) h8t[h8[I_]] = idx;
                if (LANE <= last4 && n4[I_] > last4) h4t[h4[I_]] = idx;
            })
            LDS_ORDER();          // (the stores below may hit the same slots: they stay behind these, also for the compiler)
        }
        if (m >= L) { ip += L * step; width = width < 16 ? 16 : 64; if (ip <= mflimit) fetch(); continue; }

        // ---- the sequence at lane m (:572-706)
        int P = ip + m * step;
        const uint32_t ipIndex = (uint32_t)P + kHcBase;
        const bool via8 = RL(has8, m);
        const uint32_t cand = via8 ? RL(t8, m) : RL(t4, m);
        uint32_t dist = ipIndex - cand;
        const uint32_t pos8 = RL(x8, m), m2d = ipIndex + 1 - pos8;
        // one look at ip + 1 for a longer one, :616-650 (only behind a short match inside the block, only at a candidate inside the block)
        const bool alt = !via8 && m2d <= 65535u && pos8 >= prefixIdx && cand >= prefixIdx && P < mflimit;
        const int lim1 = (kD && cand < prefixIdx) ? min_(matchlimit, P + (int)(prefixIdx - cand)) : matchlimit;
        const MidMeasure mm = mid_measure(src, P, (int)dist, alt, P + 1, (int)m2d, anchor, matchlimit, lim1, pfx);
        int ml = mm.ml1, back = mm.bk1;
        if (alt && mm.ml2 > ml) {
            const uint32_t hx = RL(h8x, m);
            LANES({ if (LANE == 0) h8t[hx] = ipIndex + 1; })
            LDS_ORDER();
            P++; ml = mm.ml2; dist = m2d; back = mm.bk2;
        }
        P -= back; ml += back;                                                                 // :673-675
        out.put(P, ml, (int)dist);
        const int E = P + ml;
        const uint32_t endIdx = (uint32_t)E + kHcBase;
        const bool endOk = endIdx - 2 < ilimitIdx;                                             // :695
        ip = E; anchor = E; width = kMidWidth0;
        if (ip <= mflimit) fetch();                                                            // the next batch's bytes are on their way ...
        {   // ... while the tables are filled around the match: two positions behind its start -- with the index of the search
            // position although ip may have moved (:678-680) -- and four before its end (:695-706); one load each, the stores
            // in the reference's order
            LV(uint32_t, f8); LV(uint32_t, f4);
            LANES({
                const int at = LANE == 0 ? P + 1 : LANE == 1 ? P + 2 : LANE == 2 ? E - 5 : LANE == 3 ? E - 3 : LANE == 4 ? E - 2 : E - 1;
                const bool on = LANE < 2 || (LANE < 6 && endOk && (LANE != 2 || E - pfx > 5));
                uint64_t v = 0;
                if (on) v = LANE == 5 ? (uint64_t)ld32u(src + at) : ld64u(src + at);
                f8[I_] = mid_hash8v(v); f4[I_] = mid_hash4v((uint32_t)v);
            })
            // (one lane makes all of them: its own stores are ordered, two lanes' stores to one slot are not)
            const uint32_t s0 = RL(f8, 0), s1 = RL(f8, 1), s2 = RL(f8, 2), s3 = RL(f8, 3), s4 = RL(f8, 4), q0 = RL(f4, 0), q4 = RL(f4, 4), q5 = RL(f4, 5);
            const bool far = E - pfx > 5;
            LANES({ if (LANE == 0) {
                h8t[s0] = ipIndex + 1; h8t[s1] = ipIndex + 2; h4t[q0] = ipIndex + 1;
                if (endOk) {
                    if (far) h8t[s2] = endIdx - 5;
                    h8t[s3] = endIdx - 3; h8t[s4] = endIdx - 2;
                    h4t[q4] = endIdx - 2; h4t[q5] = endIdx - 1;
                }
            } })
        }
    }
    out.finish();
    *lastAnchor = anchor - pfx;
    return out.n;
}

}  // namespace plz4
