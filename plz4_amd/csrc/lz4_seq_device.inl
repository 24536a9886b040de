// lz4_seq_device.inl -- the level-1 encoder (LZ4_compress_fast, accel 1; /root/reference/internal/pkg/clz4/lz4.c:930-1338) cut
// in two along the one line that is serial in it:
//
//   PARSE  wave_parse_l1     one wavefront per block, its 16 KiB hash table in LDS: plays liblz4's greedy parser -- which
//                            positions are probed, what the table returns, where matches start and how far they run
//                            (lz4.c:1005-1101, :1182-1300) -- and writes ONE 8-byte record per sequence to device memory:
//                            (probe position, candidate distance, forward length).  Nothing else: no output bytes, no
//                            literal lengths, no back-extension, no capacity checks.  None of those feed back into the
//                            parser (the catch-up of lz4.c:1105-1109 moves the sequence's start, not its end; limitedOutput
//                            only decides whether the result is thrown away).
//   EMIT   seq_emit_*        one LANE per sequence, any number of waves per block: the catch-up, the literal and match
//                            lengths, token / length bytes / literals / offset (lz4.c:1112-1226), the last literals
//                            (lz4.c:1302-1329) and the limitedOutput verdict.  Sizes first (seq_emit_sizes, per chunk of
//                            1024 sequences), a scan over a block's chunks (seq_emit_scan), then every chunk is written at
//                            its final place (seq_emit_write).
//
// The serial wave therefore issues a fraction of the instructions the fused encoder of lz4_device.inl needs per 64 input
// bytes, and everything that is data-parallel runs at full occupancy with no LDS.  The fused encoder stays for what this
// path does not take: blocks above 4 MiB (raw block API; a record packs positions into 22 bits) and the dictionary / linked
// modes (wave_encode_block_ext / _dict).
//
// limitedOutput (lz4.c:1114-1117, :1187-1210, :1305-1314) as one comparison: liblz4 returns 0 iff the size of the COMPLETE
// block exceeds the capacity.  Every left-hand side of the per-sequence tests is <= that size (a match is followed by at
// least the last-literals token and its 5 literals, lz4.c:964: `tokenPos + 1 + lit + 8 + lit/255` <= end of the sequence + 6
// and `afterOffset + 6 + (mc + 240)/255` == end of the sequence + 6), and the last test (lz4.c:1305) is the complete size
// itself.  So the parser never needs the capacity, and the verdict is taken once the sizes are known (seq_emit_scan).
#pragma once
#include "lz4_device.inl"

namespace plz4 {

enum : int {
    kSeqMaxBlock = 1 << 22,        // positions and lengths are packed into 22 bits
    kSeqChunk    = 1024            // sequences per emit chunk
};

// per-block results of the parse / scan stages
struct SeqInfo { int32_t nseq; int32_t lastAnchor; int32_t total; int32_t stored; };

// entries a block of n bytes can need: every sequence consumes at least MINMATCH input bytes
DEV int seq_capacity(int n) { return n / kMinMatch + 2; }
static inline int seq_capacity_host(int n) { return n / 4 + 2; }

DEV uint64_t seq_pack(uint32_t pos, uint32_t fwd, uint32_t off) { return (uint64_t)pos | ((uint64_t)fwd << 22) | ((uint64_t)off << 44); }
DEV uint32_t seq_pos(uint64_t r) { return (uint32_t)r & 0x3FFFFFu; }
DEV uint32_t seq_fwd(uint64_t r) { return (uint32_t)(r >> 22) & 0x3FFFFFu; }
DEV uint32_t seq_off(uint64_t r) { return (uint32_t)(r >> 44) & 0xFFFFu; }

#if defined(PLZ4_EMU)
#define LANE_RANK(mask) (__builtin_popcountll((mask) & ((1ull << LANE) - 1ull)))
#else
#define LANE_RANK(mask) ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)((mask) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(mask), 0u)))
#endif

// 20 bytes from a position on: [x, x+20) as five dwords, kept as loaded (see Win24 in lz4_device.inl)
struct Win20 { uint32_t w[5]; };
DEV uint64_t win20_seq(const Win20& p) { return (uint64_t)p.w[0] | ((uint64_t)p.w[1] << 32); }
DEV Win20 load_win20(const uint8_t* src, int x)
{
    const v16u_t a = *(const v16u_t*)(src + x);
    Win20 w;
    w.w[0] = a.w[0]; w.w[1] = a.w[1]; w.w[2] = a.w[2]; w.w[3] = a.w[3];
    w.w[4] = ld32u(src + x + 16);
    return w;
}
// equal bytes of [x+4, x+20) in two windows: 0..16
DEV int win20_fwd(const Win20& p, const Win20& c)
{
    const uint32_t x1 = p.w[1] ^ c.w[1], x2 = p.w[2] ^ c.w[2], x3 = p.w[3] ^ c.w[3], x4 = p.w[4] ^ c.w[4];
    int n = x4 ? 12 + (__builtin_ctz(x4) >> 3) : 16;
    n = x3 ? 8 + (__builtin_ctz(x3) >> 3) : n;
    n = x2 ? 4 + (__builtin_ctz(x2) >> 3) : n;
    n = x1 ? (__builtin_ctz(x1) >> 3) : n;
    return n;
}

// ------------------------------------------------------------------------------------------ PARSE
// Plays LZ4_compress_generic(noDict, byU16 / byU32, accel 1) over src[0, n), n <= 4 MiB, and writes one record per sequence to
// seq[] (at most seq_capacity(n)).  Returns their number; *lastAnchor = where the last literals start (lz4.c:1302).
// `tab` = 16 KiB of LDS owned by this wave.  Two kinds of batch as in wave_encode_block_tt (lz4_device.inl), same parser state
// between them; see there for the grid batch's table protocol (tagged entries, atomic-max commit, min / max patch).
template <bool U16>
DEV int wave_parse_l1_tt(const uint8_t* __restrict__ src, const int n, void* tab, uint64_t* __restrict__ seq, int* lastAnchor)
{
    const int      sh      = U16 ? 0 : 10;
    const uint32_t tagMask = (1u << sh) - 1u;
    {   // fresh table per block (LZ4_initStream, lz4.c:1384): every slot = "position 0"
        const uint32_t e0 = (sh && n >= 4) ? (seq_tag(UNI(ld32u(src))) & tagMask) : 0u;
        uint32_t* t = (uint32_t*)tab;
        LANES({ for (int i = LANE; i < kHashBytes / 4; i += 64) t[i] = e0; })
    }
    LDS_FENCE();

    const int lastProbe  = n - kMfLimit + 1;      // mflimitPlusOne (lz4.c:963)
    const int matchLimit = n - kLastLiterals;     // lz4.c:964
    int nseq = 0, anchor = 0;

    if (n >= kMinLength) {
        int  insPos  = 0; bool hasIns = true;      // pending table insert ("First Byte" lz4.c:1005-1010; ip-2 lz4.c:1236-1242)
        int  rePos   = 0; bool hasRe  = false;     // pending immediate re-test at ip after a match (lz4.c:1255-1294)
        int  sBase   = 1; int sIter = 0;           // search started at sBase; next un-probed probe number
        int  width   = 16;                         // generic batches: 16 lanes first, 64 when a search drags on
        LV(Win20, Pn); int prefBase = -1;          // next window's bytes, requested one batch ahead
        LV(Win20, Pc);
        LANES({ for (int k = 0; k < 5; ++k) Pn[I_].w[k] = 0; Pc[I_] = Pn[I_]; })

        for (;;) {
            // ================================================================ GRID batch
            if (!U16 && sIter <= 64) {
                const int probeStart = hasRe ? rePos : sBase + sIter;
                const int firstPos   = hasIns ? insPos : probeStart;
                const int base       = firstPos & ~63;
                if (base >= 64 && base + 96 <= n) {
                    uint32_t* T = (uint32_t*)tab;
                    LV(int, act); LV(uint32_t, h); LV(uint32_t, r);
                    LV(uint32_t, ent); LV(uint32_t, rent);           // my table entry / the entry I displaced
                    LV(int, hit); LV(int, fwd); LV(int, eLane); LV(int, cand); LV(Win20, Cw);
                    // ---- 1. every lane: its window, the table exchange, the request for its candidate's window
                    LANES({
                        const int q = base + LANE;
                        const int isIns = hasIns && q == insPos;
                        act[I_] = isIns || q >= probeStart;
                        hit[I_] = 0; fwd[I_] = 0; r[I_] = 0; h[I_] = 0; ent[I_] = 0; rent[I_] = 0; cand[I_] = 0;
                        if (base != prefBase) Pn[I_] = load_win20(src, q);
                        Pc[I_] = Pn[I_];
                        if (act[I_]) {
                            h[I_] = seq_hash<false>(win20_seq(Pc[I_]));
                            const uint32_t tg = seq_tag(Pc[I_].w[0]) & tagMask;
                            ent[I_]  = ((uint32_t)q << sh) | tg;
                            rent[I_] = lds_max_rtn(&T[h[I_]], ent[I_]);
                            r[I_]    = rent[I_] >> sh;
                            cand[I_] = !isIns && r[I_] < (uint32_t)q && r[I_] + kMaxDist >= (uint32_t)q && (rent[I_] & tagMask) == tg;
                            if (cand[I_]) Cw[I_] = load_win20(src, (int)r[I_]);
                        }
                        if (base + 160 <= n) Pn[I_] = load_win20(src, q + 64);   // request the next window now, use it next batch
                    })
                    prefBase = (base + 160 <= n) ? base + 64 : -1;
                    LANES({
                        if (act[I_] && cand[I_] && Cw[I_].w[0] == Pc[I_].w[0]) {
                            hit[I_] = 1;
                            fwd[I_] = win20_fwd(Pc[I_], Cw[I_]);
                        }
                        eLane[I_] = LANE + kMinMatch + fwd[I_];     // lane index just past a match that starts here
                    })
                    // LDS atomics on one slot are expected to resolve in ascending lane order (then r is the
                    // nearest earlier twin or the pre-batch value).  Any other order shows up as r >= q somewhere.
                    const uint64_t misorder = BALLOT(act[I_] && r[I_] >= (uint32_t)(base + LANE));
                    if (misorder) {
                        LANES({ if (act[I_]) lds_min(&T[h[I_]], rent[I_]); })  // min over a slot's group == its pre-batch value
                        LDS_FENCE();
                        goto generic_batch;
                    }
                    {
                        uint64_t hits  = BALLOT(hit[I_]);
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
                        LV(int, stA);
                        // after a twin repair at lane b the walk is redone from b only: what it did below b does not depend on b
                        uint64_t keep = 0; int resume = -1;
                        for (;;) {
                            LV(int, nextHit);      // first recorded match at or after the end of the match that starts here (64: none)
                            {
                                const uint64_t hitsL = hits;
                                LANES({
                                    const uint64_t ah = (eLane[I_] < 64) ? (hitsL >> eLane[I_]) : 0;
                                    nextHit[I_] = ah ? eLane[I_] + ctz64(ah) : 64;
                                })
                            }
                            mm = keep; eL = 0; finished = false;
                            int w = 64;
                            {
                                const int start = resume >= 0 ? resume : cur0;     // lane b was a probe: the parser is searching there
                                if (start < 64) { const uint64_t hm = hits & (~0ull << start); if (hm) w = ctz64(hm); }
                            }
                            if (w < 64 && (mm != 0 || w <= lim0)) {               // (the stride limit concerns the first match only)
                                // Hops come four to a branch.  A hop is: mark the lane, fetch its successor.  Once the walk has ended
                                // (w == 64) the remaining hops of a group only touch bit 0 of the mask, which no hop but the very
                                // first can legitimately set.  Matches longer than the speculative window are not known to the hop: it
                                // walks through them as if they ended there, and the first one it touched is put right afterwards.
                                for (;;) {
                                    const uint64_t low = (mm & 1) | (w == 0 ? 1ull : 0ull);
                                    do {
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
                                    if (mc0 == 16) { mc0 += wave_common_len(src, p0 + 20, c0 + 20, matchLimit); WL(fwd, ws, mc0); }
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
                            Send = mm ? 64 : min_(64, lim0 + 1);
                            const uint64_t mmL = mm; const int SendL = Send;
                            // end of the last executed match below each lane: ends grow along the walk, so an exclusive
                            // prefix maximum over the match lanes (DPP, no LDS round trip) is that value
                            LANES({ stA[I_] = ((mmL >> LANE) & 1) ? eLane[I_] : 0; })
                            SCAN_MAX_EXCL(stA);
                            const uint64_t hasPm = BALLOT(stA[I_] > 0);
                            LANES({ stA[I_] = stA[I_] > 0 ? stA[I_] : cur0; })                // where probing resumed before me
                            const uint64_t probes = BALLOT(LANE >= stA[I_] && LANE < SendL && LANE >= cur0);
                            E = probes | insBit0 | (hasPm & BALLOT(stA[I_] == LANE + 2));      // + the ip-2 inserts (lz4.c:1236-1242)
                            if (twins & probes) {
                                // A probe whose candidate is an earlier lane of this batch is only right if that lane was executed.
                                // Otherwise the sequential parser saw what that lane displaced (or what *it* displaced, ...):
                                // repair the first such lane in place and redo the (cheap) hop.
                                const uint64_t EL = E;
                                const uint64_t bad = twins & probes & BALLOT(!((EL >> (((int)r[I_] - base) & 63)) & 1));
                                if (bad) {
                                    const int b = ctz64(bad);
                                    uint32_t ce = RL(rent, b);                          // entry lane b displaced
                                    int tl = -1;                                        // the lane whose displaced entry `ce` is
                                    for (;;) {
                                        const uint32_t ci = ce >> sh;
                                        if (ci < (uint32_t)firstPos) break;             // a pre-batch entry
                                        const int t = (int)ci - base;
                                        if ((E >> t) & 1) break;                        // an executed lane of this batch
                                        ce = RL(rent, t);                               // a skipped lane: what it displaced
                                        tl = t;
                                    }
                                    const uint32_t cp = ce >> sh, qb = (uint32_t)(base + b);
                                    int nhit = 0, nfwd = 0;
                                    if (cp + kMaxDist >= qb && (ce & tagMask) == (RL(ent, b) & tagMask)) {
                                        Win20 Pb; for (int k = 0; k < 5; ++k) Pb.w[k] = RLF(Pc, w[k], b);
                                        Win20 Cn;
                                        if (cp >= (uint32_t)base) { const int t = (int)cp - base; for (int k = 0; k < 5; ++k) Cn.w[k] = RLF(Pc, w[k], t); }
                                        // a pre-batch entry was displaced by lane tl, and if that lane took it for a candidate (a twin
                                        // usually repeats the very same bytes) its window already holds what is needed: no memory round trip
                                        else if (tl >= 0 && ((cwValid >> tl) & 1)) { for (int k = 0; k < 5; ++k) Cn.w[k] = RLF(Cw, w[k], tl); }
                                        else { Cn = load_win20(src, (int)cp); for (int k = 0; k < 5; ++k) Cn.w[k] = UNI(Cn.w[k]); }
                                        if (Cn.w[0] == Pb.w[0]) { nhit = 1; nfwd = win20_fwd(Pb, Cn); }
                                    }
                                    WL(rent, b, ce); WL(r, b, cp); WL(hit, b, nhit); WL(fwd, b, nfwd);
                                    WL(eLane, b, b + kMinMatch + nfwd);
                                    const uint64_t bit = 1ull << b;
                                    hits = nhit ? (hits | bit) : (hits & ~bit);
                                    specialLeft = (nhit && nfwd == 16) ? (specialLeft | bit) : (specialLeft & ~bit);
                                    twins &= ~bit;
                                    cwValid &= ~bit;                                    // lane b's Cw no longer belongs to its (new) rent
                                    keep = mm & (bit - 1); resume = b;
                                    continue;
                                }
                            }
                            break;
                        }
                        // lz4.c:1233: a match that ends at or past the last probe position ends the block.  Decided here, from the
                        // final walk: the pass that finished a long match may have been redone after a twin repair, and the redo
                        // sees that match as an ordinary one.
                        finished = (mm != 0) && (base + eL >= lastProbe);

                        // ---- 4. one record per executed match
                        if (mm) {
                            const uint64_t mmL = mm; const int at = nseq;
                            LANES({
                                if ((mmL >> LANE) & 1)
                                    seq[at + LANE_RANK(mmL)] = seq_pack((uint32_t)(base + LANE), (uint32_t)fwd[I_], (uint32_t)(base + LANE) - r[I_]);
                            })
                            nseq += __builtin_popcountll(mm);
                            anchor = base + eL;
                        }
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
                        if (anyTwins) { LDS_ORDER(); LANES({ if ((EL2 >> LANE) & 1) lds_max(&T[h[I_]], ent[I_]); }) }
                        LDS_ORDER();
                        width = 64;
                        continue;
                    }
                }
            }
generic_batch:
            // ================================================================ GENERIC batch: one lane per probe of the search loop
            {
            const int pre = (hasIns ? 1 : 0) + (hasRe ? 1 : 0);
            LV(int, q); LV(uint32_t, h); LV(uint32_t, old); LV(uint32_t, rb); LV(uint32_t, lo4);
            LV(uint32_t, ent); LV(uint32_t, oldE);
            LV(int, ok); LV(int, hit);

            // ---- positions + termination test (lz4.c:1051-1055): probe i happens only if pos+stride <= lastProbe
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

            // ---- hash, table read, speculative commit, read-back
            LANES({
                hit[I_] = 0;
                if (LANE < nproc) {
                    const uint64_t s8 = ld64u(src + q[I_]);
                    lo4[I_] = (uint32_t)s8;
                    h[I_] = seq_hash<U16>(s8);
                    ent[I_]  = ((uint32_t)q[I_] << sh) | (seq_tag(lo4[I_]) & tagMask);
                    oldE[I_] = tab_get<U16>(tab, h[I_]);
                    old[I_]  = oldE[I_] >> sh;
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
                    if ((U16 || old[I_] + kMaxDist >= cur) && ((oldE[I_] ^ ent[I_]) & tagMask) == 0)
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
                int used = keep;
                if (hasIns && used > 0) { hasIns = false; used--; }
                if (hasRe  && used > 0) { hasRe = false; used--; }
                sIter += used;
                if (keep == nproc && endInBatch && !hasIns && !hasRe) break;      // -> last literals (lz4.c:1055)
                width = 64;
                continue;
            }

            // ---- a match: winner lane w.  Its length (lz4.c:1182-1185), one record.
            const int w = keep - 1;
            const int p = RL(q, w);
            const int c = (int)RL(old, w);
            const int mc = wave_common_len(src, p + kMinMatch, c + kMinMatch, matchLimit);
            {
                const int at = nseq;
                LANES({ if (LANE == 0) seq[at] = seq_pack((uint32_t)p, (uint32_t)mc, (uint32_t)(p - c)); })
            }
            nseq++;
            const int ip = p + kMinMatch + mc;
            anchor = ip;
            if (ip >= lastProbe) break;                                        // lz4.c:1233

            // next batch: insert ip-2 (lz4.c:1236-1242), re-test ip (lz4.c:1255-1294), then search from ip+1 (lz4.c:1298)
            hasIns = true; insPos = ip - 2;
            hasRe = true;  rePos = ip;
            sBase = ip + 1; sIter = 0; width = 16;
            }
        }
    }
    *lastAnchor = anchor;
    return nseq;
}

// LZ4_compress_fast_extState's table choice (lz4.c:1389): byU16 below 64 KiB + 11
DEV int wave_parse_l1(const uint8_t* __restrict__ src, int n, void* tab, uint64_t* __restrict__ seq, int* lastAnchor)
{
    if (n < k64KLimit) return wave_parse_l1_tt<true>(src, n, tab, seq, lastAnchor);
    return wave_parse_l1_tt<false>(src, n, tab, seq, lastAnchor);
}

// ------------------------------------------------------------------------------------------ EMIT
// What one sequence is in the output, from its record and the record before it (one lane each).
struct SeqOut { int anchor, lit, mlen, extL, extM, size; uint32_t off; };
DEV SeqOut seq_measure(const uint8_t* __restrict__ src, uint64_t rec, uint64_t prev, bool first)
{
    SeqOut o;
    const int pos = (int)seq_pos(rec), fwd = (int)seq_fwd(rec);
    o.off = seq_off(rec);
    o.anchor = first ? 0 : (int)seq_pos(prev) + kMinMatch + (int)seq_fwd(prev);
    // catch-up (lz4.c:1105-1109): while ip > anchor && match > lowLimit && ip[-1] == match[-1].  A re-test has ip == anchor.
    const int cnd = pos - (int)o.off;
    const int maxBack = min_(pos - o.anchor, cnd);
    int bk = 0;
    bool open = true;
    while (open && bk + 4 <= maxBack) {
        const uint32_t x = ld32u(src + pos - 4 - bk) ^ ld32u(src + cnd - 4 - bk);
        if (x) { bk += __builtin_clz(x) >> 3; open = false; } else bk += 4;
    }
    while (open && bk < maxBack && src[pos - 1 - bk] == src[cnd - 1 - bk]) ++bk;
    o.lit  = pos - bk - o.anchor;
    o.mlen = fwd + bk;
    o.extL = o.lit  >= 15 ? (o.lit  - 15) / 255 + 1 : 0;
    o.extM = o.mlen >= 15 ? (o.mlen - 15) / 255 + 1 : 0;
    o.size = 1 + o.extL + o.lit + 2 + o.extM;
    return o;
}

// bytes of the sequences [c*kSeqChunk, min(nseq, (c+1)*kSeqChunk)) of a block
DEV uint32_t seq_emit_sizes(const uint8_t* __restrict__ src, const uint64_t* __restrict__ seq, int nseq, int c)
{
    const int i0 = c * kSeqChunk, i1 = min_(nseq, i0 + kSeqChunk);
    LV(int, acc);
    LANES({ acc[I_] = 0; })
    for (int i = i0; i < i1; i += 64) {
        LANES({
            const int k = i + LANE;
            if (k < i1) acc[I_] += seq_measure(src, seq[k], k ? seq[k - 1] : 0, k == 0).size;
        })
    }
    SCAN_INCL(acc);
    return (uint32_t)RL(acc, 63);
}

// Exclusive scan of a block's chunk sizes + the last literals (lz4.c:1302-1329) + the limitedOutput verdict (see the top of this
// file).  chunkOff[c] <- bytes before chunk c.  Returns the block's compressed size, 0 when liblz4 returns 0.
DEV int seq_emit_scan(const uint32_t* __restrict__ chunkBytes, uint32_t* __restrict__ chunkOff, int nseq, int lastAnchor, int n, int cap)
{
    const int nChunks = (nseq + kSeqChunk - 1) / kSeqChunk;
    int run = 0;
    for (int c0 = 0; c0 < nChunks; c0 += 64) {
        LV(int, v);
        LANES({ v[I_] = (c0 + LANE < nChunks) ? (int)chunkBytes[c0 + LANE] : 0; })
        LV(int, s);
        LANES({ s[I_] = v[I_]; })
        SCAN_INCL(s);
        const int runL = run;
        LANES({ if (c0 + LANE < nChunks) chunkOff[c0 + LANE] = (uint32_t)(runL + s[I_] - v[I_]); })
        run += RL(s, 63);
    }
    const int last = n - lastAnchor;
    const int64_t total = (int64_t)run + 1 + (last >= 15 ? (last - 15) / 255 + 1 : 0) + last;
    const bool limited = !(cap >= compress_bound(n));                       // lz4.c:1388
    if (limited && total > (int64_t)cap) return 0;
    return (int)total;
}

// length bytes behind a token by one lane: (len - 15) as 0xFF... and a final byte (lz4.c:1123-1128, :1213-1223)
DEV int lane_len_ext(uint8_t* dst, int o, int rest) { for (; rest >= 255; rest -= 255) dst[o++] = 255; dst[o++] = (uint8_t)rest; return o; }
// exact-length copy by one lane (the bytes next to it belong to other lanes)
DEV void lane_copy(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, int len)
{
    int i = 0;
    for (; i + 16 <= len; i += 16) *(v16u_t*)(d + i) = *(const v16u_t*)(s + i);
    if (len & 8) { st64u(d + i, ld64u(s + i)); i += 8; }
    if (len & 4) { st32u(d + i, ld32u(s + i)); i += 4; }
    if (len & 2) { st16u(d + i, ld16u(s + i)); i += 2; }
    if (len & 1) d[i] = s[i];
}

// Writes chunk c of a block at dst + chunkOff (token, literal length bytes, literals, offset, match length bytes per sequence:
// lz4.c:1112-1226); the wave that writes the last chunk (or chunk 0 of a block without sequences) also writes the last
// literals (lz4.c:1302-1329).  Long literal runs and long length-byte runs are left to the whole wave.
DEV void seq_emit_write(const uint8_t* __restrict__ src, int n, const uint64_t* __restrict__ seq, int nseq, int lastAnchor, int c,
                        uint32_t chunkOff, uint8_t* __restrict__ dst)
{
    const int i0 = c * kSeqChunk, i1 = min_(nseq, i0 + kSeqChunk);
    int op = (int)chunkOff;
    for (int i = i0; i < i1; i += 64) {
        LV(SeqOut, so); LV(int, tok);
        LANES({
            const int k = i + LANE;
            if (k < i1) so[I_] = seq_measure(src, seq[k], k ? seq[k - 1] : 0, k == 0);
            else { so[I_].anchor = 0; so[I_].lit = 0; so[I_].mlen = 0; so[I_].extL = 0; so[I_].extM = 0; so[I_].size = 0; so[I_].off = 0; }
            tok[I_] = so[I_].size;
        })
        SCAN_INCL(tok);
        const int opL = op;
        LANES({ tok[I_] = opL + tok[I_] - so[I_].size; })
        op = RL(tok, 63) + RLF(so, size, 63);
        LANES({
            if (i + LANE < i1) {
                const SeqOut& s = so[I_];
                int o = tok[I_];
                dst[o++] = (uint8_t)((min_(s.lit, 15) << 4) | min_(s.mlen, 15));
                if (s.extL) { if (s.extL <= 8) o = lane_len_ext(dst, o, s.lit - 15); else o += s.extL; }
                if (s.lit <= 64) lane_copy(dst + o, src + s.anchor, s.lit);
                o += s.lit;
                st16u(dst + o, (uint16_t)s.off);
                o += 2;
                if (s.extM) { if (s.extM <= 8) lane_len_ext(dst, o, s.mlen - 15); }
            }
        })
        // the long ones, one at a time by the whole wave
        uint64_t big = BALLOT(i + LANE < i1 && (so[I_].lit > 64 || so[I_].extL > 8 || so[I_].extM > 8));
        for (; big; big &= big - 1) {
            const int w = ctz64(big);
            const int t = RL(tok, w), lit = RLF(so, lit, w), mlen = RLF(so, mlen, w), eL = RLF(so, extL, w), eM = RLF(so, extM, w);
            if (eL > 8) emit_len_ext(dst, t + 1, lit - 15);
            if (lit > 64) wave_copy(dst + t + 1 + eL, src + RLF(so, anchor, w), lit);
            if (eM > 8) emit_len_ext(dst, t + 1 + eL + lit + 2, mlen - 15);
        }
    }
    const int nChunks = (nseq + kSeqChunk - 1) / kSeqChunk;
    if (c == (nChunks ? nChunks - 1 : 0)) {
        const int last = n - lastAnchor;
        if (last >= 15) {
            LANES({ if (LANE == 0) dst[op] = 0xF0; })
            op = emit_len_ext(dst, op + 1, last - 15);
        } else {
            LANES({ if (LANE == 0) dst[op] = (uint8_t)(last << 4); })
            op++;
        }
        wave_copy(dst + op, src + lastAnchor, last);
    }
}

}  // namespace plz4
