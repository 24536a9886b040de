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
enum : int { kSeqEngineFailed = -2 };     // SeqInfo::nseq of a block whose parser had no input to trust (-1: not sized for; both: no records)

// entries a block of n bytes can need: every sequence consumes at least MINMATCH input bytes.  The parser's array has one more
// (a dump entry at index seq_capacity(n)).
DEV int seq_capacity(int n) { return n / kMinMatch + 2; }
static inline int seq_capacity_host(int n) { return n / 4 + 2; }

DEV uint64_t seq_pack(uint32_t pos, uint32_t fwd, uint32_t off) { return (uint64_t)pos | ((uint64_t)fwd << 22) | ((uint64_t)off << 44); }
DEV uint32_t seq_pos(uint64_t r) { return (uint32_t)r & 0x3FFFFFu; }
DEV uint32_t seq_fwd(uint64_t r) { return (uint32_t)(r >> 22) & 0x3FFFFFu; }
DEV uint32_t seq_off(uint64_t r) { return (uint32_t)(r >> 44) & 0xFFFFu; }

// ((1 << width) - 1) << offset, width and offset below 64 (s_bfm_b64)
#define BFM64(width, offset) ((((uint64_t)1 << ((width) & 63)) - 1ull) << ((offset) & 63))
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
// equal bytes of [x+4, x+20) in two windows: 0..16.  Arithmetic only: on this machine a divergent `if` -- the compiler's
// s_and_saveexec + s_cbranch_execz skeleton, taken or not -- costs about 45 cycles (scripts/micro/execz.hip), a select 4.
DEV uint32_t eq_bytes(uint32_t x)                 // leading equal bytes of a dword pair from their XOR: 0..4
{
    return (x ? (uint32_t)__builtin_ctz(x) : 32u) >> 3;                   // (v_ffbl_b32 + v_min_u32 32: no branch)
}
DEV uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t bytes)      // bytes [bytes, bytes + 4) of the pair (lo, hi): v_alignbyte_b32
{
#if defined(PLZ4_EMU)
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (bytes & 3)));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, bytes);
#endif
}
DEV int win20_fwd(const Win20& p, const Win20& c)
{
    const uint32_t b1 = eq_bytes(p.w[1] ^ c.w[1]), b2 = eq_bytes(p.w[2] ^ c.w[2]), b3 = eq_bytes(p.w[3] ^ c.w[3]), b4 = eq_bytes(p.w[4] ^ c.w[4]);
    const uint32_t m1 = (b1 == 4u), m2 = m1 & (b2 == 4u), m3 = m2 & (b3 == 4u);
    return (int)(b1 + (m1 ? b2 : 0u) + (m2 ? b3 : 0u) + (m3 ? b4 : 0u));
}

// ------------------------------------------------------------------------------------------ PARSE
// Plays LZ4_compress_generic(noDict, byU16 / byU32, accel 1) over src[0, n), n <= 4 MiB, and writes one record per sequence to
// seq[] (at most seq_capacity(n)).  Returns their number; *lastAnchor = where the last literals start (lz4.c:1302).
// `tab` = 16 KiB of LDS owned by this wave.  Two kinds of batch as in wave_encode_block_tt (lz4_device.inl), same parser state
// between them; see there for the grid batch's table protocol (tagged entries, atomic-max commit, min / max patch).
//
// What is new here is how a grid batch gets its candidates' bytes.  The chain  table -> candidate address -> memory -> compare ->
// walk -> table patch -> next batch's table read  has a memory round trip in it, and nothing to do meanwhile now that the
// writer is gone.  So the batches run as a pipeline of three stages (batch k-1, k, k+1), each with its window P, its hash /
// entry, and a PEEK: at the top of batch k -- the table is exact then: batch k-1 is patched, batch k not yet committed --
// every lane of batch k+1 reads its slot (a plain read) and requests the 20 bytes at that entry's position.  When batch k+1
// commits, one iteration later, the entry it displaces is either the peeked one (bytes already here), or a lane of batch k that
// the parser executed, or an earlier lane of batch k+1 itself -- and those two have their windows in registers (prev.P,
// cur.P): one ds_bpermute round each.  Nothing waits for memory except a cold start (block head, after a generic batch or a
// long match).  Windows are requested two batches ahead into the stage batch k-1 no longer needs; the three stages rotate
// by unrolling (grid(S2,S0,S1), grid(S0,S1,S2), grid(S1,S2,S0)), not by copies.
#if defined(PLZ4_EMU)
#define SHFLF(x, f, l) ((x)[(l) & 63].f)
unsigned long long plz4_emu_cnt[8];               // test diagnostics: [0] grid batches, [1] primes, [2] atomics out of order, [3] second rounds for a long match alone, [4] long matches measured, [5] batches whose commit
                                                  // returned another entry to a probe, [6] batches with a second round, [7] entries outside the registers
#define EMU_CNT(i, v) (plz4_emu_cnt[(i)] += (unsigned long long)(v))
#else
#define SHFLF(x, f, l) plz4_bpermute((x)[0].f, (l))
#define EMU_CNT(i, v) do {} while (0)
#endif
struct GStage { Win20 P; Win20 C; uint32_t h, ent, pk; };     // window, candidate window, slot, own entry, peeked entry
// diagnostics build (-DPLZ4_STATS, scripts/stats_probe_l1.py): cycle stamps around the sections of a grid batch, each section
// closed by an explicit wait so that its stamp owns the latency it exposes
#if defined(PLZ4_STATS) && !defined(PLZ4_EMU)
#define STAT_WAIT_VM()   __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define STAT_WAIT_LGKM() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define STAT_WAIT_VM()   do {} while (0)
#define STAT_WAIT_LGKM() do {} while (0)
#endif
enum { P_BATCH = 0, P_CYC_MEM, P_CYC_LDS, P_CYC_REFRESH, P_CYC_CMP, P_CYC_WALK, P_CYC_TAIL, P_REPAIR, P_PRIME, P_GENERIC, P_CYC_TOTAL, P_BLOCKS, P_SEQ, P_CYC_GEN, P_CYC_NH, P_CYC_HOP, P_CYC_E, P_CYC_SLOW, P_HOPS };

// kLdsWin: the windows of the batches ahead come through `scr` (256 bytes of LDS owned by this wave) instead of one 20-byte load per
// lane: the 64 windows of a batch are 83 consecutive bytes, but 64 overlapping per-lane loads are 128 accesses to the CU's vector
// cache for them (its tag pipeline is what ten parser waves per CU saturate: TA busy 65 %, TCP stalled on pending misses 46 % of
// the time, profiles/r04b_ta_counters.txt).  The lanes fetch consecutive dwords a batch earlier (one coalesced load), park them in
// LDS and every lane reads its 20 bytes back from its byte offset.
template <bool U16, int kLdsWin = 0>
DEV int wave_parse_l1_tt(const uint8_t* __restrict__ src, const int n, void* tab, uint64_t* __restrict__ seq, int* lastAnchor, uint8_t* scr = nullptr)
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
    const int seqDump    = seq_capacity(n);       // one entry behind the records: where lanes without a record store
    int nseq = 0, anchor = 0;
    STAT_DECL;
    const unsigned long long tBlock0 = STAT_NOW(); (void)tBlock0;

    if (n >= kMinLength) {
        int  insPos  = 0; bool hasIns = true;      // pending table insert ("First Byte" lz4.c:1005-1010; ip-2 lz4.c:1236-1242)
        int  rePos   = 0; bool hasRe  = false;     // pending immediate re-test at ip after a match (lz4.c:1255-1294)
        int  sBase   = 1; int sIter = 0;           // search started at sBase; next un-probed probe number
        int  width   = 16;                         // generic batches: 16 lanes first, 64 when a search drags on
        // the pipeline's three stages; which one is batch k-1 / k / k+1 rotates with the unrolled loop below
        LV(GStage, S0); LV(GStage, S1); LV(GStage, S2);
        LV(uint32_t, pf);                              // kLdsWin: dword LANE of [base + 128, base + 384), requested a batch ago
        LANES({ pf[I_] = 0; })
        LANES({ for (int k = 0; k < 5; ++k) { S0[I_].P.w[k] = 0; S0[I_].C.w[k] = 0; } S0[I_].h = 0; S0[I_].ent = 0; S0[I_].pk = 0; S1[I_] = S0[I_]; S2[I_] = S0[I_]; })
        enum { kGridNext = 0, kGridDone = 1, kGridGeneric = 2, kGridStop = 3 };

        // One batch of the steady state: WALK, then COMMIT, then VERIFY.  On entry: cur.P = the window of `base`, cur.h / ent = its
        // slot and entry, cur.pk = what the slot held at the top of the previous batch, cur.C = the window at that entry's position
        // (requested a batch ago), next.P = the window of base + 64 (requested a batch ago), prev.P = the window of base - 64
        // when the batch before was a grid batch.
        //   walk    every lane is compared with its peeked candidate; a scalar hop over the hits plays the parser from the
        //           first probe lane on and the executed lanes (probes + the ip-2 inserts) follow per lane.  No table traffic.
        //   commit  ONLY the executed lanes exchange their entry with the table (one atomic max each, lanes of one slot in lane
        //           order): the table never holds a position the parser did not insert, so there is nothing to patch
        //           afterwards, and what a lane gets back is what the sequential parser read there -- the slot's content from
        //           before the batch, or the nearest executed lane below it.
        //   verify  a probe that got back the entry it had assumed is done.  One that did not (its slot was taken by an executed
        //           lane of this batch or of the one before, after the peek) takes the returned entry as its candidate -- that
        //           window is in the registers of this stage or the previous one, a ds_bpermute away -- and is compared again; if
        //           neither hit nor length of any probe changed, the walk stands (only offsets moved).  Else the commits are
        //           taken back (min with what each displaced restores the slots) and the round is repeated on the new
        //           candidates.  The sequential parse is the one consistent outcome (a lane depends on lower lanes only), so the
        //           first round that verifies is it.  Four rounds without agreement, or an entry whose window is not in the
        //           registers: commits taken back, the generic batch takes over.
        // The loads are unconditional, so the number of memory operations in flight at any point of the loop is fixed.
        auto grid = [&](LVREF(GStage, prev), LVREF(GStage, cur), LVREF(GStage, next), const int base) -> int {
            const int probeStart = hasRe ? rePos : sBase + sIter;
            const int firstPos   = hasIns ? insPos : probeStart;
            if ((sIter > 64) | ((firstPos & ~63) != base) | (base + 224 > n)) return kGridStop;
            uint32_t* T = (uint32_t*)tab;
            EMU_CNT(0, 1); STAT(P_BATCH, 1);
            const unsigned long long ts0 = STAT_NOW(); (void)ts0;
            STAT_WAIT_VM();
            const unsigned long long ts1 = STAT_NOW(); (void)ts1;
            STAT(P_CYC_MEM, ts1 - ts0);
            // ---- 1. peek for batch k+1 (the table holds exactly the parser's inserts up to this batch)
            LANES({
                next[I_].h   = seq_hash<false>(win20_seq(next[I_].P));
                next[I_].ent = ((uint32_t)(base + 64 + LANE) << sh) | (seq_tag(next[I_].P.w[0]) & tagMask);
                next[I_].pk  = T[next[I_].h];
            })
            LDS_ORDER();
            // ---- 2. this batch against its peeked candidates
            LV(uint32_t, ce); LV(uint32_t, rent); LV(bool, hit); LV(int, fwd); LV(int, eLane);
            auto compare = [&](LVREF(Win20, W), LVREF(bool, hitV), LVREF(int, fwdV)) {
                LANES({
                    const uint32_t q = (uint32_t)(base + LANE), cp = ce[I_] >> sh;
                    const bool ok = (cp < q) & (cp + kMaxDist >= q) & (((ce[I_] ^ cur[I_].ent) & tagMask) == 0) & (W[I_].w[0] == cur[I_].P.w[0]);
                    const int f = win20_fwd(cur[I_].P, W[I_]);
                    hitV[I_] = ok;
                    fwdV[I_] = ok ? f : 0;
                })
            };
            LV(Win20, W);
            LANES({ ce[I_] = cur[I_].pk; rent[I_] = 0; W[I_] = cur[I_].C; })
            compare(W, hit, fwd);
            LANES({ eLane[I_] = LANE + kMinMatch + fwd[I_]; })    // lane index just past a match that starts here
            STAT_WAIT_LGKM();
            const unsigned long long ts2 = STAT_NOW(); (void)ts2;
            STAT(P_CYC_LDS, ts2 - ts1);
            // ---- 3. requests for the batches to come: the peeked candidates of batch k+1 (every lane loads: its candidate, or
            // its own position when it has none), the window of batch k+2 into the stage batch k-1 is done with
            LANES({
                const uint32_t q1 = (uint32_t)(base + 64 + LANE), pr = next[I_].pk >> sh;
                const bool pc = (pr < q1) & (pr + kMaxDist >= q1) & (((next[I_].pk ^ next[I_].ent) & tagMask) == 0);
                next[I_].C = load_win20(src, (int)(pc ? pr : q1));
                if (!kLdsWin) prev[I_].P = load_win20(src, base + 128 + LANE);
            })
            if (kLdsWin == 2) {
                // (no LDS to spare -- ten tables fill the CU: the same through the lane-exchange network, six dwords from the lanes
                // that hold them and a byte alignment each)
                LV(uint32_t, g0); LV(uint32_t, g1); LV(uint32_t, g2); LV(uint32_t, g3); LV(uint32_t, g4); LV(uint32_t, g5);
                LANES({ const int j = LANE >> 2; g0[I_] = SHFL(pf, j); g1[I_] = SHFL(pf, j + 1); g2[I_] = SHFL(pf, j + 2); g3[I_] = SHFL(pf, j + 3); g4[I_] = SHFL(pf, j + 4); g5[I_] = SHFL(pf, j + 5); })
                LANES({
                    const uint32_t sft = (uint32_t)(LANE & 3);
                    prev[I_].P.w[0] = align_bytes(g1[I_], g0[I_], sft); prev[I_].P.w[1] = align_bytes(g2[I_], g1[I_], sft);
                    prev[I_].P.w[2] = align_bytes(g3[I_], g2[I_], sft); prev[I_].P.w[3] = align_bytes(g4[I_], g3[I_], sft);
                    prev[I_].P.w[4] = align_bytes(g5[I_], g4[I_], sft);
                })
                LANES({ pf[I_] = ld32u(src + min_(base + 192 + 4 * LANE, n - 4)); })
            } else if (kLdsWin) {
                // the window of batch k+2 out of the dwords requested a batch ago; the dwords of batch k+3 on their way
                LANES({ ((uint32_t*)scr)[LANE] = pf[I_]; })
                LDS_ORDER();
                LANES({
                    const v16u_t a = *(const v16u_t*)(scr + LANE);
                    prev[I_].P.w[0] = a.w[0]; prev[I_].P.w[1] = a.w[1]; prev[I_].P.w[2] = a.w[2]; prev[I_].P.w[3] = a.w[3];
                    prev[I_].P.w[4] = ld32u(scr + LANE + 16);
                })
                LDS_ORDER();
                LANES({ pf[I_] = ld32u(src + min_(base + 192 + 4 * LANE, n - 4)); })
            }
            const unsigned long long ts3 = STAT_NOW(); (void)ts3;
            STAT(P_CYC_CMP, ts3 - ts2);

            const int  cur0 = probeStart - base;                  // first probe lane (>= 64: none in this batch, only the pending insert)
            const bool re0  = hasRe;
            const uint64_t insBit0 = hasIns ? (1ull << (insPos - base)) : 0;
            int lim0 = sBase + 65 - base; if (lim0 > 63) lim0 = 63;   // probe number <= 65 keeps the stride at 1
            uint64_t mm = 0;           // executed match lanes
            int      eL = 0;           // end lane of the last executed match
            uint64_t E = 0;            // executed lanes (probes + inserts)
            uint64_t probes = 0;
            int      Send = 0;
            bool     finished = false;
            // the parser over this batch's hits: scalar hop over the recorded matches only, everything else derived per lane
            auto walk = [&]() {
                const uint64_t hits = BALLOT(hit[I_]);
                uint64_t specialLeft = hits & BALLOT(fwd[I_] == 16);    // longer than the speculative window: the hop needs its end
                LV(int, nextHit);      // first recorded match at or after the end of the match that starts here (64: none)
                LANES({
                    const uint64_t ah = (eLane[I_] < 64) ? (hits >> eLane[I_]) : 0;
                    nextHit[I_] = ah ? eLane[I_] + ctz64(ah) : 64;
                })
                mm = 0; eL = 0; finished = false;
                int w = 64;
                if (cur0 < 64) { const uint64_t hm = hits & (~0ull << cur0); if (hm) w = ctz64(hm); }
                if (w > lim0) w = 64;                                  // (the stride limit concerns the first match only)
                const unsigned long long tw1 = STAT_NOW(); (void)tw1;
                if (w < 64) {
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
                            STAT(P_HOPS, 4);
                        } while (w < 64);
                        mm = (mm & ~1ull) | low;
                        const uint64_t sp = mm & specialLeft;
                        if (!sp) break;
                        EMU_CNT(4, 1);
                        const int ws = ctz64(sp);
                        mm &= (2ull << ws) - 1;                       // what the walk did after it is void
                        const int p0 = base + ws, c0 = (int)(RL(ce, ws) >> sh);
                        const int mc0 = 16 + wave_common_len(src, p0 + 20, c0 + 20, matchLimit);
                        WL(fwd, ws, mc0);
                        specialLeft &= ~(1ull << ws);
                        const int e1 = ws + kMinMatch + mc0;
                        WL(eLane, ws, e1);
                        if (base + e1 >= lastProbe) { finished = true; break; }   // lz4.c:1233 (only a long match gets there)
                        const uint64_t hm = (e1 < 64) ? (hits & (~0ull << e1)) : 0;
                        w = hm ? ctz64(hm) : 64;
                        if (w >= 64) break;
                    }
                    eL = RL(eLane, 63 - __builtin_clzll(mm));
                }
                const unsigned long long tw2 = STAT_NOW(); (void)tw2;
                STAT(P_CYC_HOP, tw2 - tw1);
                // which lanes did the sequential parser execute: the end of the last executed match below each lane (ends grow
                // along the walk, so an exclusive prefix maximum over the match lanes is that value)
                Send = mm ? 64 : min_(64, lim0 + 1);
                const uint64_t mmL = mm; const int SendL = Send;
                LV(int, stA);
                LANES({ stA[I_] = LANE_IN(mmL) ? eLane[I_] : 0; })
                SCAN_MAX_EXCL(stA);
                const uint64_t hasPm = BALLOT(stA[I_] > 0);
                LANES({ stA[I_] = stA[I_] > 0 ? stA[I_] : cur0; })                // where probing resumed before me
                probes = BALLOT(LANE >= stA[I_] && LANE < SendL && LANE >= cur0);
                E = probes | insBit0 | (hasPm & BALLOT(stA[I_] == LANE + 2));      // + the ip-2 inserts (lz4.c:1236-1242)
                STAT(P_CYC_E, STAT_NOW() - tw2);
            };
            // the same walk without a branch in it, for the first round: eight hops unconditionally (a hop past the end only
            // touches bit 0 again), more only if the batch has more matches; matches longer than the speculative window are
            // taken at that length and reported by the caller's check (a uniform branch costs this machine 80-90 cycles, taken
            // or not: scripts/micro/branch.hip -- the steady state is written to have a handful of them per batch, not forty)
            auto walk_fast = [&](const uint64_t hits) {
                LV(int, nextHit);
                LANES({
                    const uint64_t ah = (eLane[I_] < 64) ? (hits >> eLane[I_]) : 0;
                    nextHit[I_] = ah ? eLane[I_] + ctz64(ah) : 64;
                })
                const uint64_t hm = (cur0 < 64) ? (hits & (~0ull << (cur0 & 63))) : 0;
                int w = hm ? ctz64(hm) : 64;
                w = (w > lim0) ? 64 : w;                               // (the stride limit concerns the first match only)
                const uint64_t low = (w == 0 ? 1ull : 0ull);
                const bool any = w < 64;
                uint64_t m = 0;
                const unsigned long long tw1 = STAT_NOW(); (void)tw1;
                for (int u = 0; u < 6; ++u) {                           // (4 .. 8 unconditional hops measure the same on text)
                    const int n1 = RL(nextHit, w & 63);
                    m |= 1ull << (w & 63);
                    w = (w < 64) ? n1 : 64;
                }
                while (w < 64) {
                    for (int u = 0; u < 4; ++u) {
                        const int n1 = RL(nextHit, w & 63);
                        m |= 1ull << (w & 63);
                        w = (w < 64) ? n1 : 64;
                    }
                }
                STAT(P_HOPS, 8);
                mm = any ? ((m & ~1ull) | low) : 0;
                eL = RL(eLane, (63 - __builtin_clzll(mm | 1ull)) & 63);
                eL = mm ? eL : 0;
                finished = false;
                const unsigned long long tw2 = STAT_NOW(); (void)tw2;
                STAT(P_CYC_HOP, tw2 - tw1);
                Send = mm ? 64 : min_(64, lim0 + 1);
                const uint64_t mmL = mm; const int SendL = Send;
                LV(int, stA);
                LANES({ stA[I_] = LANE_IN(mmL) ? eLane[I_] : 0; })
                SCAN_MAX_EXCL(stA);
                const uint64_t hasPm = BALLOT(stA[I_] > 0);
                LANES({ stA[I_] = stA[I_] > 0 ? stA[I_] : cur0; })
                probes = BALLOT((LANE >= stA[I_]) & (LANE < SendL) & (LANE >= cur0));
                E = probes | insBit0 | (hasPm & BALLOT(stA[I_] == LANE + 2));
                STAT(P_CYC_E, STAT_NOW() - tw2);
            };
            // ---- 4. first round, straight: walk -> commit -> verify, every step unconditional, ONE question at the end
            uint64_t committed = 0;
            bool giveUp = false;
            {
                const uint64_t hits1 = BALLOT(hit[I_]);
                const uint64_t special1 = hits1 & BALLOT(fwd[I_] == 16);
                walk_fast(hits1);
                const unsigned long long tc0 = STAT_NOW(); (void)tc0;
                const uint64_t EL = E;
                // (a lane that is not executed exchanges a 0: the slot stays as it is)
                LANES({ rent[I_] = lds_max_rtn(&T[cur[I_].h], LANE_IN(EL) ? cur[I_].ent : 0u); })
                committed = EL;
                const uint64_t upd = probes & BALLOT(rent[I_] != ce[I_]);
                // (ascending lane order of the atomics on one slot is what makes the returned entry the sequential one: any other
                // order shows up as a position at or above the lane's own)
                const uint64_t misorder = EL & BALLOT((rent[I_] >> sh) >= (uint32_t)(base + LANE));
                const uint64_t noRegs   = upd & BALLOT(rent[I_] != cur[I_].pk && (rent[I_] >> sh) < (uint32_t)base);
                LV(bool, hitN); LV(int, fwdN); LV(Win20, Wn);
                uint64_t diff = 0;
                if (upd) {                   // (about half of the batches on text: worth the one conditional)
                    LANES({
                        const bool u = LANE_IN(upd);
                        ce[I_] = u ? rent[I_] : ce[I_];
                        const int t = (int)(ce[I_] >> sh) - base;               // the lane of this batch whose position that is
                        const bool same = ce[I_] == cur[I_].pk;
                        for (int k = 0; k < 5; ++k) {
                            const uint32_t a = SHFLF(cur, P.w[k], t);
                            Wn[I_].w[k] = same ? cur[I_].C.w[k] : a;
                        }
                    })
                    compare(Wn, hitN, fwdN);
                    // (a window that compares equal to its end says nothing about the length behind it: that one is measured again)
                    diff = upd & BALLOT((hitN[I_] != hit[I_]) | (fwdN[I_] != fwd[I_]) | (fwdN[I_] == 16));
                    LANES({
                        const bool u = LANE_IN(upd);
                        hit[I_] = u ? hitN[I_] : hit[I_]; fwd[I_] = u ? fwdN[I_] : fwd[I_]; eLane[I_] = LANE + kMinMatch + fwd[I_];
                    })
                }
                STAT(P_CYC_REFRESH, STAT_NOW() - tc0);
                EMU_CNT(5, upd != 0);
                const uint64_t trouble = misorder | noRegs | diff | (mm & special1);
                if (trouble) {
                    // ---- rounds of walk -> commit -> verify (the walk with the long matches measured) until one verifies
                    EMU_CNT(6, 1); STAT(P_REPAIR, 1);
                    EMU_CNT(3, (diff | misorder | noRegs) == 0);                       // (only because of a match longer than the window)
                    if (misorder | noRegs) { EMU_CNT(2, misorder != 0); EMU_CNT(7, noRegs != 0); giveUp = true; }
                    for (int round = 1; !giveUp; ++round) {
                        {                                                      // take the last round's commits back
                            const uint64_t cm = committed;
                            LANES({ if (LANE_IN(cm)) lds_min(&T[cur[I_].h], rent[I_]); })   // min over a slot's group == its pre-batch value
                            LDS_ORDER();
                            committed = 0;
                        }
                        if (round == 4) { giveUp = true; break; }
                        walk();
                        const uint64_t EL2 = E;
                        LANES({ if (LANE_IN(EL2)) rent[I_] = lds_max_rtn(&T[cur[I_].h], cur[I_].ent); })
                        committed = EL2;
                        const uint64_t upd2 = probes & BALLOT(rent[I_] != ce[I_]);
                        if (EL2 & BALLOT((rent[I_] >> sh) >= (uint32_t)(base + LANE))) { EMU_CNT(2, 1); giveUp = true; break; }
                        if (!upd2) break;                                      // every probe read what it had assumed
                        if (upd2 & BALLOT(rent[I_] != cur[I_].pk && (rent[I_] >> sh) < (uint32_t)base)) { EMU_CNT(7, 1); giveUp = true; break; }
                        LANES({
                            if (LANE_IN(upd2)) ce[I_] = rent[I_];
                            const int t = (int)(ce[I_] >> sh) - base;
                            const bool same = ce[I_] == cur[I_].pk;
                            for (int k = 0; k < 5; ++k) {
                                const uint32_t a = SHFLF(cur, P.w[k], t);
                                Wn[I_].w[k] = same ? cur[I_].C.w[k] : a;
                            }
                        })
                        compare(Wn, hitN, fwdN);
                        const uint64_t diff2 = upd2 & BALLOT(hitN[I_] != hit[I_] || fwdN[I_] != fwd[I_] || fwdN[I_] == 16);
                        if (!diff2) break;                                     // same walk, same executed set: only offsets moved
                        LANES({ if (LANE_IN(upd2)) { hit[I_] = hitN[I_]; fwd[I_] = fwdN[I_]; eLane[I_] = LANE + kMinMatch + fwdN[I_]; } })
                    }
                }
            }
            if (giveUp) {
                if (committed) {
                    const uint64_t cm = committed;
                    LANES({ if (LANE_IN(cm)) lds_min(&T[cur[I_].h], rent[I_]); })
                }
                LDS_FENCE();
                return kGridGeneric;
            }
            // ---- 4b. batch k+1's slots once more: the table holds this batch's inserts now, and a slot that changed since the peek
            // was taken by an executed lane of THIS batch -- whose window is here.  Its bytes replace the requested ones, so the
            // next batch starts from candidates that are exact against everything before it; what its commit can still return
            // differently is a lane of its own.
            LANES({
                const uint32_t late = T[next[I_].h];
                const bool ch = late != next[I_].pk;
                const int t = (int)(late >> sh) - base;
                for (int k = 0; k < 5; ++k) {
                    const uint32_t x = SHFLF(cur, P.w[k], t);
                    next[I_].C.w[k] = ch ? x : next[I_].C.w[k];
                }
                next[I_].pk = late;
            })
            const unsigned long long ts5 = STAT_NOW(); (void)ts5;
            STAT(P_CYC_WALK, ts5 - ts3);

            // ---- 5. one record per executed match
            {
                // (every lane stores: the ones without a match into the dump entry behind the block's last possible record)
                const uint64_t mmL = mm; const int at = nseq;
                LANES({
                    const int slot = LANE_IN(mmL) ? at + LANE_RANK(mmL) : seqDump;
                    seq[slot] = seq_pack((uint32_t)(base + LANE), (uint32_t)fwd[I_], (uint32_t)(base + LANE) - (ce[I_] >> sh));
                })
                nseq += __builtin_popcountll(mm);
                anchor = mm ? base + eL : anchor;
            }
            if (finished) return kGridDone;

            // ---- 6. parser state after this batch
            {   // (selects, not branches)
                const bool hm = mm != 0, c0 = cur0 < 64;
                const bool reM = eL >= Send;                                // with a match: its re-test is not executed in this batch
                const bool reN = c0 & !(Send > cur0) & re0;                 // without: the pending re-test stays pending
                const int  sB  = hm ? base + eL + 1 : sBase;
                const int  sI  = hm ? (reM ? 0 : (base + Send) - sB) : ((c0 & (Send > cur0)) ? (base + Send) - sBase : sIter);
                const bool nRe = hm ? reM : (c0 ? reN : hasRe);
                const int  nRp = hm ? base + eL : (reN ? base + cur0 : rePos);
                hasIns = hm & (eL - 2 >= 64); insPos = hasIns ? base + eL - 2 : insPos;
                sBase = sB; sIter = sI; hasRe = nRe; rePos = nRp;
            }
            width = 64;
            STAT(P_CYC_TAIL, STAT_NOW() - ts5);
            return kGridNext;
        };

        for (;;) {
            // ================================================================ GRID batch
            if (!U16 && sIter <= 64) {
                const int probeStart = hasRe ? rePos : sBase + sIter;
                const int firstPos   = hasIns ? insPos : probeStart;
                int base = firstPos & ~63;
                if (base >= 64 && base + 224 <= n) {
                    // prime the pipeline (the table is exact here): windows of this batch and the next, this batch's slots
                    // peeked, its candidates requested.  The memory round trips of this start are the only ones the grid
                    // batches ever wait for.
                    EMU_CNT(1, 1); STAT(P_PRIME, 1);
                    LANES({
                        S0[I_].P = load_win20(src, base + LANE);
                        S1[I_].P = load_win20(src, base + 64 + LANE);
                        S0[I_].h   = seq_hash<false>(win20_seq(S0[I_].P));
                        S0[I_].ent = ((uint32_t)(base + LANE) << sh) | (seq_tag(S0[I_].P.w[0]) & tagMask);
                        S0[I_].pk  = ((const uint32_t*)tab)[S0[I_].h];
                        const uint32_t q0 = (uint32_t)(base + LANE), pr = S0[I_].pk >> sh;
                        const bool pc = pr < q0 && pr + kMaxDist >= q0 && ((S0[I_].pk ^ S0[I_].ent) & tagMask) == 0;
                        S0[I_].C = load_win20(src, (int)(pc ? pr : q0));
                        if (kLdsWin) pf[I_] = ld32u(src + min_(base + 128 + 4 * LANE, n - 4));
                    })
                    LDS_ORDER();
                    int rc;
                    for (;;) {
                        rc = grid(S2, S0, S1, base); if (rc != kGridNext) break; base += 64;
                        rc = grid(S0, S1, S2, base); if (rc != kGridNext) break; base += 64;
                        rc = grid(S1, S2, S0, base); if (rc != kGridNext) break; base += 64;
                    }
                    if (rc == kGridDone) break;
                    if (rc == kGridStop) {
                        // the next batch is not the consecutive one (a long match, a search past 64 misses, the block's end):
                        // start over if it is still a grid batch, else fall through to the generic one
                        if (sIter <= 64) {
                            const int ps = hasRe ? rePos : sBase + sIter;
                            const int nb = (hasIns ? insPos : ps) & ~63;
                            if (nb >= 64 && nb + 224 <= n) continue;
                        }
                    }
                }
            }
            // ================================================================ GENERIC batch: one lane per probe of the search loop
            {
            STAT(P_GENERIC, 1);
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
    STAT(P_CYC_TOTAL, STAT_NOW() - tBlock0); STAT(P_BLOCKS, 1); STAT(P_SEQ, nseq);
    STAT_FLUSH();
    return nseq;
}

// LZ4_compress_fast_extState's table choice (lz4.c:1389): byU16 below 64 KiB + 11
template <int kLdsWin = 0>
DEV int wave_parse_l1(const uint8_t* __restrict__ src, int n, void* tab, uint64_t* __restrict__ seq, int* lastAnchor, uint8_t* scr = nullptr)
{
    if (n < k64KLimit) return wave_parse_l1_tt<true>(src, n, tab, seq, lastAnchor);
    return wave_parse_l1_tt<false, kLdsWin>(src, n, tab, seq, lastAnchor, scr);
}

// ------------------------------------------------------------------------------------------ EMIT
// What one sequence is in the output, from its record, the record before it and its catch-up length (one lane each).
// The catch-up (lz4.c:1105-1109: while ip > anchor && match > lowLimit && ip[-1] == match[-1]; a re-test has ip == anchor) is
// measured once, by the sizes pass, and kept in a byte per sequence for the write pass (255: longer, measured again).
struct SeqOut { int anchor, lit, mlen, extL, extM, size; uint32_t off; };
DEV int seq_anchor(uint64_t prev, bool first) { return first ? 0 : (int)seq_pos(prev) + kMinMatch + (int)seq_fwd(prev); }
DEV SeqOut seq_layout(uint64_t rec, int anchor, int bk)
{
    SeqOut o;
    const int pos = (int)seq_pos(rec), fwd = (int)seq_fwd(rec);
    o.off = seq_off(rec);
    o.anchor = anchor;
    o.lit  = pos - bk - anchor;
    o.mlen = fwd + bk;
    o.extL = o.lit  >= 15 ? (o.lit  - 15) / 255 + 1 : 0;
    o.extM = o.mlen >= 15 ? (o.mlen - 15) / 255 + 1 : 0;
    o.size = 1 + o.extL + o.lit + 2 + o.extM;
    return o;
}
// the catch-up from `bk` bytes on (the first step below has compared that many), at most maxBack
DEV int seq_backext_from(const uint8_t* __restrict__ src, int pos, int cnd, int maxBack, int bk)
{
    bool open = true;
    while (open && bk + 4 <= maxBack) {
        const uint32_t x = ld32u(src + pos - 4 - bk) ^ ld32u(src + cnd - 4 - bk);
        if (x) { bk += __builtin_clz(x) >> 3; open = false; } else bk += 4;
    }
    while (open && bk < maxBack && src[pos - 1 - bk] == src[cnd - 1 - bk]) ++bk;
    return bk;
}
// its first four bytes without a branch: the dwords in front of the two positions (clamped to the block's start, shifted so
// that the byte right before the position is the top one), equal leading bytes of their XOR, at most maxBack
DEV int seq_backext4(const uint8_t* __restrict__ src, int pos, int cnd, int maxBack, bool* more)
{
    // (a match that starts at the anchor cannot catch up, maxBack == 0: its candidate-side load -- a random sector, the bytes this
    // pass is bound by -- goes to the position's dword instead, which the neighbouring lanes read anyway; the result is 0 either way)
    const int pa = max_(pos - 4, 0), ca = maxBack > 0 ? max_(cnd - 4, 0) : pa;
    const uint32_t xa = ld32u(src + pa) << ((8 * (4 - (pos - pa))) & 31), xc = ld32u(src + ca) << ((8 * (4 - (cnd - ca))) & 31);
    const uint32_t x = xa ^ xc;
    const int e4 = x ? (__builtin_clz(x) >> 3) : 4;
    *more = (e4 == 4) & (maxBack > 4);
    return min_(e4, max_(min_(maxBack, 4), 0));
}

// bytes of the sequences [c*kSeqChunk, min(nseq, (c+1)*kSeqChunk)) of a block; their catch-up lengths go to bkOut[].
// Four sequences per lane and trip: their loads are independent, so one memory round trip serves 256 sequences.
// kBack = false: records that carry the final match (the HC parsers', lz4hc_lazy_device.inl): no catch-up, bkOut unused.
template <bool kBack = true>
DEV uint32_t seq_emit_sizes(const uint8_t* __restrict__ src, const uint64_t* __restrict__ seq, uint8_t* __restrict__ bkOut, int nseq, int c)
{
    const int i0 = c * kSeqChunk, i1 = min_(nseq, i0 + kSeqChunk);
    LV(int, acc);
    LANES({ acc[I_] = 0; })
    for (int i = i0; i < i1; i += 256) {
        LV(uint64_t, rec0); LV(uint64_t, rec1); LV(uint64_t, rec2); LV(uint64_t, rec3);
        LV(int, bk0); LV(int, bk1); LV(int, bk2); LV(int, bk3);
        LV(int, an0); LV(int, an1); LV(int, an2); LV(int, an3);
        LV(bool, mo0); LV(bool, mo1); LV(bool, mo2); LV(bool, mo3);
#define SEQ_SZ_STEP(U, REC, BK, AN, MO) \
        LANES({ \
            const int k_ = min_(i + (U) * 64 + LANE, i1 - 1); \
            REC[I_] = seq[k_]; \
            AN[I_]  = seq_anchor(k_ ? seq[k_ - 1] : 0, k_ == 0); \
            const int pos_ = (int)seq_pos(REC[I_]), cnd_ = pos_ - (int)seq_off(REC[I_]); \
            bool m_ = false; \
            BK[I_] = kBack ? seq_backext4(src, pos_, cnd_, min_(pos_ - AN[I_], cnd_), &m_) : 0; \
            MO[I_] = m_; \
        })
        SEQ_SZ_STEP(0, rec0, bk0, an0, mo0) SEQ_SZ_STEP(1, rec1, bk1, an1, mo1) SEQ_SZ_STEP(2, rec2, bk2, an2, mo2) SEQ_SZ_STEP(3, rec3, bk3, an3, mo3)
#undef SEQ_SZ_STEP
        if (kBack && BALLOT(mo0[I_] | mo1[I_] | mo2[I_] | mo3[I_])) {               // a catch-up beyond four bytes (rare): the loop
#define SEQ_SZ_MORE(REC, BK, AN, MO) \
            LANES({ if (MO[I_]) { const int pos_ = (int)seq_pos(REC[I_]), cnd_ = pos_ - (int)seq_off(REC[I_]); \
                                  BK[I_] = seq_backext_from(src, pos_, cnd_, min_(pos_ - AN[I_], cnd_), 4); } })
            SEQ_SZ_MORE(rec0, bk0, an0, mo0) SEQ_SZ_MORE(rec1, bk1, an1, mo1) SEQ_SZ_MORE(rec2, bk2, an2, mo2) SEQ_SZ_MORE(rec3, bk3, an3, mo3)
#undef SEQ_SZ_MORE
        }
#define SEQ_SZ_FIN(U, REC, BK, AN) \
        LANES({ \
            const int k_ = i + (U) * 64 + LANE; \
            if (k_ < i1) { \
                acc[I_] += seq_layout(REC[I_], AN[I_], BK[I_]).size; \
                if (kBack) bkOut[k_] = (uint8_t)min_(BK[I_], 255); \
            } \
        })
        SEQ_SZ_FIN(0, rec0, bk0, an0) SEQ_SZ_FIN(1, rec1, bk1, an1) SEQ_SZ_FIN(2, rec2, bk2, an2) SEQ_SZ_FIN(3, rec3, bk3, an3)
#undef SEQ_SZ_FIN
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
template <bool kBack = true>
DEV void seq_emit_write(const uint8_t* __restrict__ src, int n, const uint64_t* __restrict__ seq, const uint8_t* __restrict__ bkIn, int nseq, int lastAnchor, int c,
                        uint32_t chunkOff, uint8_t* __restrict__ dst)
{
    const int i0 = c * kSeqChunk, i1 = min_(nseq, i0 + kSeqChunk);
    int op = (int)chunkOff;
    for (int i = i0; i < i1; i += 64) {
        LV(SeqOut, so); LV(int, tok);
        LANES({
            const int k = i + LANE;
            if (k < i1) {
                const uint64_t rec = seq[k];
                const int an = seq_anchor(k ? seq[k - 1] : 0, k == 0);
                int bk = kBack ? (int)bkIn[k] : 0;
                if (kBack && bk == 255) { const int pos = (int)seq_pos(rec), cnd = pos - (int)seq_off(rec); bk = seq_backext_from(src, pos, cnd, min_(pos - an, cnd), 0); }
                so[I_] = seq_layout(rec, an, bk);
            }
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
