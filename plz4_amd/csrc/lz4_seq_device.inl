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
unsigned long long plz4_emu_cnt[8]; unsigned long long plz4_emu_deep[4];               // test diagnostics: grid batches, cold starts, lanes reloaded from memory, twin repairs
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
    STAT_DECL;
    const unsigned long long tBlock0 = STAT_NOW(); (void)tBlock0;

    if (n >= kMinLength) {
        int  insPos  = 0; bool hasIns = true;      // pending table insert ("First Byte" lz4.c:1005-1010; ip-2 lz4.c:1236-1242)
        int  rePos   = 0; bool hasRe  = false;     // pending immediate re-test at ip after a match (lz4.c:1255-1294)
        int  sBase   = 1; int sIter = 0;           // search started at sBase; next un-probed probe number
        int  width   = 16;                         // generic batches: 16 lanes first, 64 when a search drags on
        // the pipeline's three stages; which one is batch k-1 / k / k+1 rotates with the unrolled loop below
        LV(GStage, S0); LV(GStage, S1); LV(GStage, S2);
        LANES({ for (int k = 0; k < 5; ++k) { S0[I_].P.w[k] = 0; S0[I_].C.w[k] = 0; } S0[I_].h = 0; S0[I_].ent = 0; S0[I_].pk = 0; S1[I_] = S0[I_]; S2[I_] = S0[I_]; })
        enum { kGridNext = 0, kGridDone = 1, kGridGeneric = 2, kGridStop = 3 };

        // One batch of the steady state.  On entry: cur.P = the window of `base`, cur.h / ent = its slot and entry, cur.pk = what the
        // slot held when the table was last exact before this batch, cur.C = the window at that entry's position (requested a
        // batch ago), next.P = the window of base + 64 (requested a batch ago), prev.P = the window of base - 64 when the batch
        // before was a grid batch.  Every load below is unconditional: the number of memory operations in flight at any point of
        // the loop is fixed, so waiting for one of them never means waiting for a younger one.
        auto grid = [&](LVREF(GStage, prev), LVREF(GStage, cur), LVREF(GStage, next), const int base) -> int {
            if (sIter > 64) return kGridStop;
            const int probeStart = hasRe ? rePos : sBase + sIter;
            const int firstPos   = hasIns ? insPos : probeStart;
            if ((firstPos & ~63) != base || base + 224 > n) return kGridStop;
            uint32_t* T = (uint32_t*)tab;
            LV(bool, act); LV(uint32_t, r); LV(uint32_t, rent); LV(bool, hit); LV(int, fwd); LV(int, eLane); LV(bool, cand);
            EMU_CNT(0, 1); STAT(P_BATCH, 1);
            const unsigned long long ts0 = STAT_NOW(); (void)ts0;
            STAT_WAIT_VM();
            const unsigned long long ts1 = STAT_NOW(); (void)ts1;
            STAT(P_CYC_MEM, ts1 - ts0);
            // ---- 1. peek for batch k+1 on the exact table, then this batch's table exchange
            LANES({
                next[I_].h   = seq_hash<false>(win20_seq(next[I_].P));
                next[I_].ent = ((uint32_t)(base + 64 + LANE) << sh) | (seq_tag(next[I_].P.w[0]) & tagMask);
                next[I_].pk  = T[next[I_].h];
            })
            LDS_ORDER();
            LANES({
                const int q = base + LANE;
                const int isIns = hasIns && q == insPos;
                act[I_] = isIns || q >= probeStart;
                hit[I_] = false; fwd[I_] = 0; r[I_] = 0; rent[I_] = 0; cand[I_] = false;
                if (act[I_]) {
                    rent[I_] = lds_max_rtn(&T[cur[I_].h], cur[I_].ent);
                    r[I_]    = rent[I_] >> sh;
                    cand[I_] = !isIns && r[I_] < (uint32_t)q && r[I_] + kMaxDist >= (uint32_t)q && ((rent[I_] ^ cur[I_].ent) & tagMask) == 0;
                }
            })
            // LDS atomics on one slot are expected to resolve in ascending lane order (then r is the nearest earlier twin or the
            // pre-batch value): any other order shows up as r >= q somewhere.  And an entry that is neither the peeked one nor a
            // position of this batch or the one before cannot be (nothing else touched the table since the peek).  Either way:
            // the commits are taken back and the generic batch takes over.
            STAT_WAIT_LGKM();
            const unsigned long long ts2 = STAT_NOW(); (void)ts2;
            STAT(P_CYC_LDS, ts2 - ts1);
            const uint64_t misorder = BALLOT(act[I_] && (r[I_] >= (uint32_t)(base + LANE) ||
                                                         (cand[I_] && rent[I_] != cur[I_].pk && r[I_] + 64u < (uint32_t)base)));
            if (misorder) {
                EMU_CNT(2, 1);
                LANES({ if (act[I_]) lds_min(&T[cur[I_].h], rent[I_]); })  // min over a slot's group == its pre-batch value
                LDS_FENCE();
                return kGridGeneric;
            }
            // ---- 2. candidates and their windows.  The entry a lane displaced (rent) is the slot's content under the assumption
            // that every earlier lane of this batch was executed.  Its window: the peeked one when the entry is the peeked entry,
            // else the entry is a position of this batch or of the batch before (nothing else touched the table since the peek)
            // and the window is in registers there: one ds_bpermute round per source, no branch.
            LV(bool, pb);                                  // a lane that can be a probe (active, not the pending insert)
            LV(Win20, W0);                                 // window of rent's position
            LANES({ pb[I_] = act[I_] && !(hasIns && base + LANE == insPos); })
            auto gather = [&](LVREF(uint32_t, ceV), LVREF(Win20, W)) {
                LANES({
                    const int t = (int)(ceV[I_] >> sh) - base;               // (& 63: the lane in either batch)
                    const bool same = ceV[I_] == cur[I_].pk;
                    for (int k = 0; k < 5; ++k) {
                        const uint32_t a = SHFLF(cur, P.w[k], t), b = SHFLF(prev, P.w[k], t);
                        W[I_].w[k] = same ? cur[I_].C.w[k] : (t >= 0 ? a : b);
                    }
                })
            };
            auto compare = [&](LVREF(uint32_t, ceV), LVREF(Win20, W), LVREF(bool, hitV), LVREF(int, fwdV)) {
                LANES({
                    const uint32_t q = (uint32_t)(base + LANE), cp = ceV[I_] >> sh;
                    const bool ok = pb[I_] && cp < q && cp + kMaxDist >= q && ((ceV[I_] ^ cur[I_].ent) & tagMask) == 0;
                    hitV[I_] = ok && W[I_].w[0] == cur[I_].P.w[0];
                    fwdV[I_] = hitV[I_] ? win20_fwd(cur[I_].P, W[I_]) : 0;
                })
            };
            gather(rent, W0);
            STAT_WAIT_LGKM();
            const unsigned long long ts3 = STAT_NOW(); (void)ts3;
            STAT(P_CYC_REFRESH, ts3 - ts2);
            compare(rent, W0, hit, fwd);
            LANES({ eLane[I_] = LANE + kMinMatch + fwd[I_]; })    // lane index just past a match that starts here
            // ---- 3. requests for the batches to come: the peeked candidates of batch k+1 (every lane loads: its candidate, or
            // its own position when it has none), the window of batch k+2 into registers batch k-1 is done with (its C: its P
            // still serves this batch's chains; they change places at the end)
            LANES({
                const uint32_t q1 = (uint32_t)(base + 64 + LANE), pr = next[I_].pk >> sh;
                const bool pc = pr < q1 && pr + kMaxDist >= q1 && ((next[I_].pk ^ next[I_].ent) & tagMask) == 0;
                next[I_].C = load_win20(src, (int)(pc ? pr : q1));
                prev[I_].C = load_win20(src, base + 128 + LANE);
            })

            const uint64_t hits0  = BALLOT(hit[I_]);
            const unsigned long long ts4 = STAT_NOW(); (void)ts4;
            STAT(P_CYC_CMP, ts4 - ts3);
            const uint64_t twins0 = BALLOT(act[I_] && r[I_] >= (uint32_t)firstPos);   // earlier twin inside this batch
            const uint64_t anyTwins = twins0;
            const uint64_t special0 = hits0 & BALLOT(fwd[I_] == 16);           // longer than the speculative window: the hop needs its end
            const int  cur0 = probeStart - base;                  // first probe lane (>= 64: none in this batch)
            const bool re0  = hasRe;
            const uint64_t insBit0 = hasIns ? (1ull << (insPos - base)) : 0;
            int lim0 = sBase + 65 - base; if (lim0 > 63) lim0 = 63;   // probe number <= 65 keeps the stride at 1

            uint64_t mm = 0;           // executed match lanes
            int      eL = 0;           // end lane of the last executed match
            uint64_t E = 0;            // executed lanes (probes + inserts)
            int      Send = 0;
            bool     finished = false;
            // ---- 4. the walk, in straight passes: scalar hop over the recorded matches, the executed lanes derived per lane
            // (walk_pass).  Then: does a probe's candidate lie in a lane of this batch the parser did not execute?  Such lanes
            // -- all of them at once -- follow the chain of displaced entries down to the first executed lane or to an entry
            // from before the batch (resolve), fetch that window from the registers and compare again; the walk is redone only
            // if a probe's hit or length changed, and the result stands once a resolve under the final executed set changes no
            // probe's candidate (the sequential parse is the one consistent assignment: a lane depends on lower lanes only).
            // A match longer than the speculative window under the hop, a chain that leaves the registers, or three rounds
            // without agreement send the batch through the general loop below instead.
            uint64_t probes = 0;
            auto walk_pass = [&](const uint64_t hitsM, LVREF(int, eV)) {
                LV(int, nextHit);      // first recorded match at or after the end of the match that starts here (64: none)
                LANES({
                    const uint64_t ah = (eV[I_] < 64) ? (hitsM >> eV[I_]) : 0;
                    nextHit[I_] = ah ? eV[I_] + ctz64(ah) : 64;
                })
                mm = 0; eL = 0;
                int w = 64;
                if (cur0 < 64) { const uint64_t hm = hitsM & (~0ull << cur0); if (hm) w = ctz64(hm); }
                if (w > lim0) w = 64;                                  // (the stride limit concerns the first match only)
                const unsigned long long tw1 = STAT_NOW(); (void)tw1;
                if (w < 64) {
                    const uint64_t low = (w == 0 ? 1ull : 0ull);
                    do {
                        for (int u = 0; u < 4; ++u) {
                            const int n1 = RL(nextHit, w & 63);
                            mm |= 1ull << (w & 63);
                            w = (w < 64) ? n1 : 64;
                        }
                        STAT(P_HOPS, 4);
                    } while (w < 64);
                    mm = (mm & ~1ull) | low;
                    eL = RL(eV, 63 - __builtin_clzll(mm));
                }
                const unsigned long long tw2 = STAT_NOW(); (void)tw2;
                STAT(P_CYC_HOP, tw2 - tw1);
                Send = mm ? 64 : min_(64, lim0 + 1);
                const uint64_t mmL = mm; const int SendL = Send;
                LV(int, stA);
                LANES({ stA[I_] = ((mmL >> LANE) & 1) ? eV[I_] : 0; })
                SCAN_MAX_EXCL(stA);
                const uint64_t hasPm = BALLOT(stA[I_] > 0);
                LANES({ stA[I_] = stA[I_] > 0 ? stA[I_] : cur0; })                // where probing resumed before me
                probes = BALLOT(LANE >= stA[I_] && LANE < SendL && LANE >= cur0);
                E = probes | insBit0 | (hasPm & BALLOT(stA[I_] == LANE + 2));      // + the ip-2 inserts (lz4.c:1236-1242)
                STAT(P_CYC_E, STAT_NOW() - tw2);
            };
            bool slow;
            walk_pass(hits0, eLane);
            slow = (mm & special0) != 0;
            EMU_CNT(4, slow);
#if defined(PLZ4_EMU) && defined(PLZ4_EMU_TRACE)
            if (base < 200) { fprintf(stderr, "base %d cur0 %d probes %llx E %llx twins %llx mm %llx slow %d hasIns %d insPos %d\n", base, cur0, (unsigned long long)probes, (unsigned long long)E, (unsigned long long)twins0, (unsigned long long)mm, (int)slow, (int)hasIns, insPos);
                for (int l = 0; l < 64; ++l) fprintf(stderr, " l%d r%u pk%u hit%d fwd%d |", l, rent[l] >> sh, cur[l].pk >> sh, (int)hit[l], fwd[l]); fprintf(stderr, "\n"); }
#endif
            if (!slow && (twins0 & probes)) {
                const uint64_t EL0 = E;
                const uint64_t bad = twins0 & probes & BALLOT(!((EL0 >> (((int)r[I_] - base) & 63)) & 1));
                if (bad) {
                    EMU_CNT(5, 1);
#if defined(PLZ4_EMU) && defined(PLZ4_EMU_TRACE)
                    if (base == 64) { fprintf(stderr, "base 64 bad %llx probes %llx E %llx twins %llx mm %llx\n", (unsigned long long)bad, (unsigned long long)probes, (unsigned long long)E, (unsigned long long)twins0, (unsigned long long)mm);
                        for (int l = 0; l < 64; ++l) fprintf(stderr, " l%d r%u pk%u hit%d fwd%d |", l, rent[l] >> sh, cur[l].pk >> sh, (int)hit[l], fwd[l]); fprintf(stderr, "\n"); }
#endif
                    LV(uint32_t, ce); LV(Win20, W1);
                    LANES({ ce[I_] = rent[I_]; })
                    int it = 0;
                    for (; it < 3 && !slow; ++it) {
                        // resolve under E: from the commit's answer down the chain of displaced entries
                        LV(uint32_t, ce2);
                        LANES({ ce2[I_] = rent[I_]; })
                        for (;;) {
                            const uint64_t EL = E;
                            const uint64_t mv = BALLOT(pb[I_] && (ce2[I_] >> sh) >= (uint32_t)firstPos && !((EL >> (((int)(ce2[I_] >> sh) - base) & 63)) & 1));   // (an entry of [base, firstPos) is older than the batch)
                            if (!mv) break;
                            LANES({ const uint32_t nx = SHFL(rent, (int)(ce2[I_] >> sh) - base); if ((mv >> LANE) & 1) ce2[I_] = nx; })
                        }
                        if (!(BALLOT(ce2[I_] != ce[I_]) & probes)) break;          // every probe has the candidate the parser saw
                        LANES({ ce[I_] = ce2[I_]; })
                        // an entry from before the previous batch that is not the peeked one: its window is not in the registers
                        if (BALLOT(pb[I_] && ce[I_] != cur[I_].pk && (ce[I_] >> sh) + 64u < (uint32_t)base) & probes) { slow = true; break; }
                        LV(bool, hitN); LV(int, fwdN);
                        gather(ce, W1);
                        compare(ce, W1, hitN, fwdN);
                        const uint64_t diff = BALLOT(hitN[I_] != hit[I_] || fwdN[I_] != fwd[I_]) & probes;
                        LANES({ hit[I_] = hitN[I_]; fwd[I_] = fwdN[I_]; eLane[I_] = LANE + kMinMatch + fwdN[I_]; r[I_] = ce[I_] >> sh; })
                        if (!diff) break;                                          // same walk, same executed set: consistent
                        EMU_CNT(6, 1);
                        const uint64_t hitsN = BALLOT(hit[I_]);
                        walk_pass(hitsN, eLane);
                        if (mm & hitsN & BALLOT(fwd[I_] == 16)) slow = true;
                    }
                    if (it == 3) slow = true;
                    if (slow) {                                                    // back to the commit's answer for the general loop
                        EMU_CNT(7, 1);
                        compare(rent, W0, hit, fwd);
                        LANES({ eLane[I_] = LANE + kMinMatch + fwd[I_]; r[I_] = rent[I_] >> sh; })
                    }
                }
            }
            const unsigned long long tw3 = STAT_NOW(); (void)tw3;
            if (slow) {
            uint64_t hits = hits0, twins = twins0, specialLeft = special0;
            uint64_t cwValid = BALLOT(cand[I_]);                       // lanes whose C holds the window of their rent's position
            mm = 0; eL = 0; E = 0; Send = 0;
            STAT(P_REPAIR, 1);
            // ---- 4. scalar hop over the recorded matches only; everything else is derived per lane
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
                // ---- 5. which lanes did the sequential parser execute
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
                        EMU_CNT(3, 1);
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
                        if (cp + kMaxDist >= qb && ((ce ^ RLF(cur, ent, b)) & tagMask) == 0) {
                            Win20 Pb; for (int k = 0; k < 5; ++k) Pb.w[k] = RLF(cur, P.w[k], b);
                            Win20 Cn;
                            if (cp >= (uint32_t)base) { const int t = (int)cp - base; for (int k = 0; k < 5; ++k) Cn.w[k] = RLF(cur, P.w[k], t); }
                            // a pre-batch entry was displaced by lane tl, and if that lane took it for a candidate (a twin
                            // usually repeats the very same bytes) its window already holds what is needed: no memory round trip
                            else if (tl >= 0 && ((cwValid >> tl) & 1)) { for (int k = 0; k < 5; ++k) Cn.w[k] = RLF(W0, w[k], tl); }
                            else { Cn = load_win20(src, (int)cp); for (int k = 0; k < 5; ++k) Cn.w[k] = UNI(Cn.w[k]); }
                            if (Cn.w[0] == Pb.w[0]) { nhit = 1; nfwd = win20_fwd(Pb, Cn); }
                        }
                        WL(rent, b, ce); WL(r, b, cp); WL(hit, b, nhit != 0); WL(fwd, b, nfwd);
                        WL(eLane, b, b + kMinMatch + nfwd);
                        const uint64_t bit = 1ull << b;
                        hits = nhit ? (hits | bit) : (hits & ~bit);
                        specialLeft = (nhit && nfwd == 16) ? (specialLeft | bit) : (specialLeft & ~bit);
                        twins &= ~bit;
                        cwValid &= ~bit;                                    // lane b's C no longer belongs to its (new) rent
                        keep = mm & (bit - 1); resume = b;
                        continue;
                    }
                }
                break;
            }
            }
            // lz4.c:1233: a match that ends at or past the last probe position ends the block.  Decided here, from the
            // final walk: the pass that finished a long match may have been redone after a twin repair, and the redo
            // sees that match as an ordinary one.
            finished = (mm != 0) && (base + eL >= lastProbe);
            const unsigned long long ts5 = STAT_NOW(); (void)ts5;
            STAT(P_CYC_WALK, ts5 - ts4); STAT(P_CYC_SLOW, ts5 - tw3);

            // ---- 6. one record per executed match
            if (mm) {
                const uint64_t mmL = mm; const int at = nseq;
                LANES({
                    if ((mmL >> LANE) & 1)
                        seq[at + LANE_RANK(mmL)] = seq_pack((uint32_t)(base + LANE), (uint32_t)fwd[I_], (uint32_t)(base + LANE) - r[I_]);
                })
                nseq += __builtin_popcountll(mm);
                anchor = base + eL;
            }
            if (finished) return kGridDone;

            // ---- 7. parser state after this batch
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

            // ---- 8. patch the table to the sequential result
            const uint64_t EL2 = E;
            LANES({ if (act[I_] && !((EL2 >> LANE) & 1)) lds_min(&T[cur[I_].h], rent[I_]); })
            if (anyTwins) { LDS_ORDER(); LANES({ if ((EL2 >> LANE) & 1) lds_max(&T[cur[I_].h], cur[I_].ent); }) }
            LDS_ORDER();
            width = 64;
            LANES({ prev[I_].P = prev[I_].C; })                            // the window of batch k+2 takes its place
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
