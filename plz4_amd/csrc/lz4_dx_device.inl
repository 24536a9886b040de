// lz4_dx_device.inl -- LZ4_decompress_safe (/root/reference/internal/pkg/clz4/lz4.c:2022-2445) for a FEW blocks, cut across the
// whole chip.  The record decoder of lz4_device.inl is one wavefront per block: a 4 MiB block is 75 ms of one wave whatever else
// the chip does, which is what plz4.DecompressBlock (plz4_block.go:131-172) or a reader with a handful of blocks in flight would
// pay.  A block's decode is a chain twice over -- the token chain (where a sequence starts is only known once the one before it is
// parsed) and the copy chain (a match copies bytes an earlier match produced) -- and both are cut here the way a GPU cuts chains:
// by pointer jumping.
//
//   A  dx_segment_table  one wave per 8 KiB of INPUT: for every byte position p of the segment, as if a sequence started there,
//                        where the token chain from p leaves the segment and how many output bytes it passes (64 positions at a
//                        time, from the segment's end backwards: five doubling rounds inside the batch, one table lookup behind it).
//   B  dx_stitch         one wave per block: the true chain's entry into every segment and the output position there, one table
//                        lookup per segment (a sequence the tables could not tell -- length bytes without end -- is parsed here).
//   C  wave_dx_fill      one wave per segment again, now on the TRUE chain: literals go to their place in the output; a match
//                        is not copied -- its source may not exist yet -- but written down as pointers, ptr[o + i] = o + i - offset,
//                        into a table that starts as the identity.  The reference's accept / reject rules are walked as they are
//                        (fast loop in the middle of a block; the last two segments as one unit with the safe loop's rules): whatever
//                        they would not let pass flags the block, and a flagged block is decoded again by the one-wave decoder,
//                        whose result and error code are the reference's by construction.
//   D  dx_jump           ptr[p] <- ptr[ptr[p]] over every output byte, in place, until nothing moves (log2 of the deepest copy
//                        chain: 8-12 rounds on text, 22 on a 4 MiB run of one byte): every byte then points at the literal it is.
//   E  dx_gather         out[p] <- out[ptr[p]].
//
// Compiled for the CPU as-is by tests/emu (emu_dx_decode) and checked there against the oracle on valid and corrupt blocks.
#pragma once
#include "lz4_device.inl"

namespace plz4 {

enum : int { kDxSeg = 8192,                 // input bytes per table segment / fill unit
             kDxExt = 32,                   // length bytes a table entry looks through (longer: the stitch parses that sequence itself)
             kDxMaxOut = (4 << 20) + 8,     // the path is taken for outputs up to here (blk/pool.go:23-26: bsz + 8)
             kDxRounds = 24 };              // jump rounds launched (2^24 > kDxMaxOut; a block stops taking part once nothing moves)

struct DxUnit { int32_t ip, op, stop, pad; };               // where the true chain enters a segment (ip < 0: it does not), the output position there, where the unit ends
struct DxInfo { int32_t bad, outLen, tailFrom, pad; uint32_t moved[kDxRounds + 1]; };

DEV uint64_t dx_ent(uint32_t exitPos, uint32_t sum, bool slow) { return (uint64_t)exitPos | ((uint64_t)(slow ? 1u : 0u) << 31) | ((uint64_t)sum << 32); }
DEV uint32_t dx_exit(uint64_t e) { return (uint32_t)e & 0x7FFFFFFFu; }
DEV bool     dx_slow(uint64_t e) { return (((uint32_t)e) >> 31) != 0u; }
DEV uint32_t dx_sum(uint64_t e) { return (uint32_t)(e >> 32); }
DEV int dx_segments(int n) { return n > 0 ? (n + kDxSeg - 1) / kDxSeg : 1; }
DEV int dx_tail_from(int nseg) { return nseg >= 2 ? nseg - 2 : 0; }           // the last two segments are one unit (the block's end rules)

// One lane: the sequence that would start at input position p.  *next = where the one behind it starts, *outLen = the bytes it
// produces.  false: not to be told here (too close to the input's end -- the tail unit walks there itself -- or more than kDxExt
// length bytes).  Any byte string parses as SOMETHING: whether p is a real sequence start is the stitch's business.
DEV bool dx_parse_at(const uint8_t* __restrict__ in, const int n, const int p, int* next, int* outLen)
{
    if (p + 24 > n) return false;
    const uint32_t t = in[p];
    int l = (int)(t >> 4), q = p + 1;
    if (l == 15) {
        uint32_t b; int k = 0;
        do { if (q + 24 > n || k == kDxExt) return false; b = in[q++]; l += (int)b; ++k; } while (b == 255u);
    }
    q += l;
    if (q + 24 > n) return false;
    int m = (int)(t & 15u) + kMinMatch;
    q += 2;
    if ((t & 15u) == 15u) {
        uint32_t b; int k = 0;
        do { if (q + 24 > n || k == kDxExt) return false; b = in[q++]; m += (int)b; ++k; } while (b == 255u);
    }
    *next = q; *outLen = l + m;
    return true;
}

// A.  T[p] for the positions of segment j (T has one entry per input byte).
DEV void dx_segment_table(const uint8_t* __restrict__ in, const int n, const int j, uint64_t* __restrict__ T)
{
    const int s0 = j * kDxSeg, s1 = min_(s0 + kDxSeg, n);
    if (s0 >= s1) return;
    for (int base = (s1 - 1) & ~63; base >= s0; base -= 64) {
        LV(int, tgt); LV(int, acc); LV(int, slow);
        LANES({
            const int p = base + LANE;
            int nx = p, ol = 0;
            const bool ok = p < s1 && dx_parse_at(in, n, p, &nx, &ol);
            tgt[I_] = ok ? nx : p; acc[I_] = ok ? ol : 0; slow[I_] = (p < s1 && !ok) ? 1 : 0;
        })
        // inside the batch: a chain advances at least 3 bytes per sequence, so five doublings reach past its 64 positions
        for (int r = 0; r < 5; ++r) {
            LV(int, t2); LV(int, a2); LV(int, s2);
            LANES({ const int sl = (tgt[I_] - base) & 63; t2[I_] = SHFL(tgt, sl); a2[I_] = SHFL(acc, sl); s2[I_] = SHFL(slow, sl); })
            LANES({
                const int p = base + LANE;
                if (!slow[I_] && tgt[I_] > p && tgt[I_] < base + 64) { tgt[I_] = t2[I_]; acc[I_] += a2[I_]; slow[I_] = s2[I_]; }
            })
        }
        // behind the batch: the table of the positions above it (written by the steps before this one)
        LANES({
            const int p = base + LANE;
            uint32_t ex = (uint32_t)tgt[I_], sm = (uint32_t)acc[I_]; bool sl = slow[I_] != 0;
            if (p < s1 && !sl && tgt[I_] >= base + 64 && tgt[I_] < s1) { const uint64_t e = T[tgt[I_]]; ex = dx_exit(e); sm += dx_sum(e); sl = dx_slow(e); }
            if (p < s1) T[p] = dx_ent(ex, sm, sl);
        })
        WAVE_FENCE();
    }
}

// one sequence at position e, whatever its length bytes (uniform code: every lane reads the same bytes); false: it runs into the
// last 24 bytes of the input
DEV bool dx_parse_uniform(const uint8_t* __restrict__ in, const int n, const int e, int* next, int64_t* outLen)
{
    if (e + 24 > n) return false;
    const uint32_t t = UNI((uint32_t)in[e]);
    int64_t l = (int64_t)(t >> 4); int q = e + 1;
    if (l == 15) { uint32_t b; do { if (q + 24 > n) return false; b = UNI((uint32_t)in[q]); ++q; l += b; } while (b == 255u); }
    if ((int64_t)q + l + 24 > (int64_t)n) return false;
    q += (int)l;
    int64_t m = (int64_t)(t & 15u) + kMinMatch;
    q += 2;
    if ((t & 15u) == 15u) { uint32_t b; do { if (q + 24 > n) return false; b = UNI((uint32_t)in[q]); ++q; m += b; } while (b == 255u); }
    *next = q; *outLen = l + m;
    return true;
}

// B.  units[0 .. tailFrom]: the true chain's entries.  Returns 0, or 1 when the block is left to the one-wave decoder.
DEV int dx_stitch(const uint8_t* __restrict__ in, const int n, const int cap, const uint64_t* __restrict__ T, DxUnit* units, const int nseg)
{
    if (n <= 0 || cap <= 0 || cap > kDxMaxOut) return 1;
    const int jt = dx_tail_from(nseg);
    int e = 0; int64_t O = 0;
    bool early = false;                                                  // the tail unit starts before its segments (a sequence that reaches the block's end)
    for (int j = 0; j < jt; ++j) {
        const int s1 = (j + 1) * kDxSeg;
        DxUnit u; u.ip = -1; u.op = 0; u.stop = s1; u.pad = 0;
        if (!early && e < s1) {
            u.ip = e; u.op = (int)O;
            while (e < s1) {
                const uint64_t x = UNI(T[e]);
                O += dx_sum(x); e = (int)dx_exit(x);
                if (dx_slow(x)) {                                       // the chain stands at a sequence the tables could not tell
                    int nx = 0; int64_t ol = 0;
                    if (!dx_parse_uniform(in, n, e, &nx, &ol)) {        // ... and it runs into the block's end: everything from here on is the tail unit's
                        early = true; u.stop = e;
                        if (u.ip >= e) u.ip = -1;
                        break;
                    }
                    O += ol; e = nx;
                }
                if (O > (int64_t)cap) return 1;
            }
        }
        LANES({ if (LANE == 0) units[j] = u; })
    }
    if (e >= n) return 1;                                                // (the last sequence is the tail unit's: it cannot start at the end)
    DxUnit u; u.ip = e; u.op = (int)O; u.stop = n; u.pad = 0;
    LANES({ if (LANE == 0) units[jt] = u; })
    return 0;
}

// a match as pointers: ptr[op + i] = op + i - offset.  An overlapping match (offset < len) points into itself, which is what
// its bytes are (lz4.c:2406-2414).
DEV void dx_fill_match(uint32_t* __restrict__ ptr, const int64_t op, const int offset, const int len)
{
    LANES({ for (int i = LANE; i < len; i += 64) ptr[op + i] = (uint32_t)(op + i - offset); })
}

// The vector path of the one-wave decoder (wave_decode_plain_batch, lz4_device.inl) with the copies taken out: 64 lanes look at
// the next 64 input bytes as 64 hypothetical sequence starts, a scalar hop follows the real chain while the sequences are plain,
// a prefix sum places them; literals are written, matches become pointers.  Sequences that start at or behind ipStop are the next
// unit's.  Returns the sequences taken (0: the sequential step's turn).
DEV int dx_plain_batch(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t* __restrict__ ptr, int* ipp, int64_t* opp,
                       LVREF(v16u_t, win), int* winIp, const int ipStop)
{
    const int ip0 = *ipp; const int64_t op0 = *opp;
    if (*winIp != ip0) { LANES({ win[I_] = *(const v16u_t*)(src + ip0 + LANE); }) }
    LV(uint32_t, b0); LV(int, ll); LV(int, ml); LV(int, off); LV(int, nxt); LV(int, outLen); LV(int, plain); LV(int, coop);
    LANES({
        const uint64_t w0 = (uint64_t)win[I_].w[0] | ((uint64_t)win[I_].w[1] << 32);
        const uint64_t w1 = (uint64_t)win[I_].w[2] | ((uint64_t)win[I_].w[3] << 32);
        const uint32_t t = (uint32_t)(w0 & 0xFF);
        const int l = (int)(t >> 4), mn = (int)(t & 15);
        const int offAt = LANE + 1 + l;
        const uint64_t sel = (l <= 5) ? w0 : (l <= 9 ? ((w0 >> 32) | (w1 << 32)) : w1);
        const int      sft = 8 * (1 + l - (l <= 5 ? 0 : (l <= 9 ? 4 : 8)));
        const uint32_t o16 = (uint32_t)((sel >> sft) & 0xFFFF);
        const bool     lng = (mn == 15);
        const int      ek  = 3 + l;
        const uint32_t e   = (uint32_t)(((ek < 8) ? (w0 >> (8 * ek)) : (w1 >> (8 * (ek & 7)))) & 0xFF);
        const int      mlen = lng ? 19 + (int)e : mn + kMinMatch;
        const bool parse = (l < 14) && (offAt <= 64) && (!lng || (l <= 12 && e < 255));
        const bool simple = parse && !lng && o16 >= (uint32_t)mlen;
        b0[I_] = t; ll[I_] = l; ml[I_] = mlen; off[I_] = (int)o16;
        nxt[I_] = offAt + 2 + (lng ? 1 : 0); outLen[I_] = l + mlen;
        coop[I_]  = parse && !simple && o16 >= 1;
        plain[I_] = simple || coop[I_];
    })
    const uint64_t plainMask = BALLOT(plain[I_]);
    if (!(plainMask & 1)) return 0;
    uint64_t members = 0;
    {
        LV(int, nxtC);
        LANES({ nxtC[I_] = min_(nxt[I_], 64); })
        int cur = 0, seen = 0;
        do {
            for (int u = 0; u < 4; ++u) {
                members |= 1ull << (cur & 63);
                cur = RL(nxtC, cur & 63);
                seen |= cur;
            }
        } while (seen < 64);
        const uint64_t odd = members & ~plainMask;
        if (odd) members &= (1ull << ctz64(odd)) - 1;
        if (ipStop - ip0 < 64) members &= (1ull << (ipStop - ip0)) - 1;    // (ip0 < ipStop: bit 0 stays)
    }
    LV(int, acc); LV(int, outStart); LV(int, sp);
    { const uint64_t mL = members; LANES({ acc[I_] = ((mL >> LANE) & 1) ? outLen[I_] : 0; }) }
    SCAN_INCL(acc);
    {
        const uint64_t mL = members;
        LANES({
            outStart[I_] = (int)op0 + acc[I_] - (((mL >> LANE) & 1) ? outLen[I_] : 0);
            sp[I_]       = outStart[I_] + ll[I_] - off[I_];
        })
        const uint64_t stop = BALLOT(((mL >> LANE) & 1) && (sp[I_] < 0 || acc[I_] > 1024));
        if (stop) members &= (1ull << ctz64(stop)) - 1;
    }
    if (!members) return 0;
    const uint64_t mL = members;
    const int last = 63 - __builtin_clzll(members);
    const int ipn  = ip0 + RL(nxt, last);
    LANES({ win[I_] = *(const v16u_t*)(src + ipn + LANE); })                 // (ipn + 63 + 16 < ip0 + 160 <= iend)
    *winIp = ipn;
    // literal bytes: every window byte finds the member it follows
    LANES({
        const uint64_t upto = mL & ((LANE >= 63) ? ~0ull : ((2ull << LANE) - 1));
        const int m  = upto ? 63 - __builtin_clzll(upto) : LANE;
        const int os = SHFL(outStart, m), lm = SHFL(ll, m);
        if (upto && LANE > m && LANE <= m + lm) dst[os + (LANE - m - 1)] = (uint8_t)b0[I_];
    })
    // matches: pointers, nothing is read
    const uint64_t coopM = mL & BALLOT(coop[I_]);
    LANES({
        if (((mL & ~coopM) >> LANE) & 1) {
            const int o = outStart[I_] + ll[I_];
            for (int i = 0; i < ml[I_]; ++i) ptr[o + i] = (uint32_t)(o + i - off[I_]);
        }
    })
    for (uint64_t pend = coopM; pend; pend &= pend - 1) {
        const int f = ctz64(pend);
        dx_fill_match(ptr, (int64_t)RL(outStart, f) + RL(ll, f), RL(off, f), RL(ml, f));
    }
    *ipp = ipn;
    *opp = (int64_t)RL(outStart, last) + RL(outLen, last);
    return __builtin_popcountll(members);
}

// C.  The sequences from (ip0, op0) on: a unit in the middle of a block (tail = false) walks the reference's fast loop
// (lz4.c:2083-2209) up to ipStop and returns -1 for anything that loop would hand to the safe loop or reject; the tail unit walks to
// the end of the block under both loops' rules (:2215-2435).  Returns the output position reached, or -1: the block is left to the
// one-wave decoder, whose verdict is the reference's.  An offset of 0 (liblz4 zero-fills, :499-507) is left to it as well.
DEV int64_t wave_dx_fill(const uint8_t* __restrict__ src, const int n, uint8_t* __restrict__ dst, const int cap, uint32_t* __restrict__ ptr,
                         const int ip0, const int64_t op0, const int ipStop, const bool tail)
{
    const int iend = n;
    const int64_t oend = cap;
    int ip = ip0; int64_t op = op0;
    bool fast = (oend - op) >= 64;
    if (!tail && !fast) return -1;
    LV(v16u_t, win); int winIp = -1;
    LANES({ win[I_].w[0] = 0; win[I_].w[1] = 0; win[I_].w[2] = 0; win[I_].w[3] = 0; })
    for (;;) {
        if (!tail && ip >= ipStop) return op;
        if (fast && ip + 160 <= iend && op + 1088 <= oend) {
            const int nm = dx_plain_batch(src, dst, ptr, &ip, &op, win, &winIp, tail ? iend : ipStop);
            if (nm > 0) continue;
        }
        const uint32_t token = UNI((uint32_t)src[ip]); ip++;
        int64_t ll = token >> 4, ml; int offset; int64_t mpos;
        if (fast) {
            bool toSafeLit = false;
            if (ll == 15) {
                const int64_t a = read_more_len(src, &ip, iend - 15, true);
                if (a < 0) return -1;
                ll += a;
                if (op + ll > oend - 32 || (int64_t)ip + ll > iend - 32) toSafeLit = true;
            } else if (!(ip <= iend - 17)) {
                toSafeLit = true;
            }
            if (toSafeLit) { if (!tail) return -1; fast = false; goto safe_literals; }
            wave_copy(dst + op, src + ip, (int)ll);
            ip += (int)ll; op += ll;
            offset = (int)UNI((uint32_t)ld16u(src + ip)); ip += 2;
            mpos = op - offset;
            ml = token & 15;
            if (ml == 15) {
                const int64_t a = read_more_len(src, &ip, iend - kLastLiterals + 1, false);
                if (a < 0) return -1;
                ml += a + kMinMatch;
            } else ml += kMinMatch;
            if (op + ml >= oend - 64) { if (!tail) return -1; fast = false; goto safe_match; }
            if (mpos < 0 || offset == 0) return -1;                            // lz4.c:2161
            dx_fill_match(ptr, op, offset, (int)ml);
            op += ml;
            continue;
        }
        if (ll != 15 && ip < iend - 16 && op <= oend - 32) {                   // shortcut, lz4.c:2230-2261
            wave_copy(dst + op, src + ip, (int)ll);
            op += ll; ip += (int)ll;
            ml = token & 15;
            offset = (int)UNI((uint32_t)ld16u(src + ip)); ip += 2;
            mpos = op - offset;
            if (ml != 15 && offset >= 8 && mpos >= 0) {
                dx_fill_match(ptr, op, offset, (int)ml + kMinMatch);
                op += ml + kMinMatch;
                continue;
            }
            goto match_len;
        }
        if (ll == 15) {
            const int64_t a = read_more_len(src, &ip, iend - 15, true);
            if (a < 0) return -1;
            ll += a;
        }
safe_literals:
        if (op + ll > oend - kMfLimit || (int64_t)ip + ll > iend - (2 + 1 + kLastLiterals)) {
            if ((int64_t)ip + ll != iend || op + ll > oend) return -1;         // lz4.c:2312-2318
            wave_copy(dst + op, src + ip, (int)ll);
            ip += (int)ll; op += ll;
            break;
        }
        wave_copy(dst + op, src + ip, (int)ll);
        ip += (int)ll; op += ll;
        offset = (int)UNI((uint32_t)ld16u(src + ip)); ip += 2;
        mpos = op - offset;
        ml = token & 15;
match_len:
        if (ml == 15) {
            const int64_t a = read_more_len(src, &ip, iend - kLastLiterals + 1, false);
            if (a < 0) return -1;
            ml += a;
        }
        ml += kMinMatch;
safe_match:
        if (mpos < 0 || offset == 0) return -1;                                // lz4.c:2356
        if (op + ml > oend - kLastLiterals) return -1;                         // lz4.c:2421-2423
        dx_fill_match(ptr, op, offset, (int)ml);
        op += ml;
    }
    return op;
}

// D.  One jump of the 64 x 4 pointers from p0 on: true when one of them moved.
DEV bool dx_jump(uint32_t* __restrict__ ptr, const int p0, const int outLen)
{
    LV(int, mv);
    LANES({
        mv[I_] = 0;
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + 4 * LANE + k;
            if (p < outLen) {
                const uint32_t q = ptr[p];
                if (q != (uint32_t)p) { const uint32_t r = ptr[q]; if (r != q) { ptr[p] = r; mv[I_] = 1; } }
            }
        }
    })
    return BALLOT(mv[I_]) != 0;
}
// E.  out[p] <- out[ptr[p]] for the 64 x 4 bytes from p0 on (a pointer leads to a literal now, and literals are never written here)
DEV void dx_gather(uint8_t* __restrict__ out, const uint32_t* __restrict__ ptr, const int p0, const int outLen)
{
    LANES({
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + 4 * LANE + k;
            if (p < outLen) { const uint32_t q = ptr[p]; if (q != (uint32_t)p) out[p] = out[q]; }
        }
    })
}

}  // namespace plz4
