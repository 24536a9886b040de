// Lane-emulation harness: compiles plz4_amd/csrc/lz4_device.inl (the SAME source hipcc builds for gfx950)
// with -DPLZ4_EMU so that the kernel logic can be checked against the oracle on a machine without a GPU.
// Test infrastructure only: built into tests/emu/_build/, never loaded by plz4_amd, not a CPU fallback.
#define PLZ4_EMU 1
#include "../../plz4_amd/csrc/lz4_device.inl"
#include "../../plz4_amd/csrc/lz4_seq_device.inl"
#include "../../plz4_amd/csrc/lz4hc_device.inl"
#include <stdlib.h>

int plz4_emu_descending = 0;
static int plz4_emu_old_dict = 0;      // 1: every dictionary mode through the one-sequence-per-batch encoder (cross-check)

extern "C" {

void emu_set_descending(int d) { plz4_emu_descending = d; }
void emu_set_old_dict(int d) { plz4_emu_old_dict = d; }

// the fused encoder (lz4_device.inl): what the kernels run for blocks above 4 MiB and, in its external-segment mode, for
// dictionaries / linked blocks
int emu_encode_block_fused(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    static thread_local uint32_t lds[plz4::kHashBytes / 4];
    return plz4::wave_encode_block(src, n, dst, cap, lds);
}

// parse -> sizes -> scan -> write (lz4_seq_device.inl), stage by stage as the kernels run them; nseqOut: sequences found
int emu_encode_block_seq(const uint8_t* src, int n, uint8_t* dst, int cap, int* nseqOut)
{
    using namespace plz4;
    static thread_local uint32_t lds[kHashBytes / 4];
    if (n < 0 || n > kSeqMaxBlock) return -1;
    uint64_t* seq = (uint64_t*)malloc(((size_t)seq_capacity(n) + 1) * 8);       // + the dump entry
    int lastAnchor = 0;
    const int nseq = wave_parse_l1(src, n, lds, seq, &lastAnchor);
    {   // the duplex kernel's build of the parser (windows through an LDS scratch, lz4_seq_device.inl kLdsWin): same records
        static thread_local uint8_t scr[256 + 64];
        uint64_t* seq2 = (uint64_t*)malloc(((size_t)seq_capacity(n) + 1) * 8);
        int lastAnchor2 = 0;
        const int nseq2 = wave_parse_l1<1>(src, n, lds, seq2, &lastAnchor2, scr);
        bool same = nseq2 == nseq && lastAnchor2 == lastAnchor && (nseq <= 0 || memcmp(seq, seq2, (size_t)nseq * 8) == 0);
        const int nseq3 = wave_parse_l1<2>(src, n, lds, seq2, &lastAnchor2, nullptr);          // ... and through the lane exchange
        same = same && nseq3 == nseq && lastAnchor2 == lastAnchor && (nseq <= 0 || memcmp(seq, seq2, (size_t)nseq * 8) == 0);
        free(seq2);
        if (!same) { free(seq); return -999998; }
    }
    if (nseqOut) *nseqOut = nseq;
    const int nChunks = (nseq + kSeqChunk - 1) / kSeqChunk;
    uint32_t* cb = (uint32_t*)malloc((size_t)(nChunks + 1) * 4);
    uint32_t* co = (uint32_t*)malloc((size_t)(nChunks + 1) * 4);
    uint8_t* bk = (uint8_t*)malloc((size_t)seq_capacity(n) + 1);
    co[0] = 0;
    for (int c = 0; c < nChunks; ++c) cb[c] = seq_emit_sizes(src, seq, bk, nseq, c);
    const int total = seq_emit_scan(cb, co, nseq, lastAnchor, n, cap);
    if (total > 0) for (int c = 0; c < (nChunks ? nChunks : 1); ++c) seq_emit_write(src, n, seq, bk, nseq, lastAnchor, c, co[c], dst);
    free(seq); free(cb); free(co); free(bk);
    return total;
}

int emu_encode_block(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    if (n <= plz4::kSeqMaxBlock) return emu_encode_block_seq(src, n, dst, cap, nullptr);
    return emu_encode_block_fused(src, n, dst, cap);
}

// Both builds of the decoder's vector path: the LDS-staged one the record kernels run, and the one that copies through
// memory (dictionary / raw kernels).  They must agree byte for byte; -999999 flags a disagreement.
int emu_decode_block(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    static thread_local uint8_t lds[plz4::kDecLdsBytes];
    const int r1 = plz4::wave_decode_block<true>(src, n, dst, cap, nullptr, 0, lds);
    uint8_t* alt = (uint8_t*)malloc((size_t)(cap > 0 ? cap : 1) + 64);
    memset(alt, 0, (size_t)(cap > 0 ? cap : 1) + 64);
    const int r2 = plz4::wave_decode_block<false>(src, n, alt, cap);
    const bool same = (r1 == r2) && (r1 <= 0 || memcmp(dst, alt, (size_t)r1) == 0);
    free(alt);
    return same ? r1 : -999999;
}

// both forms of the digest (four lanes straight from memory; staged through LDS): they must agree
uint32_t emu_xxh32(const uint8_t* p, int n)
{
    static thread_local __attribute__((aligned(16))) uint8_t lds[4096];
    const uint32_t a = plz4::wave_xxh32(p, n), b = plz4::wave_xxh32_staged(p, n, lds);
    return a == b ? a : ~a;
}

// streaming content checksum: the pieces of `p` (piece k = lens[k] bytes) written one after the other, then Sum32
uint32_t emu_xxh32_stream(const uint8_t* p, const int* lens, int nPieces)
{
    plz4::XxhStream st;
    plz4::wave_xxh32_stream_reset(&st);
    for (int k = 0; k < nPieces; ++k) { plz4::wave_xxh32_stream_update(&st, p, lens[k]); p += lens[k]; }
    return plz4::xxh32_stream_sum(st);
}

int emu_compress_hc(const uint8_t* src, int n, uint8_t* dst, int cap, int level)
{
    static thread_local uint8_t* ws = nullptr;
    if (!ws) ws = (uint8_t*)malloc(plz4::kHcWorkBytes);
    plz4::HcWork w;
    w.hash = (uint32_t*)ws; w.chain = (uint16_t*)(ws + plz4::kHcHashEntries * 4);
    w.opt = (plz4::HcOpt*)(ws + plz4::kHcHashEntries * 4 + plz4::kHcChainEntries * 2); w.pre = nullptr; w.rank = nullptr; w.list = nullptr;
    return plz4::hc_compress(src, n, dst, cap, level, w);
}

static plz4::HcWork emu_hc_work(uint8_t* ws)
{
    plz4::HcWork w;
    w.hash = (uint32_t*)ws; w.chain = (uint16_t*)(ws + plz4::kHcHashEntries * 4);
    w.opt = (plz4::HcOpt*)(ws + plz4::kHcHashEntries * 4 + plz4::kHcChainEntries * 2); w.pre = nullptr; w.rank = nullptr; w.list = nullptr;
    return w;
}

// HC with an external segment (mode 1: seg[0..segLen) is laid out right before a copy of the block, as the kernels do) or
// under a dictionary context (mode 2: tables built by hc_prime_dict, block <= 4 KiB).
int emu_compress_hc_dict(const uint8_t* src, int n, uint8_t* dst, int cap, int level, const uint8_t* seg, int segLen, int mode)
{
    static thread_local uint8_t* ws = nullptr; static thread_local uint8_t* dws = nullptr;
    if (!ws) { ws = (uint8_t*)malloc(plz4::kHcWorkBytes); dws = (uint8_t*)malloc(plz4::kHcWorkBytes); }
    plz4::HcDict d; d.mode = mode; d.len = segLen; d.bytes = nullptr; d.hash = nullptr; d.chain = nullptr;
    if (mode == plz4::kHcExt) {
        uint8_t* cat = (uint8_t*)malloc((size_t)segLen + (size_t)n + 64);
        if (segLen) memcpy(cat, seg, (size_t)segLen);
        if (n) memcpy(cat + segLen, src, (size_t)n);
        const int r = plz4::hc_compress(cat + segLen, n, dst, cap, level, emu_hc_work(ws), d);
        free(cat);
        return r;
    }
    uint8_t* dcopy = (uint8_t*)malloc((size_t)segLen + 64);
    if (segLen) memcpy(dcopy, seg, (size_t)segLen);
    const plz4::HcWork dw = emu_hc_work(dws);
    plz4::hc_prime_dict(dcopy, segLen, level, dw);
    d.bytes = dcopy; d.hash = dw.hash; d.chain = dw.chain;
    const int r = plz4::hc_compress(src, n, dst, cap, level, emu_hc_work(ws), d);
    free(dcopy);
    return r;
}

int emu_encode_block_dict(const uint8_t* src, int n, uint8_t* dst, int cap, const uint8_t* dict, int dictSize, int mode,
                          const uint32_t* dictTable)
{
    static thread_local uint32_t lds[plz4::kHashBytes / 4];
    if (mode == plz4::kDictCtxLookup || plz4_emu_old_dict) {
        plz4::DictEnc dc{dict, dictSize, mode, dictTable};
        return plz4::wave_encode_block_dict(src, n, dst, cap, dc, lds);
    }
    // as the kernels do: the segment is laid out right before (a copy of) the block
    const bool seg = (mode == plz4::kDictLoad || mode == plz4::kDictCtxCopy);
    const int segLen = seg ? dictSize : 0;
    uint8_t* cat = (uint8_t*)malloc((size_t)segLen + (size_t)n + 64);
    if (segLen) memcpy(cat, dict, (size_t)segLen);
    if (n) memcpy(cat + segLen, src, (size_t)n);
    memset(cat + segLen + n, 0, 64);
    const int r = plz4::wave_encode_block_ext(cat + segLen, n, dst, cap, mode, segLen, dictTable, lds);
    free(cat);
    return r;
}

// With a dictionary too, both builds of the vector path must agree (the record kernels run the LDS-staged one).
int emu_decode_block_dict(const uint8_t* src, int n, uint8_t* dst, int cap, const uint8_t* dict, int dictSize)
{
    static thread_local uint8_t lds[plz4::kDecLdsBytes];
    const int r1 = plz4::wave_decode_block<true>(src, n, dst, cap, dict, dictSize, lds);
    uint8_t* alt = (uint8_t*)malloc((size_t)(cap > 0 ? cap : 1) + 64);
    memset(alt, 0, (size_t)(cap > 0 ? cap : 1) + 64);
    const int r2 = plz4::wave_decode_block<false>(src, n, alt, cap, dict, dictSize);
    const bool same = (r1 == r2) && (r1 <= 0 || memcmp(dst, alt, (size_t)r1) == 0);
    free(alt);
    return same ? r1 : -999999;
}

}

// ---- HC level 12 in its three device phases (lz4hc12_device.inl), back to back on the CPU.
#include "../../plz4_amd/csrc/lz4hc12_device.inl"
#include "../../plz4_amd/csrc/lz4hc_lazy_device.inl"
namespace {
struct H12Emu {
    uint8_t* padded; uint16_t* chain; uint32_t* rank; uint32_t* listBase; uint32_t* offsets; int n, nPos, nPad;
    plz4::Hc12Tabs tabs;
    // [skipLo, skipHi): positions that are never inserted (an external segment's last three)
    H12Emu(const uint8_t* src, int n_, int skipLo = 0, int skipHi = 0) : n(n_)
    {
        using namespace plz4;
        nPos = n - kMfLimit + 1 > 0 ? n - kMfLimit + 1 : 0;
        nPad = ((n > 0 ? n : 0) + 1 + 1023) / 1024 * 1024;
        padded = (uint8_t*)calloc((size_t)(n > 0 ? n : 0) + 64, 1);
        if (n > 0) memcpy(padded, src, (size_t)n);
        chain = (uint16_t*)malloc((size_t)nPad * 2 + 64);
        rank = (uint32_t*)malloc((size_t)nPad * 4);
        listBase = (uint32_t*)malloc(((size_t)nPad + 8) * 4);
        memset(listBase, 0xEE, 32);
        offsets = (uint32_t*)calloc(kHcHashEntries, 4);
        // what the histogram kernel does: how many positions each hash has, then where its run starts
        const int nIns = n >= 4 ? n - 3 : 0;
        for (int p = 0; p < nIns; ++p) if (!(p >= skipLo && p < skipHi)) offsets[hc12_hash(ld32u(padded + p))]++;
        uint32_t run = 0;
        for (int h = 0; h < kHcHashEntries; ++h) { const uint32_t c = offsets[h]; offsets[h] = run; run += c; }
        uint32_t* lastT = (uint32_t*)malloc(kHcHashEntries / 2 * 4); uint32_t* curT = (uint32_t*)malloc(kHcHashEntries / 2 * 4);
        hc12_build_lists(padded, n, offsets, chain, rank, listBase + 8, nPad, lastT, curT, skipLo, skipHi);
        free(lastT); free(curT);
        tabs.src = padded; tabs.chain = chain; tabs.rank = rank; tabs.list = listBase + 8;
    }
    ~H12Emu() { free(padded); free(chain); free(rank); free(listBase); free(offsets); }
    plz4::Hc12F plain(int p) const
    {
        using namespace plz4;
        Hc12Flat ch; ch.c = chain;
        Hc12Lane<Hc12Flat> L; L.init(padded, n, p, ch((uint32_t)p));
        while (!L.step(ch)) {}
        return L.result();
    }
    plz4::Hc12F walk(int p, long* trips) const                  // the kernel's search, one lane
    {
        using namespace plz4;
        if (p + 32 > n) return plain(p);                    // as the kernel: the last positions are the parser's
        Hc12SrcFlat sw; sw.g = padded;
        Hc12Walk<Hc12SrcFlat> Q; Q.init(tabs, sw, n, p, chain[p]);
        for (bool done = false; !done; ) {
            if (trips) trips[Q.phase]++;
            switch (Q.phase) {
            case kPhFilter:  done = Q.is_near() ? Q.filter_trip<true>(tabs, sw) : Q.filter_trip<false>(tabs, sw); break;
            case kPhCount:   Q.count_trip(tabs, sw); break;
            case kPhScan:    done = Q.scan_trip(tabs); break;
            case kPhRank:    Q.rank_trip(tabs); break;
            case kPhPattern: done = Q.pattern_trip(tabs, sw); break;
            default:         done = true; break;
            }
        }
        return Q.result();
    }
};
}
// ncEvery > 0: every ncEvery-th position is left to the parser's own search (the path for positions the search phase skips);
// nl: entries of the price table kept in "LDS" (small values exercise the global part).
namespace { int emu_emit_records(const uint8_t* src, int n, const uint64_t* seq, int nseq, int lastAnchor, uint8_t* dst, int cap); }
// maxSegs > 0: the parser walked in segments and stitched, records through the emit stage (what the kernels run up to 4 MiB)
extern "C" int emu_compress_hc12(const uint8_t* src, int n, uint8_t* dst, int cap, int ncEvery, int nl, int maxSegs, int minSeg)
{
    using namespace plz4;
    H12Emu E(src, n);
    Hc12F* F = (Hc12F*)malloc(sizeof(Hc12F) * (size_t)(E.nPos + 1));
    int skipUntil = 0;                  // as the search kernel: positions inside a match longer than the parser's "sufficient" are left out
    for (int p = 0; p < E.nPos; ++p) {
        if ((ncEvery > 0 && p % ncEvery == ncEvery - 1) || p < skipUntil) { F[p].len = kHc12NotComputed; F[p].off = 0; continue; }
        F[p] = E.walk(p, nullptr);
        if (F[p].len > kHc12Sufficient + 8 && p + F[p].len - kHc12Sufficient > skipUntil) skipUntil = p + F[p].len - kHc12Sufficient;
    }
    Hc12Ws w;
    w.nl = nl;
    w.ent = (Hc12Ent*)aligned_alloc(16, sizeof(Hc12Ent) * (size_t)((nl + 3) & ~3));
    w.gprice = (int*)malloc(4 * kHc12OptEntries); w.glitlen = (int*)malloc(4 * kHc12OptEntries); w.gmloff = (uint32_t*)malloc(4 * kHc12OptEntries);
    uint64_t seq[64]; w.seq = seq;
    int r;
    if (maxSegs > 0 && n <= kSeqMaxBlock) {
        const int segs = lz_segments(n, maxSegs, minSeg), segCap = lz_seg_cap(lz_seg_len(n, segs));
        uint64_t* rec = (uint64_t*)malloc((size_t)segs * segCap * 8), *bridge = (uint64_t*)malloc((size_t)segs * segCap * 8);
        uint64_t* starts = (uint64_t*)malloc((size_t)segs * kLzStarts * 8);
        LzSegMeta* meta = (LzSegMeta*)malloc(sizeof(LzSegMeta) * (size_t)segs);
        LzPiece* pieces = (LzPiece*)malloc(sizeof(LzPiece) * 2 * (size_t)segs);
        for (int j = segs - 1; j >= 0; --j) hc12_segment(E.padded, n, F, E.chain, w, segs, j, rec, meta, starts);
        int lastAnchor = 0;
        const int nseq = hc12_stitch(E.padded, n, F, E.chain, w, segs, rec, bridge, meta, starts, pieces, &lastAnchor);
        uint64_t* all = (uint64_t*)malloc(((size_t)n / 4 + 64) * 8);
        for (int k = 0; k < 2 * segs; ++k) for (int i = 0; i < pieces[k].cnt; ++i) all[pieces[k].dst + i] = pieces[k].src[i];
        r = emu_emit_records(E.padded, n, all, nseq, lastAnchor, dst, cap);
        free(all); free(rec); free(bridge); free(starts); free(meta); free(pieces);
    } else
    r = hc12_parse(E.padded, n, dst, cap, F, E.chain, w);
    free(w.ent); free(w.gprice); free(w.glitlen); free(w.gmloff);
    free(F);
    return r;
}

// diagnostics: which positions' search results does the level-12 parser use?  need[n] <- 1 for those (one walk over the block)
extern "C" int emu_hc12_need(const uint8_t* src, int n, uint8_t* need)
{
    using namespace plz4;
    memset(need, 0, (size_t)n);
    plz4_emu_f_need = need;
    uint8_t* dst = (uint8_t*)malloc((size_t)n + n / 255 + 64);
    const int r = emu_compress_hc12(src, n, dst, n + n / 255 + 16, 0, 1024, 0, 8192);
    free(dst);
    plz4_emu_f_need = nullptr;
    return r;
}

// The search as the kernel runs it (Hc12Walk: lists, phases) against the plain chain walk (Hc12Lane), every position of the
// block: returns the number of positions whose answers differ.  trips (optional, 10 counters): phase trips in all.
extern "C" int emu_hc12_search_check(const uint8_t* src, int n, long* trips)
{
    using namespace plz4;
    H12Emu E(src, n);
    int bad = 0, skipUntil = 0;
    for (int p = 0; p < E.nPos; ++p) {
        if (p < skipUntil) continue;
        const Hc12F a = E.plain(p), b = E.walk(p, trips);
        if (a.len != b.len || a.off != b.off) bad++;
        if (a.len > kHc12Sufficient + 8 && p + a.len - kHc12Sufficient > skipUntil) skipUntil = p + a.len - kHc12Sufficient;
    }
    return bad;
}

// HC levels 3..11 on the chain built up front (HcWork::pre): the parsers of lz4hc_device.inl without their own table inserts.
extern "C" int emu_compress_hc_pre(const uint8_t* src, int n, uint8_t* dst, int cap, int level)
{
    using namespace plz4;
    const int nPad = ((n > 0 ? n : 0) + 1 + 1023) / 1024 * 1024;
    uint8_t* padded = (uint8_t*)calloc((size_t)(n > 0 ? n : 0) + 64, 1);
    if (n > 0) memcpy(padded, src, (size_t)n);
    uint16_t* chain = (uint16_t*)malloc((size_t)nPad * 2 + 64);
    uint32_t* tab = (uint32_t*)malloc((size_t)kHcHashEntries * 4);
    hc12_build_chain(padded, n, chain, nPad, tab);
    static thread_local uint8_t* ws = nullptr;
    if (!ws) ws = (uint8_t*)malloc(kHcWorkBytes);
    HcWork w = emu_hc_work(ws);
    w.pre = chain;
    const int r = hc_compress(padded, n, dst, cap, level, w);
    free(tab); free(chain); free(padded);
    return r;
}

// ... and with the per-hash lists as well: the hash-chain levels look at up to 63 candidates per round (hc_find_wider_lists)
extern "C" int emu_compress_hc_lists(const uint8_t* src, int n, uint8_t* dst, int cap, int level)
{
    using namespace plz4;
    H12Emu E(src, n);
    static thread_local uint8_t* ws = nullptr;
    if (!ws) ws = (uint8_t*)malloc(kHcWorkBytes);
    HcWork w = emu_hc_work(ws);
    w.pre = E.chain; w.rank = E.rank; w.list = E.listBase + 8;
    return hc_compress(E.padded, n, dst, cap, level, w);
}

// ---- HC levels 3..9 as the kernels run them for independent blocks (lz4hc_lazy_device.inl): lists, the first search of every
// position (Hc12Walk with the level's parameters; ncEvery > 0: every ncEvery-th one left to the parser), the deciding parser
// that writes records over F, the emit stage without catch-up.
#include "../../plz4_amd/csrc/lz4hc_lazy_device.inl"
namespace {
int emu_emit_records(const uint8_t* src, int n, const uint64_t* seq, int nseq, int lastAnchor, uint8_t* dst, int cap)
{
    using namespace plz4;
    const int nChunks = (nseq + kSeqChunk - 1) / kSeqChunk;
    uint32_t* cb = (uint32_t*)malloc((size_t)(nChunks + 1) * 4);
    uint32_t* co = (uint32_t*)malloc((size_t)(nChunks + 1) * 4);
    co[0] = 0;
    for (int c = 0; c < nChunks; ++c) cb[c] = seq_emit_sizes<false>(src, seq, nullptr, nseq, c);
    const int total = seq_emit_scan(cb, co, nseq, lastAnchor, n, cap);
    if (total > 0) for (int c = 0; c < (nChunks ? nChunks : 1); ++c) seq_emit_write<false>(src, n, seq, nullptr, nseq, lastAnchor, c, co[c], dst);
    free(cb); free(co);
    return total;
}
}
extern "C" int emu_compress_hc_lazy(const uint8_t* src, int n, uint8_t* dst, int cap, int level, int maxSegs, int minSeg)
{
    using namespace plz4;
    if (n < 0 || n > kSeqMaxBlock) return -1;
    H12Emu E(src, n);
    static thread_local uint8_t* ows = nullptr;
    if (!ows) ows = (uint8_t*)malloc(kHcWorkBytes);
    HcWork w = emu_hc_work(ows);                      // (levels 10..11 keep their price table there)
    w.pre = E.chain; w.rank = E.rank; w.list = E.listBase + 8;
    // pass 1: every segment; pass 2: the bridges; pass 3: the pieces gathered into one array
    const int segs = lz_segments(n, maxSegs, minSeg), segCap = lz_seg_cap(lz_seg_len(n, segs));
    uint64_t* rec = (uint64_t*)malloc((size_t)segs * segCap * 8), *bridge = (uint64_t*)malloc((size_t)segs * segCap * 8);
    uint64_t* starts = (uint64_t*)malloc((size_t)segs * kLzStarts * 8);
    LzSegMeta* meta = (LzSegMeta*)malloc(sizeof(LzSegMeta) * (size_t)segs);
    LzPiece* pieces = (LzPiece*)malloc(sizeof(LzPiece) * 2 * (size_t)segs);
    for (int j = segs - 1; j >= 0; --j) hc_lazy_segment<false>(E.padded, n, level, w, segs, j, rec, meta, starts);
    int lastAnchor = 0;
    const int nseq = hc_lazy_stitch<false>(E.padded, n, level, w, segs, rec, bridge, meta, starts, pieces, &lastAnchor);
    uint64_t* seq = (uint64_t*)malloc(((size_t)n / 4 + 64) * 8);
    for (int k = 0; k < 2 * segs; ++k) for (int i = 0; i < pieces[k].cnt; ++i) seq[pieces[k].dst + i] = pieces[k].src[i];
    const int r = emu_emit_records(E.padded, n, seq, nseq, lastAnchor, dst, cap);
    free(seq); free(rec); free(bridge); free(starts); free(meta); free(pieces);
    return r;
}

// HC levels 3..12 behind an external segment as the kernels run them (a linked block, or a block > 4 KiB under an attached
// dictionary): chain and lists over segment + block (the segment's last three positions left out), the level's walk in segments
// with the segment-aware finders, stitched, records through the emit stage.
extern "C" int emu_compress_hc_lazy_ext(const uint8_t* src, int n, uint8_t* dst, int cap, int level, const uint8_t* seg, int segLen, int maxSegs, int minSeg)
{
    using namespace plz4;
    if (n < 0 || n > kSeqMaxBlock || segLen < 0 || segLen > 65536) return -1;
    uint8_t* cat = (uint8_t*)calloc((size_t)segLen + (size_t)n + 64, 1);
    if (segLen) memcpy(cat, seg, (size_t)segLen);
    if (n) memcpy(cat + segLen, src, (size_t)n);
    const int N = segLen + n;
    H12Emu E(cat, N, segLen > 3 ? segLen - 3 : 0, segLen);
    static thread_local uint8_t* ows = nullptr;
    if (!ows) ows = (uint8_t*)malloc(kHcWorkBytes);
    HcWork w = emu_hc_work(ows);
    w.pre = E.chain; w.rank = E.rank; w.list = E.listBase + 8;
    const uint8_t* const blk = E.padded + segLen;
    const int segs = lz_segments(n, maxSegs, minSeg), segCap = lz_seg_cap(lz_seg_len(n, segs));
    uint64_t* rec = (uint64_t*)malloc((size_t)segs * segCap * 8), *bridge = (uint64_t*)malloc((size_t)segs * segCap * 8);
    uint64_t* starts = (uint64_t*)malloc((size_t)segs * kLzStarts * 8);
    LzSegMeta* meta = (LzSegMeta*)malloc(sizeof(LzSegMeta) * (size_t)segs);
    LzPiece* pieces = (LzPiece*)malloc(sizeof(LzPiece) * 2 * (size_t)segs);
    for (int j = segs - 1; j >= 0; --j) hc_lazy_segment<true>(blk, n, level, w, segs, j, rec, meta, starts, segLen);
    int lastAnchor = 0;
    const int nseq = hc_lazy_stitch<true>(blk, n, level, w, segs, rec, bridge, meta, starts, pieces, &lastAnchor, segLen);
    uint64_t* seq = (uint64_t*)malloc(((size_t)n / 4 + 64) * 8);
    for (int k = 0; k < 2 * segs; ++k) for (int i = 0; i < pieces[k].cnt; ++i) seq[pieces[k].dst + i] = pieces[k].src[i];
    const int r = emu_emit_records(blk, n, seq, nseq, lastAnchor, dst, cap);
    free(seq); free(rec); free(bridge); free(starts); free(meta); free(pieces); free(cat);
    return r;
}

// level 2 as the kernels run it for independent blocks: the batch walk over the two tables, records, emit
extern "C" int emu_compress_hc_mid(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    using namespace plz4;
    if (n < 0 || n > kSeqMaxBlock) return -1;
    uint8_t* padded = (uint8_t*)calloc((size_t)(n > 0 ? n : 0) + 64, 1);
    if (n > 0) memcpy(padded, src, (size_t)n);
    uint32_t* tabs = (uint32_t*)malloc(32768 * 4);
    uint64_t* seq = (uint64_t*)malloc(((size_t)n / 4 + 64) * 8);
    int lastAnchor = 0;
    const int nseq = hc_mid_parse(padded, n, tabs, tabs + 16384, seq, &lastAnchor);
    const int r = emu_emit_records(padded, n, seq, nseq, lastAnchor, dst, cap);
    free(seq); free(tabs); free(padded);
    return r;
}

// ... and behind an external segment (a linked block, a block > 4 KiB under an attached dictionary)
extern "C" int emu_compress_hc_mid_ext(const uint8_t* src, int n, uint8_t* dst, int cap, const uint8_t* seg, int segLen)
{
    using namespace plz4;
    if (n < 0 || n > kSeqMaxBlock || segLen < 0 || segLen > 65536) return -1;
    uint8_t* cat = (uint8_t*)calloc((size_t)segLen + (size_t)n + 64, 1);
    if (segLen) memcpy(cat, seg, (size_t)segLen);
    if (n) memcpy(cat + segLen, src, (size_t)n);
    uint32_t* tabs = (uint32_t*)malloc(32768 * 4);
    uint64_t* seq = (uint64_t*)malloc(((size_t)n / 4 + 64) * 8);
    int lastAnchor = 0;
    const int nseq = hc_mid_parse<true>(cat + segLen, n, tabs, tabs + 16384, seq, &lastAnchor, segLen);
    const int r = emu_emit_records(cat + segLen, n, seq, nseq, lastAnchor, dst, cap);
    free(seq); free(tabs); free(cat);
    return r;
}

// ---- the few-block decoder (lz4_dx_device.inl) as the kernels run it: tables per input segment, stitch, pointer fill per unit,
// jump rounds, gather.  Returns the decoded size; -999999: the block is left to the one-wave decoder (any anomaly); -888888: the
// units disagree about where they meet (a bug).
#include "../../plz4_amd/csrc/lz4_dx_device.inl"
extern "C" int emu_dx_decode(const uint8_t* src, int n, uint8_t* dst, int cap, int* roundsOut)
{
    using namespace plz4;
    if (roundsOut) *roundsOut = 0;
    uint8_t* in = (uint8_t*)calloc((size_t)(n > 0 ? n : 0) + 64, 1);
    if (n > 0) memcpy(in, src, (size_t)n);
    const int nseg = dx_segments(n), jt = dx_tail_from(nseg);
    uint64_t* T = (uint64_t*)calloc((size_t)(n > 0 ? n : 1) + 64, 8);
    DxUnit* units = (DxUnit*)calloc((size_t)nseg + 1, sizeof(DxUnit));
    const int capw = cap > 0 ? cap : 1;
    uint32_t* ptr = (uint32_t*)malloc(((size_t)capw + 64) * 4);
    for (int p = 0; p < capw + 64; ++p) ptr[p] = (uint32_t)p;
    int result = -999999;
    for (int j = nseg - 1; j >= 0; --j) dx_segment_table(in, n, j, T);                  // (any order: segments are independent)
    if (dx_stitch(in, n, cap, T, units, nseg) == 0) {
        bool bad = false; int64_t outLen = -1;
        for (int j = jt; j >= 0 && !bad; --j) {
            if (j < jt && units[j].ip < 0) continue;
            const int64_t r = wave_dx_fill(in, n, dst, cap, ptr, units[j].ip, units[j].op, units[j].stop, j == jt);
            if (r < 0) { bad = true; break; }
            if (j == jt) outLen = r;
            else {                                                                   // where this unit stops is where the next one starts
                int k = j + 1; while (k < jt && units[k].ip < 0) ++k;
                if (units[k].op != (int)r) { result = -888888; bad = true; }
            }
        }
        if (!bad) {
            int rounds = 0;
            for (; rounds < kDxRounds; ++rounds) {
                bool moved = false;
                // (from the top down: a pointer never sees one that was already moved in this round -- the slowest the kernel's
                // unordered workgroups can be)
                for (int p0 = (((int)outLen - 1) / 256) * 256; p0 >= 0; p0 -= 256) moved |= dx_jump(ptr, p0, (int)outLen);
                if (!moved) break;
            }
            if (roundsOut) *roundsOut = rounds;
            for (int p0 = 0; p0 < (int)outLen; p0 += 256) dx_gather(dst, ptr, p0, (int)outLen);
            result = (int)outLen;
        }
    }
    free(in); free(T); free(units); free(ptr);
    return result;
}

// diagnostics of the level-1 parser's pipeline (see plz4_emu_cnt in lz4_seq_device.inl); reset on read
extern "C" void emu_parse_counters(unsigned long long* out8)
{
    for (int i = 0; i < 8; ++i) { out8[i] = plz4::plz4_emu_cnt[i]; plz4::plz4_emu_cnt[i] = 0; }
}
