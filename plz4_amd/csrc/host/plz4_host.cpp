// plz4_host.cpp -- see plz4_host.hpp.  Plain C++ (no HIP in this file); the product engine is a thin wrapper over the
// C ABI of include/plz4hip.h.
#include "plz4_host.hpp"

#include <algorithm>
#include <cstring>
#include <deque>

#include "../../../include/plz4hip.h"

namespace plz4h {

// Byte buffers of the block pipeline: resize() must not zero-fill 4 MiB that is about to be overwritten
template <class T> struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    template <class U, class... A> void construct(U* p, A&&... a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
using Bytes = std::vector<uint8_t, NoInit<uint8_t>>;


// ------------------------------------------------------------------------------------------------ errors
const char* ErrorString(int c)
{
    static const char* n[] = {
        "ok", "lz4 end mark", "lz4 closed", "lz4 corrupted", "lz4 header hash mismatch", "lz4 block hash mismatch",
        "lz4 content hash mismatch", "lz4 fail read header", "lz4 fail write header", "lz4 bad magic",
        "lz4 unsupported version", "lz4 fail read descriptor", "lz4 fail read block size", "lz4 fail read block",
        "lz4 block size overflow", "lz4 fail compress", "lz4 fail decompress", "lz4 reserved bit set",
        "lz4 invalid BD byte", "lz4 fail read content hash", "lz4 content size mismatch", "lz4 bad read offset",
        "lz4 read offset unsupported in block linked mode", "lz4 fail skip", "lz4 bad nibble", "lz4 unsupported feature",
        "io error", "EOF", "block engine failure" };
    return (c >= 0 && c < (int)(sizeof n / sizeof n[0])) ? n[c] : "?";
}
static Error E(int c, bool corrupted = false) { Error e; e.code = c; e.corrupted = corrupted; return e; }

// ------------------------------------------------------------------------------------------------ options
int BlockIdxSize(int idx)
{
    switch (idx) { case 4: return 64 << 10; case 5: return 256 << 10; case 6: return 1 << 20; case 7: return 4 << 20; }
    return 0;
}
int Options::CalcPending() const                                                  // opts/opts.go:62-95
{
    if (PendingSz == 0) return NParallel;
    if (PendingSz == -1) {
        int mult = 2;
        if (BlockSizeIdx == BlockIdx64KB) mult = 16; else if (BlockSizeIdx == BlockIdx256KB) mult = 8; else if (BlockSizeIdx == BlockIdx1MB) mult = 4;
        return mult * NParallel;
    }
    int nPending = NParallel;
    const int nCap = PendingSz / BlockIdxSize(BlockSizeIdx);
    if (nCap > nPending) nPending = nCap;
    return nPending;
}
static int batch_depth(const Options& o)
{
    if (o.GpuBatchBlocks > 0) return o.GpuBatchBlocks;
    return std::max(o.CalcPending(), 64);
}

// ------------------------------------------------------------------------------------------------ xxHash32 (xxh32zero.go)
static const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
static inline uint32_t rl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t xfin(uint32_t h, const uint8_t* p, size_t rem)
{
    for (; rem >= 4; p += 4, rem -= 4) h = rl(h + le32(p) * P3, 17) * P4;
    for (; rem; ++p, --rem) h = rl(h + (uint32_t)*p * P5, 11) * P1;
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
void Xxh32Stream::Reset() { acc[0] = P1 + P2; acc[1] = P2; acc[2] = 0; acc[3] = 0u - P1; total = 0; fill = 0; }
void Xxh32Stream::Write(const uint8_t* p, size_t n)
{
    if (total == 0) Reset();
    total += n;
    if (fill) {
        const size_t need = 16 - fill;
        if (n < need) { memcpy(buf + fill, p, n); fill += (uint32_t)n; return; }
        memcpy(buf + fill, p, need); p += need; n -= need;
        for (int i = 0; i < 4; i++) acc[i] = rl(acc[i] + le32(buf + 4 * i) * P2, 13) * P1;
        fill = 0;
    }
    for (; n >= 16; p += 16, n -= 16)
        for (int i = 0; i < 4; i++) acc[i] = rl(acc[i] + le32(p + 4 * i) * P2, 13) * P1;
    if (n) { memcpy(buf, p, n); fill = (uint32_t)n; }
}
uint32_t Xxh32Stream::Sum32() const
{
    uint32_t h = (uint32_t)total;
    if (h >= 16 || total >= 0x100000000ull) h += rl(acc[0], 1) + rl(acc[1], 7) + rl(acc[2], 12) + rl(acc[3], 18);
    else h += P5;
    return xfin(h, buf, fill);
}
uint32_t Xxh32(const uint8_t* p, size_t n) { Xxh32Stream s; s.Reset(); s.total = 0; if (n) s.Write(p, n); else s.Reset(); return s.Sum32(); }

// ------------------------------------------------------------------------------------------------ header (header/write.go:23-73)
int WriteHeaderBytes(const Options& o, uint8_t out[19])
{
    int len = 7;
    uint8_t flg = 1u << 6;
    out[0] = 0x04; out[1] = 0x22; out[2] = 0x4d; out[3] = 0x18;
    if (!o.BlockLinked)    flg |= 1u << 5;
    if (o.BlockChecksum)   flg |= 1u << 4;
    if (o.ContentChecksum) flg |= 1u << 2;
    if (o.HasContentSz) { flg |= 1u << 3; len = 15; for (int i = 0; i < 8; i++) out[6 + i] = (uint8_t)(o.ContentSz >> (8 * i)); }
    if (o.HasDictionaryId) { flg |= 1u; put32(out + len - 1, o.DictionaryId); len += 4; }
    out[4] = flg;
    out[5] = (uint8_t)((o.BlockSizeIdx & 7) << 4);
    out[len - 1] = (uint8_t)((Xxh32(out + 4, (size_t)len - 5) >> 8) & 0xFF);
    return len;
}

// ------------------------------------------------------------------------------------------------ engines
namespace {
struct HipEngine : BlockEngine {
    plz4hip_ctx* ctx = nullptr;
    ~HipEngine() override { plz4hip_ctx_destroy(ctx); }
    int CompressBatch(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, int lvl, int32_t* r) override
    { return plz4hip_compress_batch(ctx, n, s, sl, d, dc, lvl, r); }
    int DecompressBatch(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, int32_t* r) override
    { return plz4hip_decompress_batch(ctx, n, s, sl, d, dc, r); }
    int EncodeRecords(int n, const void* const* s, const int32_t* sl, int bsz, int lvl, int bc, void* const* rec, int32_t* rl_) override
    { return plz4hip_encode_records(ctx, n, s, sl, bsz, lvl, bc, rec, rl_); }
    int DecodeRecords(int n, const void* const* rec, const int32_t* rl_, int bsz, int bc, void* const* d, int32_t* r, int32_t* st) override
    { return plz4hip_decode_records(ctx, n, rec, rl_, bsz, bc, d, r, st); }
    void* DictCreate(const uint8_t* p, int n) override { plz4hip_dict* d = nullptr; return plz4hip_dict_create(ctx, p, n, &d) == 0 ? d : nullptr; }
    void  DictDestroy(void* d) override { plz4hip_dict_destroy(ctx, (plz4hip_dict*)d); }
    int CompressBatchDict(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, int lvl, void* dict, int32_t* r) override
    { return plz4hip_compress_batch_dict(ctx, n, s, sl, d, dc, lvl, (plz4hip_dict*)dict, r); }
    int DecompressBatchDict(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, void* dict, int32_t* r) override
    { return plz4hip_decompress_batch_dict(ctx, n, s, sl, d, dc, (plz4hip_dict*)dict, r); }
    int EncodeRecordsEx(int n, const void* const* s, const int32_t* sl, int bsz, int lvl, int bc, int linked, void* dict,
                        const void* pt, int ptl, void* const* rec, int32_t* rl_) override
    { return plz4hip_encode_records_ex(ctx, n, s, sl, bsz, lvl, bc, linked, (plz4hip_dict*)dict, pt, ptl, rec, rl_); }
    int DecodeRecordsEx(int n, const void* const* rec, const int32_t* rl_, int bsz, int bc, int linked, void* dict, void* win, int* wl,
                        void* const* d, int32_t* r, int32_t* st) override
    { return plz4hip_decode_records_ex(ctx, n, rec, rl_, bsz, bc, linked, (plz4hip_dict*)dict, win, wl, d, r, st); }
    void* HashNew() override { plz4hip_xxh32_stream* h = nullptr; return plz4hip_xxh32_stream_create(ctx, &h) == 0 ? h : nullptr; }
    void  HashFree(void* h) override { plz4hip_xxh32_stream_destroy(ctx, (plz4hip_xxh32_stream*)h); }
    int   HashAttach(void* h) override { return plz4hip_ctx_set_content_hash(ctx, (plz4hip_xxh32_stream*)h); }
    int   HashSum(void* h, uint32_t* out) override { return plz4hip_xxh32_stream_sum(ctx, (plz4hip_xxh32_stream*)h, out); }
    int   HashReset(void* h) override { return plz4hip_xxh32_stream_reset(ctx, (plz4hip_xxh32_stream*)h); }
};
struct VtEngine : BlockEngine {
    EngineVTable vt;
    int CompressBatch(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, int lvl, int32_t* r) override
    { return vt.compress_batch(vt.user, n, s, sl, d, dc, lvl, r); }
    int DecompressBatch(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, int32_t* r) override
    { return vt.decompress_batch(vt.user, n, s, sl, d, dc, r); }
    int EncodeRecords(int n, const void* const* s, const int32_t* sl, int bsz, int lvl, int bc, void* const* rec, int32_t* rl_) override
    { return vt.encode_records(vt.user, n, s, sl, bsz, lvl, bc, rec, rl_); }
    int DecodeRecords(int n, const void* const* rec, const int32_t* rl_, int bsz, int bc, void* const* d, int32_t* r, int32_t* st) override
    { return vt.decode_records(vt.user, n, rec, rl_, bsz, bc, d, r, st); }
    void* DictCreate(const uint8_t* p, int n) override { return vt.dict_create ? vt.dict_create(vt.user, p, n) : nullptr; }
    void  DictDestroy(void* d) override { if (vt.dict_destroy) vt.dict_destroy(vt.user, d); }
    int CompressBatchDict(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, int lvl, void* dict, int32_t* r) override
    { return vt.compress_batch_dict ? vt.compress_batch_dict(vt.user, n, s, sl, d, dc, lvl, dict, r) : -4; }
    int DecompressBatchDict(int n, const void* const* s, const int32_t* sl, void* const* d, const int32_t* dc, void* dict, int32_t* r) override
    { return vt.decompress_batch_dict ? vt.decompress_batch_dict(vt.user, n, s, sl, d, dc, dict, r) : -4; }
    int EncodeRecordsEx(int n, const void* const* s, const int32_t* sl, int bsz, int lvl, int bc, int linked, void* dict,
                        const void* pt, int ptl, void* const* rec, int32_t* rl_) override
    { return vt.encode_records_ex ? vt.encode_records_ex(vt.user, n, s, sl, bsz, lvl, bc, linked, dict, pt, ptl, rec, rl_) : -4; }
    int DecodeRecordsEx(int n, const void* const* rec, const int32_t* rl_, int bsz, int bc, int linked, void* dict, void* win, int* wl,
                        void* const* d, int32_t* r, int32_t* st) override
    { return vt.decode_records_ex ? vt.decode_records_ex(vt.user, n, rec, rl_, bsz, bc, linked, dict, win, wl, d, r, st) : -4; }
};
}  // namespace
std::unique_ptr<BlockEngine> NewHipEngine(int device, int* rc)
{
    std::unique_ptr<HipEngine> e(new HipEngine());
    const int r = plz4hip_ctx_create(device, &e->ctx);
    if (rc) *rc = r;
    if (r != PLZ4HIP_OK) return nullptr;
    return e;
}
std::unique_ptr<BlockEngine> NewVTableEngine(const EngineVTable& vt) { std::unique_ptr<VtEngine> e(new VtEngine()); e->vt = vt; return e; }

// ------------------------------------------------------------------------------------------------ Writer
namespace {
class WriterImpl : public Writer {
    Sink& wr; BlockEngine& eng; Options o;
    const int  bsz;
    const bool sync;                         // NParallel == 0 (sync/writer.go); else async/writer.go semantics
    // A source block handed to the "workers", not yet emitted: its bytes are either the writer's own (a pooled buffer) or,
    // inside one Write call, still the caller's (full blocks of a large Write are encoded straight from the caller's memory;
    // what has not been drained when Write returns is copied, io.Writer must not keep p).
    struct Blk { const uint8_t* p = nullptr; size_t n = 0; uint8_t* own = nullptr; };
    std::vector<uint8_t*> pool;              // free block buffers (bsz bytes each, never zero-filled)
    uint8_t* cur = nullptr; size_t curLen = 0;   // srcBlk / srcOff
    std::deque<Blk> queue;
    uint8_t* recArena = nullptr; size_t recArenaBlocks = 0;   // record buffers of a batch, reused
    void* devHash = nullptr;                 // the engine's content-checksum stream (null: hash on the host)
    uint8_t* getBuf() { if (!pool.empty()) { uint8_t* b = pool.back(); pool.pop_back(); return b; } return (uint8_t*)malloc((size_t)bsz); }
    void putBuf(uint8_t* b) { if (b) pool.push_back(b); }
    void clearQueue() { for (auto& b : queue) putBuf(b.own); queue.clear(); }
    void ownQueue() { for (auto& b : queue) if (!b.own) { b.own = getBuf(); memcpy(b.own, b.p, b.n); b.p = b.own; } }
    bool headerDone = false, kicked = false, closed = false, reported = false;
    Error state;                                            // first error wins (async/writer.go:552-555)
    int64_t srcMark = 0, dstMark = 0;
    Xxh32Stream hasher; bool hashing;
    void* dictH = nullptr; bool dictTried = false;          // compress.NewCompressorFactory(level, independent, dict)
    std::vector<uint8_t> lastTail; bool haveTail = false;   // linked: last <= 64 KiB of the previous block (async/writer.go:412-437)
    void progress(int64_t a, int64_t b) { if (o.Handler) o.Handler(a, b); }
    bool exMode() const { return o.BlockLinked || o.HasDictionary; }
    Error encode(int n, const void* const* src, const int32_t* len, void* const* rec, int32_t* rlen)
    {
        int rc;
        struct Attach { BlockEngine& e; void* h; Attach(BlockEngine& e_, void* h_) : e(e_), h(h_) { if (h) e.HashAttach(h); } ~Attach() { if (h) e.HashAttach(nullptr); } } at(eng, devHash);
        if (!exMode()) rc = eng.EncodeRecords(n, src, len, bsz, o.Level, o.BlockChecksum ? 1 : 0, rec, rlen);
        else {
            if (o.HasDictionary && !dictTried) { dictTried = true; dictH = eng.DictCreate(o.Dictionary.data(), (int)o.Dictionary.size()); if (!dictH) return E(ErrUnsupported); }
            rc = eng.EncodeRecordsEx(n, src, len, bsz, o.Level, o.BlockChecksum ? 1 : 0, o.BlockLinked ? 1 : 0, dictH,
                                     haveTail ? (const void*)lastTail.data() : nullptr, haveTail ? (int)lastTail.size() : -1, rec, rlen);
            if (o.BlockLinked && n > 0) {
                const uint8_t* lp = (const uint8_t*)src[n - 1]; const int ll = len[n - 1]; const int k = ll < 65536 ? ll : 65536;
                lastTail.assign(lp + (ll - k), lp + ll); haveTail = true;
            }
        }
        if (rc != 0) return E(rc == PLZ4HIP_E_UNSUPPORTED ? ErrUnsupported : ErrEngine);
        return Error();
    }
    Error sinkWrite(const uint8_t* p, size_t n, size_t* w) { size_t ww = 0; Error e = wr.write(p, n, &ww); if (w) *w = ww; return e; }
    Error writeHeader()
    {
        uint8_t h[19]; const int n = WriteHeaderBytes(o, h);
        size_t w = 0; Error e = sinkWrite(h, (size_t)n, &w);
        headerDone = true;
        if (e) return E(e.code == ErrIO ? ErrHeaderWrite : e.code);   // errors.Join(ErrHeaderWrite, err), header/write.go:67-70
        dstMark = n;
        return Error();
    }
    Error supported() const
    {
        if (o.Level < 1 || o.Level > 12) return E(ErrUnsupported);   // WithLevel clamps to [1,12] (plz4_opts.go:139-149)
        return Error();
    }
    // blk.CompressToBlk x n on the engine, then in-order emission (async/writer.go:284-381 writeLoop)
    Error drain()
    {
        if (queue.empty()) return Error();
        if (Error e = supported()) return e;
        if (!headerDone) if (Error e = writeHeader()) return e;
        const int n = (int)queue.size();
        std::vector<const void*> src(n); std::vector<int32_t> len(n), rlen(n); std::vector<void*> rec(n);
        const size_t recStride = (size_t)bsz + 8;
        if ((size_t)n > recArenaBlocks) { free(recArena); recArena = (uint8_t*)malloc((size_t)n * recStride); recArenaBlocks = recArena ? (size_t)n : 0; if (!recArena) { clearQueue(); return E(ErrEngine); } }
        struct RecView { uint8_t* p; uint8_t& operator[](size_t i) const { return p[i]; } uint8_t* data() const { return p; } };
        std::vector<RecView> recs(n);
        for (int i = 0; i < n; i++) { src[i] = queue[i].p; len[i] = (int32_t)queue[i].n; recs[i].p = recArena + (size_t)i * recStride; rec[i] = recs[i].p; }
        if (Error ee = encode(n, src.data(), len.data(), rec.data(), rlen.data())) { clearQueue(); return ee; }
        Error err;
        // linkedCompressorHC.Compress hands liblz4's "does not fit" back WITHOUT joining zerr.ErrCompress
        // (compress/linked.go:47-49, unlike every other compressor), so CompressToBlk does not fall back to a stored block
        // there (blk/blk.go:75-86): a linked frame above level 1 fails at its first incompressible block.  Kept as it is.
        const bool storedIsFatal = o.BlockLinked && o.Level > 1;
        for (int i = 0; i < n && !err; i++) {
            size_t w = 0;
            if (storedIsFatal && (recs[i][3] & 0x80)) { err = E(ErrCompress); break; }
            Error e = sinkWrite(recs[i].data(), (size_t)rlen[i], &w);
            progress(srcMark, dstMark);
            srcMark += len[i]; dstMark += (int64_t)w;
            if (e) err = e;
        }
        clearQueue();
        return err;
    }
    void enqueue(const uint8_t* p, size_t n, uint8_t* own)
    {
        if (hashing && !devHash) hasher.Write(p, n);
        Blk b; b.p = p; b.n = n; b.own = own;
        queue.push_back(b); kicked = true;
    }
    void enqueueCur() { enqueue(cur, curLen, cur); cur = nullptr; curLen = 0; }
    Error maybeDrain(bool force) { if (force || (int)queue.size() >= batch_depth(o)) return drain(); return Error(); }
    Error latch(Error e) { if (e && !state) state = e; return state; }
    Error report() { if (state) reported = true; return state; }
    Error trailer()
    {
        uint8_t t[8] = {0};
        size_t n = 4;
        if (hashing) {
            uint32_t sum = 0;
            if (devHash) { if (eng.HashSum(devHash, &sum) != 0) return E(ErrEngine); }
            else { if (hasher.total == 0) hasher.Reset(); sum = hasher.Sum32(); }
            put32(t + 4, sum); n = 8;
        }
        return sinkWrite(t, n, nullptr);
    }

public:
    ~WriterImpl() override
    {
        if (dictH) eng.DictDestroy(dictH);
        if (devHash) eng.HashFree(devHash);
        clearQueue(); putBuf(cur);
        for (uint8_t* b : pool) free(b);
        free(recArena);
    }
    WriterImpl(Sink& w, BlockEngine& e, const Options& op)
        : wr(w), eng(e), o(op), bsz(BlockIdxSize(op.BlockSizeIdx)), sync(op.NParallel == 0 && !op.BlockLinked), hashing(op.ContentChecksum)
    {
        hasher.Reset();                                      // linked => async (plz4_writer.go:44-46)
        if (hashing) devHash = eng.HashNew();                // the engine keeps the content checksum if it can (async/hash.go:99-111)
        if (!sync && o.HasContentSz && (int)(o.ContentSz / (uint64_t)bsz) + 1 > 1) kicked = true;   // async/writer.go:70-76
    }
    Error Write(const uint8_t* p, size_t n, size_t* consumed) override
    {
        size_t used = 0;
        if (consumed) *consumed = 0;
        if (state) return report();
        if (sync && !headerDone) { if (latch(writeHeader())) return report(); }            // sync/writer.go:64-68
        while (n > 0 && !state) {
            if (curLen == 0 && n >= (size_t)bsz) {                                         // a full block straight from the caller
                enqueue(p, (size_t)bsz, nullptr);
                p += bsz; n -= (size_t)bsz; used += (size_t)bsz;
                latch(maybeDrain(false));
                continue;
            }
            if (!cur) cur = getBuf();
            const size_t k = std::min(n, (size_t)bsz - curLen);
            memcpy(cur + curLen, p, k);
            curLen += k; p += k; n -= k; used += k;
            if (curLen == (size_t)bsz) {
                enqueueCur();
                latch(maybeDrain(false));
            }
        }
        if (sync && !state) latch(drain());                                                // the sync writer emits before returning
        ownQueue();                                                                        // nothing of p is kept past this call
        if (consumed) *consumed = used;
        return report();
    }
    Error ReadFrom(Source& r, int64_t* consumed) override
    {
        int64_t used = 0;
        if (consumed) *consumed = 0;
        if (state) return report();
        if (sync && !headerDone) { if (latch(writeHeader())) return report(); }
        while (!state) {
            if (!cur) cur = getBuf();
            size_t got = 0;
            Error e = r.read(cur + curLen, (size_t)bsz - curLen, &got);
            curLen += got; used += (int64_t)got;
            if (e && e.code != ErrEOF) { latch(e); break; }
            if (curLen == (size_t)bsz) { enqueueCur(); latch(maybeDrain(false)); continue; }
            if (got == 0 || e.code == ErrEOF) break;                                       // io.EOF: leave the partial block cached
        }
        if (sync && !state) latch(drain());
        if (consumed) *consumed = used;
        return report();
    }
    Error Flush() override
    {
        if (state) return report();
        if (curLen) enqueueCur();
        latch(drain());
        return report();
    }
    Error Close() override
    {
        if (closed) return sync ? state : report();
        closed = true;
        Error err;
        if (sync) {                                                                         // sync/writer.go:133-186
            if (!headerDone) err = writeHeader();
            else if (curLen) { enqueueCur(); err = drain(); }
            curLen = 0;
            if (state) return Error();                     // Close succeeds in an error state but writes no trailer
            if (err) { state = err; return err; }
            progress(srcMark, dstMark);
            err = trailer();
            state = err ? err : E(ErrClosed);
            return err;
        }
        // async (async/writer.go:135-191, :469-550)
        if (!state) {
            if (!kicked && curLen == 0) {
                // nothing was ever written: the async writer emits nothing at all
            } else if (!kicked) {
                // one sub-block payload closed before the pipeline started: _writeSync shortcut
                Bytes only(cur, cur + curLen); putBuf(cur); cur = nullptr; curLen = 0;
                Error e = supported();
                if (!e) e = writeHeader();
                const int64_t hdrSz = dstMark;
                if (!e) {
                    const void* s = only.data(); int32_t l = (int32_t)only.size(), rl_ = 0;
                    Bytes rec((size_t)bsz + 8); void* rp = rec.data();
                    e = encode(1, &s, &l, &rp, &rl_);
                    if (!e && o.BlockLinked && o.Level > 1 && (rec[3] & 0x80)) e = E(ErrCompress);    // see drain(): linked HC has no stored fallback
                    if (!e) e = sinkWrite(rec.data(), (size_t)rl_, nullptr);
                    if (!e) {
                        progress(0, hdrSz);
                        if (hashing && !devHash) { hasher.Reset(); hasher.total = 0; hasher.Write(only.data(), only.size()); }
                        e = trailer();
                    }
                }
                latch(e);
            } else {
                if (curLen) enqueueCur();
                Error e = drain();
                if (!e && !headerDone) e = writeHeader();
                if (!e) { progress(srcMark, dstMark); e = trailer(); }
                latch(e);
            }
        }
        if (reported) return Error();                       // already reported: Close returns nil
        if (!state) { state = E(ErrClosed); reported = true; return Error(); }
        return report();
    }
};
}  // namespace
std::unique_ptr<Writer> NewWriter(Sink& wr, BlockEngine& eng, const Options& o) { return std::unique_ptr<Writer>(new WriterImpl(wr, eng, o)); }

// ------------------------------------------------------------------------------------------------ Reader
namespace {
class ReaderImpl : public Reader {
    Source& rd; BlockEngine& eng; Options o;
    bool optsLive = true;
    Error state;
    // frame state
    bool inBody = false; HeaderT hdr; int bsz = 0; bool blkCheck = false, srcCheck = false, hashing = false;
    Xxh32Stream hasher; uint64_t contentSz = 0; uint32_t srcSum = 0;
    void* devHash = nullptr;                 // the engine's content-checksum stream (null: hash on the host)
    int64_t srcPos = 0, dstPos = 0;
    // decoded blocks ready for delivery + the error that follows them
    struct Out { Bytes data; int nRead; };
    std::deque<Out> ready; Error pendingErr; int pendingRead = 0;
    Bytes dstBlk; size_t dstOff = 0;
    bool linked = false; void* dictH = nullptr; bool dictTried = false;
    std::vector<uint8_t> window; int windowLen = 0;            // compress.DictT for linked frames

    void progress(int64_t a, int64_t b) { if (o.Handler) o.Handler(a, b); }
    Error readFull(uint8_t* p, size_t n, size_t* got)
    {
        size_t have = 0;
        while (have < n) {
            size_t g = 0; Error e = rd.read(p + have, n - have, &g);
            have += g;
            if (e && e.code != ErrEOF) { *got = have; return e; }
            if (g == 0 || e.code == ErrEOF) { *got = have; return E(ErrEOF); }
        }
        *got = have; return Error();
    }
    // header/read.go:26-119 (+ skip frames header/skip.go: size-prefixed, discarded)
    Error readHeader(int* nRead)
    {
        for (;;) {
            uint8_t b[19]; size_t got = 0;
            Error e = readFull(b, 7, &got); *nRead += (int)got;
            if (e) { if (e.code == ErrEOF && got == 0) return E(ErrEOF); return E(ErrHeaderRead); }
            if (!(b[0] == 0x04 && b[1] == 0x22 && b[2] == 0x4d && b[3] == 0x18)) {
                if ((b[0] & 0xF0) == 0x50 && b[1] == 0x2A && b[2] == 0x4D && b[3] == 0x18) {          // skippable frame (header/skip.go:39-76)
                    e = readFull(b + 7, 1, &got); *nRead += (int)got;                                   // 8th byte completes the size word
                    if (e) return E(ErrHeaderRead);
                    uint32_t left = le32(b + 4);
                    std::vector<uint8_t> junk(std::min<uint32_t>(std::max<uint32_t>(left, 1), 1u << 16));
                    while (left) { size_t g = 0; const size_t k = std::min<size_t>(left, junk.size()); Error s2 = readFull(junk.data(), k, &g); *nRead += (int)g; if (s2) return E(ErrSkip); left -= (uint32_t)k; }
                    continue;
                }
                return E(ErrMagic, true);
            }
            hdr = HeaderT(); hdr.Flags = b[4]; hdr.BlockDesc = b[5];
            if (((hdr.Flags >> 6) & 3) != 1) return E(ErrVersion);
            if (hdr.Flags & 2) return E(ErrReserveBitSet, true);
            if (((hdr.BlockDesc >> 4) & 7) < 4 || (hdr.BlockDesc & 0x80) || (hdr.BlockDesc & 0x0F)) return E(ErrBlockDescriptor, true);
            int len = 7;
            if (hdr.Flags & 8) { e = readFull(b + 7, 8, &got); *nRead += (int)got; if (e) return E(ErrHeaderRead); len = 15; hdr.ContentSz = 0; for (int i = 0; i < 8; i++) hdr.ContentSz |= (uint64_t)b[6 + i] << (8 * i); }
            if (hdr.Flags & 1) { e = readFull(b + len, 4, &got); *nRead += (int)got; if (e) return E(ErrHeaderRead); hdr.DictId = le32(b + len - 1); len += 4; }
            if (b[len - 1] != (uint8_t)((Xxh32(b + 4, (size_t)len - 5) >> 8) & 0xFF)) return E(ErrHeaderHash, true);
            hdr.Sz = len;
            return Error();
        }
    }
    Error startFrame(int* nRead)                                                           // rdr.go:242-296
    {
        Error e = readHeader(nRead);
        if (e) return e;
        bool clrContentChecksum = false;
        if (!(o.ReadOffset == 0 || o.ReadOffset == hdr.Sz)) {
            if (o.ReadOffset < hdr.Sz) return E(ErrReadOffset);
            if (!(hdr.Flags & 0x20)) return E(ErrReadOffsetLinked);
            int64_t left = o.ReadOffset - hdr.Sz;
            if (rd.skip(left)) { *nRead += (int)left; }
            else {
                std::vector<uint8_t> junk(1 << 16);
                while (left) { size_t g = 0; const size_t k = (size_t)std::min<int64_t>(left, (int64_t)junk.size()); Error s = readFull(junk.data(), k, &g); *nRead += (int)g; if (s) return s.code == ErrEOF ? E(ErrEOF) : s; left -= (int64_t)k; }
            }
            o.ReadOffset = 0; clrContentChecksum = true; o.SkipContentSz = true;
        }
        if (hdr.Flags & 8) { o.HasContentSz = true; o.ContentSz = hdr.ContentSz; }
        linked = !(hdr.Flags & 0x20);
        if (linked) {                                                                      // compress.NewDictT(opts.Dictionary, linked)
            window.assign(65536, 0); windowLen = 0;
            if (o.HasDictionary) { const size_t k = std::min<size_t>(o.Dictionary.size(), 65536); memcpy(window.data(), o.Dictionary.data() + (o.Dictionary.size() - k), k); windowLen = (int)k; }
        } else if (o.HasDictionary && !dictTried) {
            dictTried = true; dictH = eng.DictCreate(o.Dictionary.data(), (int)o.Dictionary.size());
            if (!dictH) return E(ErrUnsupported);
        }
        bsz = BlockIdxSize((hdr.BlockDesc >> 4) & 7);
        blkCheck = (hdr.Flags >> 4) & 1;
        srcCheck = ((hdr.Flags >> 2) & 1) && !clrContentChecksum;
        hashing = srcCheck && o.ContentChecksum;
        hasher.Reset(); contentSz = 0; inBody = true;
        if (hashing) {                                       // the engine keeps the content checksum if it can (one stream per reader)
            if (!devHash) devHash = eng.HashNew();
            else if (eng.HashReset(devHash) != 0) { eng.HashFree(devHash); devHash = nullptr; }
        }
        ready.clear(); pendingErr = Error(); pendingRead = 0; dstBlk.clear(); dstOff = 0;
        return Error();
    }
    // blk/frame.go:54-112 for up to `batch` records, then one engine call (async/reader.go:128-221 in batch form)
    void fill()
    {
        const int batch = batch_depth(o);
        std::vector<Bytes> recs; std::vector<int> reads;
        while ((int)recs.size() < batch && !pendingErr) {
            uint8_t w4[4]; size_t got = 0; int nRead = 0;
            Error e = readFull(w4, 4, &got); nRead += (int)got;
            if (e) { pendingErr = E(ErrBlockSizeRead); pendingRead = nRead; break; }
            const uint32_t word = le32(w4);
            if (word == 0) {                                                               // EOF mark (+ content hash)
                if (srcCheckRaw()) { e = readFull(w4, 4, &got); nRead += (int)got; if (e) { pendingErr = E(ErrContentHashRead); pendingRead = nRead; break; } srcSum = le32(w4); }
                pendingErr = E(EndMark); pendingRead = nRead; break;
            }
            const int64_t sz = word & 0x7FFFFFFFu;
            if (sz > bsz) { pendingErr = E(ErrBlockSizeOverflow, true); pendingRead = nRead; break; }
            const size_t total = (size_t)sz + (blkCheck ? 4 : 0);
            Bytes rec(4 + total);
            memcpy(rec.data(), w4, 4);
            e = readFull(rec.data() + 4, total, &got); nRead += (int)got;
            if (e) { pendingErr = E(ErrBlockRead); pendingRead = nRead; break; }
            recs.push_back(std::move(rec)); reads.push_back(nRead);
        }
        const int n = (int)recs.size();
        if (!n) return;
        std::vector<const void*> rp(n); std::vector<int32_t> rl_(n), res(n), st(n); std::vector<void*> dp(n);
        std::vector<Bytes> outs(n);
        for (int i = 0; i < n; i++) { rp[i] = recs[i].data(); rl_[i] = (int32_t)recs[i].size(); outs[i].resize((size_t)bsz + 8); dp[i] = outs[i].data(); }
        int rc;
        struct Attach { BlockEngine& e; void* h; Attach(BlockEngine& e_, void* h_) : e(e_), h(h_) { if (h) e.HashAttach(h); } ~Attach() { if (h) e.HashAttach(nullptr); } } at(eng, hashing ? devHash : nullptr);
        if (linked) rc = eng.DecodeRecordsEx(n, rp.data(), rl_.data(), bsz, blkCheck ? 1 : 0, 1, nullptr, window.data(), &windowLen, dp.data(), res.data(), st.data());
        else if (dictH) rc = eng.DecodeRecordsEx(n, rp.data(), rl_.data(), bsz, blkCheck ? 1 : 0, 0, dictH, nullptr, nullptr, dp.data(), res.data(), st.data());
        else rc = eng.DecodeRecords(n, rp.data(), rl_.data(), bsz, blkCheck ? 1 : 0, dp.data(), res.data(), st.data());
        if (rc != 0) { pendingErr = E(rc == PLZ4HIP_E_UNSUPPORTED ? ErrUnsupported : ErrEngine); pendingRead = 0; return; }
        for (int i = 0; i < n; i++) {
            if (st[i] != PLZ4HIP_BLK_OK) {
                // the first bad block ends the stream; anything parsed after it is dropped (first error wins)
                pendingErr = st[i] == PLZ4HIP_BLK_HASH_MISMATCH ? E(ErrBlockHash, true)
                           : st[i] == PLZ4HIP_BLK_SIZE_OVERFLOW ? E(ErrBlockSizeOverflow, true) : E(ErrDecompress, true);
                pendingRead = reads[i];
                return;
            }
            outs[i].resize((size_t)res[i]);
            ready.push_back(Out{std::move(outs[i]), reads[i]});
        }
    }
    bool srcCheckRaw() const { return (hdr.Flags >> 2) & 1; }
    // blkReader.NextBlock + rdr.nextBlock (rdr.go:207-227)
    Error nextBlock()
    {
        dstOff = 0; dstBlk.clear();
        if (ready.empty() && !pendingErr) fill();
        if (!ready.empty()) {
            Out o2 = std::move(ready.front()); ready.pop_front();
            if (hashing && !devHash) hasher.Write(o2.data.data(), o2.data.size());
            dstBlk = std::move(o2.data);
            progress(srcPos, dstPos);
            srcPos += o2.nRead; dstPos += (int64_t)dstBlk.size(); contentSz += dstBlk.size();
            return Error();
        }
        Error e = pendingErr; pendingErr = Error();
        progress(srcPos, dstPos);
        srcPos += pendingRead; pendingRead = 0;
        if (e.code == EndMark && hashing) {
            uint32_t sum = 0;
            if (devHash) { if (eng.HashSum(devHash, &sum) != 0) return E(ErrEngine); }
            else { if (hasher.total == 0) hasher.Reset(); sum = hasher.Sum32(); }
            if (sum != srcSum) e = E(ErrContentHash, true);
        }
        return e;
    }
    Error handleEndMark()                                                                  // rdr.go:91-101
    {
        Error e;
        if (o.HasContentSz && !o.SkipContentSz && o.ContentSz != contentSz) e = E(ErrContentSize, true);
        contentSz = 0; inBody = false;
        return e;
    }

public:
    ReaderImpl(Source& r, BlockEngine& e, const Options& op) : rd(r), eng(e), o(op) {}
    ~ReaderImpl() override { if (dictH) eng.DictDestroy(dictH); if (devHash) eng.HashFree(devHash); }
    Error Read(uint8_t* dst, size_t n, size_t* got) override
    {
        *got = 0;
        if (state) return state;
        for (;;) {
            Error err; size_t nRead = 0;
            if (!inBody) { int hn = 0; err = startFrame(&hn); srcPos += hn; }
            if (!err) {
                for (;;) {                                                                 // modeBody, rdr.go:298-330
                    if (dstOff < dstBlk.size()) {
                        const size_t k = std::min(n - nRead, dstBlk.size() - dstOff);
                        memcpy(dst + nRead, dstBlk.data() + dstOff, k); dstOff += k; nRead += k;
                        if (nRead == n) break;
                    }
                    if ((err = nextBlock())) break;
                }
            }
            *got = nRead;
            if (!err) return Error();
            if (err.code == EndMark) {
                err = handleEndMark();
                state = err;
                if (!err && nRead == 0 && n > 0) continue;                                 // never return (0, nil) at end of frame
                return err;
            }
            if (nRead > 0 && inBody) { state = err; return Error(); }                      // defer the error to the next call
            state = err;
            return err;
        }
    }
    Error WriteTo(Sink& w, int64_t* written) override
    {
        int64_t sum = 0;
        while (!state) {
            if (!inBody) {
                int hn = 0; state = startFrame(&hn); srcPos += hn;
                if (state) { if (state.code == ErrEOF) state = Error(); break; }
            }
            Error err;
            for (;;) {
                if (dstOff < dstBlk.size()) {
                    size_t k = 0; err = w.write(dstBlk.data() + dstOff, dstBlk.size() - dstOff, &k);
                    dstOff += k; sum += (int64_t)k;
                    if (err) break;
                }
                if ((err = nextBlock())) break;
            }
            state = err;
            if (state.code == EndMark) state = handleEndMark();
        }
        *written = sum;
        return state;
    }
    Error Close() override
    {
        if (!optsLive) return state;
        ready.clear(); dstBlk.clear(); dstOff = 0;
        if (!state) state = E(ErrClosed);
        optsLive = false;
        return Error();
    }
};
}  // namespace
std::unique_ptr<Reader> NewReader(Source& rd, BlockEngine& eng, const Options& o) { return std::unique_ptr<Reader>(new ReaderImpl(rd, eng, o)); }

// ------------------------------------------------------------------------------------------------ block API (plz4_block.go:78-172)
int CompressBlockBound(int sz) { return plz4hip_compress_bound(sz); }

Error CompressBlock(BlockEngine& eng, const uint8_t* src, size_t n, int level, std::vector<uint8_t>* dst, bool dstProvided,
                    const std::vector<uint8_t>* dict)
{
    if (level < 1) level = 1; if (level > 12) level = 12;
    if (!dstProvided) dst->assign((size_t)CompressBlockBound((int)n), 0);
    const void* s = src; int32_t sl = (int32_t)n, cap = (int32_t)dst->size(), res = 0; void* d = dst->data();
    int rc;
    if (dict) {
        void* dh = eng.DictCreate(dict->data(), (int)dict->size());
        if (!dh) return E(ErrUnsupported);
        rc = eng.CompressBatchDict(1, &s, &sl, &d, &cap, level, dh, &res);
        eng.DictDestroy(dh);
    } else rc = eng.CompressBatch(1, &s, &sl, &d, &cap, level, &res);
    if (rc != 0) return E(rc == PLZ4HIP_E_UNSUPPORTED ? ErrUnsupported : ErrEngine);
    if (res == 0) return E(ErrCompress);
    dst->resize((size_t)res);
    return Error();
}

Error DecompressBlock(BlockEngine& eng, const uint8_t* src, size_t n, std::vector<uint8_t>* dst, bool dstProvided,
                      const std::vector<uint8_t>* dict)
{
    const void* s = src; int32_t sl = (int32_t)n;
    void* dh = nullptr;
    if (dict) { dh = eng.DictCreate(dict->data(), (int)dict->size()); if (!dh) return E(ErrUnsupported); }
    struct Guard { BlockEngine& e; void* h; ~Guard() { if (h) e.DictDestroy(h); } } guard{eng, dh};
    size_t bufSize = dstProvided ? dst->size() : n * 4;
    for (int tries = 1;; ++tries) {
        if (!dstProvided) dst->assign(bufSize, 0);
        int32_t cap = (int32_t)dst->size(), res = 0; void* d = dst->data();
        const int rc = dh ? eng.DecompressBatchDict(1, &s, &sl, &d, &cap, dh, &res) : eng.DecompressBatch(1, &s, &sl, &d, &cap, &res);
        if (rc != 0) return E(rc == PLZ4HIP_E_UNSUPPORTED ? ErrUnsupported : ErrEngine);
        if (res >= 0) { dst->resize((size_t)res); return Error(); }
        if (dstProvided || tries >= 3) return E(ErrDecompress, true);                      // maxTries, plz4_block.go:8-11
        bufSize *= 2;
    }
}

}  // namespace plz4h
