// host_capi.cpp -- flat C entry points over the C++ host layer (plz4_host.hpp), for language bindings and for the
// Python test-suite (ctypes).  Memory sinks/sources with optional fault injection stand in for io.Writer/io.Reader
// (the reference's failWriter / failReader test fakes, internal/test/wr_test.go:1238-1250, rd_test.go:1493-1505).
#include <cstring>
#include <vector>

#include "plz4_host.hpp"

using namespace plz4h;

extern "C" {

typedef struct plz4h_opts {
    int32_t  nparallel, pending_sz, level, has_content_sz;
    uint64_t content_sz;
    int64_t  read_offset;
    int32_t  block_checksum, block_linked, content_checksum, skip_content_sz, has_dict_id;
    uint32_t dict_id;
    int32_t  block_size_idx, gpu_batch;
    int32_t  fail_after_writes;      // sink: the N-th Write (0-based) and all later ones fail; -1 = never
    int64_t  fail_read_at;           // source: reads fail once this many bytes were delivered; -1 = never
    int32_t  has_dictionary;
    const uint8_t* dictionary;       // WithDictionary bytes (has_dictionary != 0)
    int64_t  dictionary_len;
} plz4h_opts;

}  // extern "C"

namespace {

int enc(const Error& e) { return e.code | (e.corrupted ? 0x10000 : 0); }

Options to_options(const plz4h_opts* o, std::vector<int64_t>* prog)
{
    Options x;
    x.NParallel = o->nparallel; x.PendingSz = o->pending_sz; x.Level = o->level;
    x.HasContentSz = o->has_content_sz != 0; x.ContentSz = o->content_sz; x.ReadOffset = o->read_offset;
    x.BlockChecksum = o->block_checksum != 0; x.BlockLinked = o->block_linked != 0;
    x.ContentChecksum = o->content_checksum != 0; x.SkipContentSz = o->skip_content_sz != 0;
    x.HasDictionaryId = o->has_dict_id != 0; x.DictionaryId = o->dict_id;
    x.BlockSizeIdx = (o->block_size_idx >= 4 && o->block_size_idx <= 7) ? o->block_size_idx : BlockIdx4MB;   // WithBlockSize
    x.GpuBatchBlocks = o->gpu_batch; x.HasDictionary = o->has_dictionary != 0;
    if (x.HasDictionary && o->dictionary && o->dictionary_len > 0) x.Dictionary.assign(o->dictionary, o->dictionary + o->dictionary_len);
    if (x.Level < 1) x.Level = 1; if (x.Level > 12) x.Level = 12;                                               // WithLevel
    if (prog) x.Handler = [prog](int64_t a, int64_t b) { prog->push_back(a); prog->push_back(b); };
    return x;
}

struct MemSink : Sink {
    std::vector<uint8_t> buf; int failAfter = -1; int writes = 0;
    Error write(const uint8_t* p, size_t n, size_t* w) override
    {
        Error e;
        if (failAfter >= 0 && writes >= failAfter) { *w = 0; e.code = ErrIO; ++writes; return e; }
        ++writes; buf.insert(buf.end(), p, p + n); *w = n; return e;
    }
};
struct MemSource : Source {
    const uint8_t* p = nullptr; size_t n = 0, off = 0; int64_t failAt = -1; size_t chunk = 0;
    Error read(uint8_t* d, size_t want, size_t* got) override
    {
        Error e; *got = 0;
        if (failAt >= 0 && (int64_t)off >= failAt) { e.code = ErrIO; return e; }
        size_t k = std::min(want, n - off);
        if (chunk) k = std::min(k, chunk);
        if (failAt >= 0) k = std::min<size_t>(k, (size_t)(failAt - (int64_t)off));
        if (k == 0 && off >= n) { e.code = ErrEOF; return e; }
        memcpy(d, p + off, k); off += k; *got = k; return e;
    }
    bool skip(int64_t k) override { if (failAt >= 0) return false; if ((int64_t)(n - off) < k) return false; off += (size_t)k; return true; }
};

struct WriterH { MemSink sink; std::vector<int64_t> prog; std::unique_ptr<Writer> w; };
struct ReaderH { MemSource src; MemSink out; std::vector<int64_t> prog; std::vector<uint8_t> data; std::unique_ptr<Reader> r; };

}  // namespace

extern "C" {

void* plz4h_engine_hip(int device, int* rc) { return NewHipEngine(device, rc).release(); }
void* plz4h_engine_vtable(const EngineVTable* vt) { return NewVTableEngine(*vt).release(); }
void  plz4h_engine_free(void* e) { delete (BlockEngine*)e; }
const char* plz4h_error_string(int code) { return ErrorString(code & 0xFFFF); }

void* plz4h_writer_new(void* engine, const plz4h_opts* o)
{
    WriterH* h = new WriterH();
    h->sink.failAfter = o->fail_after_writes;
    h->w = NewWriter(h->sink, *(BlockEngine*)engine, to_options(o, &h->prog));
    return h;
}
int plz4h_writer_write(void* w, const uint8_t* p, size_t n, size_t* consumed) { return enc(((WriterH*)w)->w->Write(p, n, consumed)); }
int plz4h_writer_flush(void* w) { return enc(((WriterH*)w)->w->Flush()); }
int plz4h_writer_close(void* w) { return enc(((WriterH*)w)->w->Close()); }
int plz4h_writer_read_from(void* w, const uint8_t* p, size_t n, size_t chunk, int64_t fail_read_at, int64_t* consumed)
{
    MemSource s; s.p = p; s.n = n; s.chunk = chunk; s.failAt = fail_read_at;
    return enc(((WriterH*)w)->w->ReadFrom(s, consumed));
}
size_t plz4h_writer_output(void* w, const uint8_t** p) { WriterH* h = (WriterH*)w; *p = h->sink.buf.data(); return h->sink.buf.size(); }
size_t plz4h_writer_progress(void* w, const int64_t** p) { WriterH* h = (WriterH*)w; *p = h->prog.data(); return h->prog.size(); }
void   plz4h_writer_free(void* w) { delete (WriterH*)w; }

void* plz4h_reader_new(void* engine, const plz4h_opts* o, const uint8_t* data, size_t n)
{
    ReaderH* h = new ReaderH();
    h->data.assign(data, data + n);
    h->src.p = h->data.data(); h->src.n = n; h->src.failAt = o->fail_read_at;
    h->out.failAfter = o->fail_after_writes;
    h->r = NewReader(h->src, *(BlockEngine*)engine, to_options(o, &h->prog));
    return h;
}
int plz4h_reader_read(void* r, uint8_t* dst, size_t cap, size_t* got) { return enc(((ReaderH*)r)->r->Read(dst, cap, got)); }
int plz4h_reader_write_to(void* r, int64_t* written) { ReaderH* h = (ReaderH*)r; return enc(h->r->WriteTo(h->out, written)); }
size_t plz4h_reader_output(void* r, const uint8_t** p) { ReaderH* h = (ReaderH*)r; *p = h->out.buf.data(); return h->out.buf.size(); }
size_t plz4h_reader_progress(void* r, const int64_t** p) { ReaderH* h = (ReaderH*)r; *p = h->prog.data(); return h->prog.size(); }
int    plz4h_reader_close(void* r) { return enc(((ReaderH*)r)->r->Close()); }
void   plz4h_reader_free(void* r) { delete (ReaderH*)r; }

int plz4h_compress_block_bound(int n) { return CompressBlockBound(n); }
int plz4h_compress_block(void* engine, const uint8_t* src, size_t n, int level, uint8_t* dst, size_t cap, int dst_provided, size_t* out_len,
                         const uint8_t* dict, int64_t dict_len)
{
    std::vector<uint8_t> d; if (dst_provided) d.assign(cap, 0);
    std::vector<uint8_t> dv; if (dict_len >= 0) dv.assign(dict, dict + dict_len);
    Error e = CompressBlock(*(BlockEngine*)engine, src, n, level, &d, dst_provided != 0, dict_len >= 0 ? &dv : nullptr);
    if (!e) { if (d.size() > cap) return enc(Error{ErrCompress, false}); memcpy(dst, d.data(), d.size()); *out_len = d.size(); }
    return enc(e);
}
int plz4h_decompress_block(void* engine, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int dst_provided, size_t* out_len,
                           const uint8_t* dict, int64_t dict_len)
{
    std::vector<uint8_t> d; if (dst_provided) d.assign(cap, 0);
    std::vector<uint8_t> dv; if (dict_len >= 0) dv.assign(dict, dict + dict_len);
    Error e = DecompressBlock(*(BlockEngine*)engine, src, n, &d, dst_provided != 0, dict_len >= 0 ? &dv : nullptr);
    if (!e) { if (d.size() > cap) return enc(Error{ErrDecompress, true}); memcpy(dst, d.data(), d.size()); *out_len = d.size(); }
    return enc(e);
}
int plz4h_write_header(const plz4h_opts* o, uint8_t out[19]) { return WriteHeaderBytes(to_options(o, nullptr), out); }
uint32_t plz4h_xxh32(const uint8_t* p, size_t n) { return Xxh32(p, n); }

}  // extern "C"
