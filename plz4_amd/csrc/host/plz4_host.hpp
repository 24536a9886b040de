// plz4_host.hpp -- host side of the drop-in: a C++ mirror of plz4's Go API for the per-block path, sitting on the
// batched C ABI (include/plz4hip.h).  The reference's host code is Go; there is no Go toolchain in the build image,
// so the layers the Go shim would keep (framing, option handling, in-order emission, error latching) are
// re-stated here in C++ with the reference's names, argument meaning and error behaviour, and tested the way the
// reference tests them (tests/test_host_writer.py, tests/test_host_reader.py).  The cgo shim a plz4 maintainer
// would add is in go/ and INTEGRATION.md.
//
//   NewWriter / Writer{Write,Flush,Close,ReadFrom}   plz4_writer.go:40-53, internal/pkg/sync/writer.go, async/writer.go
//   NewReader / Reader{Read,WriteTo,Close}           plz4_reader.go:28-33, internal/pkg/rdr/rdr.go
//   CompressBlock / DecompressBlock / CompressBlockBound   plz4_block.go:78-172
//   Options (With*)                                  plz4_opts.go:70-255, internal/pkg/opts/opts.go:24-95
//   errors                                           plz4_err.go:11-45, internal/pkg/zerr/zerr.go:11-41
//
// What replaces what: the goroutine pipeline (reader -> N compress workers -> in-order writer, async/writer.go:232-381)
// becomes "collect up to nPending blocks -> one plz4hip batch call -> emit in order"; the cgo Compressor is the
// BlockEngine below; the block/content xxHash32 path moves to the device (block) and stays a host stream (content).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace plz4h {

// ---- errors (zerr.go:11-41).  `corrupted` is the errors.Is(err, ErrCorrupted) join (zerr.WrapCorrupted).
enum Code : int {
    OK = 0, EndMark, ErrClosed, ErrCorrupted, ErrHeaderHash, ErrBlockHash, ErrContentHash, ErrHeaderRead, ErrHeaderWrite,
    ErrMagic, ErrVersion, ErrDescriptorRead, ErrBlockSizeRead, ErrBlockRead, ErrBlockSizeOverflow, ErrCompress,
    ErrDecompress, ErrReserveBitSet, ErrBlockDescriptor, ErrContentHashRead, ErrContentSize, ErrReadOffset,
    ErrReadOffsetLinked, ErrSkip, ErrNibble, ErrUnsupported,
    ErrIO,        // the caller's io.Writer / io.Reader failed
    ErrEOF,       // io.EOF
    ErrEngine     // device/runtime failure of the block engine: never turned into a stored block
};
struct Error {
    int  code = OK;
    bool corrupted = false;
    bool ok() const { return code == OK; }
    explicit operator bool() const { return code != OK; }
};
inline bool Lz4Corrupted(const Error& e) { return e.corrupted; }      // plz4_err.go:43-45
const char* ErrorString(int code);

// ---- io
struct Sink   { virtual ~Sink() {}   virtual Error write(const uint8_t* p, size_t n, size_t* written) = 0; };          // io.Writer
struct Source { virtual ~Source() {} virtual Error read(uint8_t* p, size_t n, size_t* got) = 0;                       // io.Reader (got==0 && OK => EOF)
                virtual bool  skip(int64_t /*n*/) { return false; } };                                                 // io.Seeker fast path

// ---- options == opts.OptsT (opts/opts.go:24-41); defaults == parseOpts (plz4_opts.go:238-255)
enum BlockIdx : int { BlockIdx64KB = 4, BlockIdx256KB = 5, BlockIdx1MB = 6, BlockIdx4MB = 7 };
int BlockIdxSize(int idx);                                         // descriptor/index.go:22-33

struct Options {
    int      NParallel       = 1;      // 0 sync; >=1 async; <0 auto.  Here: async => batches of CalcPending() blocks
    int      PendingSz       = 0;
    int      Level           = 1;
    bool     HasContentSz    = false;  uint64_t ContentSz = 0;
    int64_t  ReadOffset      = 0;
    bool     BlockChecksum   = false;
    bool     BlockLinked     = false;
    bool     ContentChecksum = true;
    bool     SkipContentSz   = false;
    std::vector<uint8_t> Dictionary;   bool HasDictionary = false;
    bool     HasDictionaryId = false;  uint32_t DictionaryId = 0;
    int      BlockSizeIdx    = BlockIdx4MB;
    std::function<void(int64_t, int64_t)> Handler;                  // WithProgress
    int      GpuBatchBlocks  = 0;      // engine batch depth; 0 => max(CalcPending(), 64).  Not in the reference.
    int CalcPending() const;                                        // opts/opts.go:62-95
};

// ---- the codec boundary == compress.Compressor / compress.Decompressor (compress/compress.go:7-13, decompress.go:14-16),
// in the batch form of include/plz4hip.h.  Return value: 0 or a PLZ4HIP_E_* engine failure.
struct BlockEngine {
    virtual ~BlockEngine() {}
    virtual int CompressBatch(int n, const void* const* src, const int32_t* srcLen, void* const* dst, const int32_t* dstCap,
                              int level, int32_t* result) = 0;
    virtual int DecompressBatch(int n, const void* const* src, const int32_t* srcLen, void* const* dst, const int32_t* dstCap,
                                int32_t* result) = 0;
    virtual int EncodeRecords(int n, const void* const* src, const int32_t* srcLen, int bsz, int level, int blockChecksum,
                              void* const* rec, int32_t* recLen) = 0;
    virtual int DecodeRecords(int n, const void* const* rec, const int32_t* recLen, int bsz, int blockChecksum,
                              void* const* dst, int32_t* result, int32_t* status) = 0;
    // dictionaries / linked blocks (compress.NewCompressorFactory(level, independent, dict), compress.go:32-48).
    // Default: not supported by this engine.
    virtual void* DictCreate(const uint8_t* /*dict*/, int /*len*/) { return nullptr; }
    virtual void  DictDestroy(void* /*d*/) {}
    virtual int CompressBatchDict(int, const void* const*, const int32_t*, void* const*, const int32_t*, int, void*, int32_t*) { return -4; }
    virtual int DecompressBatchDict(int, const void* const*, const int32_t*, void* const*, const int32_t*, void*, int32_t*) { return -4; }
    virtual int EncodeRecordsEx(int, const void* const*, const int32_t*, int, int, int, int /*linked*/, void* /*dict*/,
                                const void* /*prevTail*/, int /*prevTailLen*/, void* const*, int32_t*) { return -4; }
    virtual int DecodeRecordsEx(int, const void* const*, const int32_t*, int, int, int /*linked*/, void* /*dict*/,
                                void* /*window*/, int* /*windowLen*/, void* const*, int32_t*, int32_t*) { return -4; }
    // streaming content checksum kept by the engine (xxh32.XXHZero, async/hash.go:99-111): while a stream is set with
    // HashAttach, EncodeRecords[Ex] also writes the plaintext of its blocks, in order, into it, and DecodeRecords[Ex] the plaintext
    // it produces.  Default: the engine has none
    // (HashNew returns null) and the writer hashes on the host.
    virtual void* HashNew() { return nullptr; }
    virtual void  HashFree(void* /*h*/) {}
    virtual int   HashAttach(void* /*h or null*/) { return -4; }
    virtual int   HashSum(void* /*h*/, uint32_t* /*out*/) { return -4; }
    virtual int   HashReset(void* /*h*/) { return -4; }
};
std::unique_ptr<BlockEngine> NewHipEngine(int device, int* rc);     // the product engine (plz4hip_ctx)

// C plug-in point for other engines (tests inject an oracle-backed one; see tests/hostlib).
struct EngineVTable {
    void* user;
    int (*compress_batch)(void*, int, const void* const*, const int32_t*, void* const*, const int32_t*, int, int32_t*);
    int (*decompress_batch)(void*, int, const void* const*, const int32_t*, void* const*, const int32_t*, int32_t*);
    int (*encode_records)(void*, int, const void* const*, const int32_t*, int, int, int, void* const*, int32_t*);
    int (*decode_records)(void*, int, const void* const*, const int32_t*, int, int, void* const*, int32_t*, int32_t*);
    // optional (may be null): dictionaries / linked blocks
    void* (*dict_create)(void*, const uint8_t*, int);
    void  (*dict_destroy)(void*, void*);
    int (*compress_batch_dict)(void*, int, const void* const*, const int32_t*, void* const*, const int32_t*, int, void*, int32_t*);
    int (*decompress_batch_dict)(void*, int, const void* const*, const int32_t*, void* const*, const int32_t*, void*, int32_t*);
    int (*encode_records_ex)(void*, int, const void* const*, const int32_t*, int, int, int, int, void*, const void*, int, void* const*, int32_t*);
    int (*decode_records_ex)(void*, int, const void* const*, const int32_t*, int, int, int, void*, void*, int*, void* const*, int32_t*, int32_t*);
};
std::unique_ptr<BlockEngine> NewVTableEngine(const EngineVTable& vt);

// ---- xxHash32 seed 0, host side (header checksum + streaming content checksum; xxh32zero.go)
uint32_t Xxh32(const uint8_t* p, size_t n);
struct Xxh32Stream { uint32_t acc[4]; uint64_t total = 0; uint8_t buf[16]; uint32_t fill = 0;
                     void Reset(); void Write(const uint8_t* p, size_t n); uint32_t Sum32() const; };

// ---- frame header / trailer (header/write.go:23-73, header/read.go:26-119, trailer/trailer.go)
int  WriteHeaderBytes(const Options& o, uint8_t out[19]);
struct HeaderT { int64_t Sz = 0; uint32_t DictId = 0; uint64_t ContentSz = 0; uint8_t Flags = 0; uint8_t BlockDesc = 0; };

// ---- Writer
class Writer {
public:
    virtual ~Writer() {}
    virtual Error Write(const uint8_t* p, size_t n, size_t* consumed) = 0;
    virtual Error Flush() = 0;
    virtual Error Close() = 0;
    virtual Error ReadFrom(Source& r, int64_t* consumed) = 0;
};
std::unique_ptr<Writer> NewWriter(Sink& wr, BlockEngine& eng, const Options& o);      // plz4_writer.go:40-53

// ---- Reader
class Reader {
public:
    virtual ~Reader() {}
    virtual Error Read(uint8_t* dst, size_t n, size_t* got) = 0;      // rdr.go:39-87
    virtual Error WriteTo(Sink& w, int64_t* written) = 0;             // rdr.go:139-174
    virtual Error Close() = 0;                                        // rdr.go:111-137
};
std::unique_ptr<Reader> NewReader(Source& rd, BlockEngine& eng, const Options& o);    // plz4_reader.go:28-33

// ---- raw block API (plz4_block.go:78-172)
int   CompressBlockBound(int sz);
Error CompressBlock(BlockEngine& eng, const uint8_t* src, size_t n, int level, std::vector<uint8_t>* dst, bool dstProvided,
                    const std::vector<uint8_t>* dict = nullptr);                                   // WithBlockDictionary
Error DecompressBlock(BlockEngine& eng, const uint8_t* src, size_t n, std::vector<uint8_t>* dst, bool dstProvided,
                      const std::vector<uint8_t>* dict = nullptr);

}  // namespace plz4h
