// plz4hip.hip -- gfx950 kernels + the C ABI declared in include/plz4hip.h.
//
// Launch shape: ONE 64-lane wavefront per LZ4 block, persistent waves pulling block indices from a device-side
// counter (every wave exits when the counter passes nBlocks).  An encoder wave owns 16 KiB of LDS (liblz4's hash
// table, lz4.h:695-697): ten encoder waves per CU when a call fills the chip (one workgroup of ten independent waves
// per CU, see ENC_WAVE_TABLE), one-wave workgroups otherwise; the record decoder stages each batch of output in
// 1.1 KiB of LDS and is bounded by its registers (24 waves per CU).  Waves never communicate.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>

#include "../../include/plz4hip.h"
#if defined(PLZ4_STATS)
__device__ unsigned long long plz4_stats[24];
#endif
#include "lz4_device.inl"
#include "lz4_seq_device.inl"
#include "lz4hc_device.inl"
#include "lz4hc12_device.inl"
#include "lz4hc_lazy_device.inl"
#include "lz4_dx_device.inl"

namespace {

using namespace plz4;

// ------------------------------------------------------------------------------------------------ kernels
struct CodecArgs {
    const uint8_t* src;  int64_t srcStride;  const int32_t* srcLen;  int64_t srcBytes;  int bsz;
    uint8_t*       dst;  int64_t dstStride;  const int32_t* dstCap;  int dstCapAll;
    const int64_t* recOff;
    int32_t*       result;
    int32_t*       status;
    uint32_t*      queue;
    int            nBlocks;
    int            blockChecksum;
    // dictionary / linked blocks
    const uint8_t*  dict;       int dictLen;        // last <= 64 KiB of the user dictionary (device), or null
    const uint32_t* dictTable;                      // LZ4_loadDictSlow table of that dictionary
    int             linked;                         // encode: block i>0 is primed with the tail of block i-1
    const uint8_t*  prevTail;   int prevTailLen;    // linked: window of block 0 (-1: block 0 starts a frame)
    uint8_t*        window;     int* windowLen;     // linked decode: the 64 KiB sliding dictionary (2 x 64 KiB ping-pong) per chain, in/out
    int             nChains;    const int32_t* chainFirst;   // linked decode of several frames at once: chain c = records [chainFirst[c], chainFirst[c+1])
    // HC levels 2..12
    int             level;
    uint8_t*        hcWork;                         // gridDim.x x kHcWorkBytes
    const uint32_t* hcDictHash;                     // HC + dictionary: the dictionary context's tables for this level's strategy
    const uint16_t* hcDictChain;                    //   (clz4.NewDictCtxHC, clz4.go:122-147), built by k_hc_dict_prime
    int             hcEx;                           // HC call with a dictionary and/or linked blocks: inputs have 64 KiB of scratch in front
    int32_t*        hcPfx;                          // such a call on the list path (k_hc_ext_prep): per block, the bytes of the external segment that
                                                    // lies right in front of it (>= 0), or -1: a block <= 4 KiB under a dictionary context (k_encode_*_hc's)
    // level 12 in three phases (lz4hc12_device.inl): per-block chain and search results of the group [blk0, blk0 + nBlocks)
    int             blk0;
    uint16_t*       h12Chain;   int64_t h12ChainStride;      // entries per block (a multiple of 1024)
    uint32_t*       h12Rank;                                 // h12ChainStride entries per block
    uint32_t*       h12List;                                 // h12ChainStride + 8 entries per block (8 readable entries in front of index 0)
    uint32_t*       h12Offsets;                              // 32768 per block: where each hash's run starts in the list
    Hc12F*          h12F;       int64_t h12FStride;          // entries per block
    uint8_t*        h12Ws;                                   // gridDim.x x kHc12WsGlobalBytes: the price table's overflow
    int32_t*        h12Err;                                  // set when a kernel gives up (spin guard)
    int             rawMode;                                 // parse kernel: 1 = raw LZ4 blocks (dstCap per block), 0 = frame records
    int             h12Gather, h12Idle;                      // search kernel: lanes a phase / the queue waits for before its section runs
    // level 1 in stages (lz4_seq_device.inl): parse -> sizes -> scan -> write -> finish over the group [blk0, blk0 + nBlocks);
    // the workspace is indexed by the block's number inside the group
    SeqInfo*        l1Info;                                  // per block
    uint64_t*       l1Seq;      int64_t l1SeqStride;         // sequence records, entries per block
    uint32_t*       l1ChunkBytes; uint32_t* l1ChunkOff; int l1MaxChunks;   // per block: bytes of / before every chunk of 1024 sequences
    uint8_t*        l1Bk;                                    // per sequence: its catch-up length (l1SeqStride bytes per block)
    int             l1MaxLen;                                // what the workspace was sized for
    // levels 3..9 (lz4hc_lazy_device.inl): a block is walked in up to lzSegs segments of at least lzMinSeg bytes at once
    int             lzSegs, lzMinSeg;
    uint64_t*       lzRec;      uint64_t* lzBridge;  int64_t lzRecStride;     // records of the segments / of the walks into them, entries per block
    LzSegMeta*      lzMeta;     uint64_t* lzStarts;  LzPiece* lzPieces;       // per block: lzSegs, lzSegs * kLzStarts, 2 * lzSegs entries
    // a few blocks decoded by the whole chip (lz4_dx_device.inl): per block, the chain table of its input (one entry per byte), the
    // pointer table of its output (one per byte), the units of its segments, its state
    uint64_t*       dxT;        int64_t dxTStride;
    uint32_t*       dxPtr;      int64_t dxPtrStride;
    DxUnit*         dxUnits;    int dxMaxSeg;   int dxRound;
    DxInfo*         dxInfo;
    // the staged level-1 call writing its records back to back (plz4hip_dev_encode_body): dst = the frame body, bodyOff[i] = where
    // record i starts (made by the call: the scan over the records' lengths), bodyCap = the body's room
    int64_t*        bodyOff;    int64_t bodyCap;
    // the staged level-1 parse tells when its block queue has run dry (its last blocks are under way, workgroups begin to leave):
    // gate <- max(gate, gateSeq); the parse of the next call on ANOTHER stream waits for that behind k_parse_gate (launch_l1)
    uint32_t*       gate;       uint32_t gateSeq;
    const int64_t*  dxSrcOff;   const int32_t* dxLen;                       // records: where a block's payload starts in src and its size (-1: not this path's)
    int32_t*        dxHashBad;                                              // records: the payload's xxh32 does not match (k_dx_rec_hash)
};

__device__ __forceinline__ int next_block(uint32_t* q)
{
    uint32_t v = 0;
    if ((threadIdx.x & 63u) == 0) v = atomicAdd(q, 1u);
    return (int)plz4_readfirstlane(v);
}

__device__ __forceinline__ int block_len(const CodecArgs& a, int i)
{
    if (a.srcLen) return a.srcLen[i];
    const int64_t rem = a.srcBytes - (int64_t)i * a.bsz;
    return (int)(rem < a.bsz ? rem : a.bsz);
}

// The level-1 encoder kernels run up to TEN independent waves per workgroup, each with its own 16 KiB table: one such
// workgroup owns all 160 KiB of a CU's LDS, i.e. ten tables per CU, where one-wave workgroups of 16 KiB only reach nine per CU
// (scripts/micro/residency.hip: "2560 16384 64" -> 9 per CU; two 80 KiB workgroups of five waves do not co-reside either --
// measured on k_encode_rec: the second one waits for the first).  The waves of a workgroup never synchronise; each pulls block
// ids from the queue until it is empty.  Only a call that fills the chip by itself is launched that way: a smaller one (the
// chunks of a host call, several of which share the GPU) keeps one-wave workgroups of 16 KiB, which leave the rest of a CU's
// LDS to other kernels -- hence the two instantiations of every encoder kernel (W = waves per workgroup).
constexpr int kEncWavesPerWg = 10;
#define ENC_WAVE_TABLE(name) \
    __shared__ uint32_t name##_all[W][kHashBytes / 4]; \
    uint32_t* const name = name##_all[W == 1 ? 0 : plz4_readfirstlane((int)(threadIdx.x >> 6))]

// LZ4 blocks only: dst[i] <- LZ4_compress_fast(src[i]), result[i] = bytes or 0.
template <int W> __global__ __launch_bounds__(64 * W) void k_encode_raw(CodecArgs a)
{
    ENC_WAVE_TABLE(lds);
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int n   = block_len(a, i);
        const int cap = a.dstCap ? a.dstCap[i] : a.dstCapAll;
        const int r   = wave_encode_block(a.src + (int64_t)i * a.srcStride, n, a.dst + (int64_t)i * a.dstStride, cap, lds);
        if ((threadIdx.x & 63u) == 0) a.result[i] = r;
    }
}

// blk.CompressToBlk on the device: [LE32 size|stored][payload][LE32 xxh32?] at dst + i*dstStride.
template <int W> __global__ __launch_bounds__(64 * W) void k_encode_rec(CodecArgs a)
{
    ENC_WAVE_TABLE(lds);
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int      n   = block_len(a, i);
        const uint8_t* s   = a.src + (int64_t)i * a.srcStride;
        uint8_t*       rec = a.dst + (int64_t)i * a.dstStride;
        int      c    = wave_encode_block(s, n, rec + 4, a.bsz, lds);       // capacity == bsz (blk/blk.go:73)
        uint32_t word = (uint32_t)c & 0x7FFFFFFFu;
        if (c == 0) {                                                        // ErrCompress -> stored raw (blk.go:78-92)
            wave_copy(rec + 4, s, n);
            c = n;
            word = 0x80000000u | ((uint32_t)n & 0x7FFFFFFFu);
        }
        int len = c + 4;
        if (a.blockChecksum) {                                               // over the payload as stored (blk.go:98-102)
            WAVE_FENCE();
            const uint32_t x = wave_xxh32(rec + 4, c);
            if ((threadIdx.x & 63u) == 0) st32u(rec + 4 + c, x);
            len += 4;
        }
        if ((threadIdx.x & 63u) == 0) { st32u(rec, word); a.result[i] = len; }
    }
}

// ---- level 1 in stages (independent blocks up to 4 MiB, no dictionary): lz4_seq_device.inl ------------------------------------
// k_l1_parse: persistent, one wave per block, the hash table in LDS -- the only serial stage; it writes one 8-byte record per
// sequence.  The others are plain data-parallel kernels, one lane per sequence, any number of waves per block, no LDS.
template <int kLdsWin = 0>
__device__ __forceinline__ void l1_parse_loop(const CodecArgs& a, uint32_t* lds, uint8_t* scr = nullptr)
{
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int gi = a.blk0 + i;
        const int n  = block_len(a, gi);
        int lastAnchor = 0, nseq = -1;                                       // -1: a block the workspace was not sized for
        if (n >= 0 && n <= a.l1MaxLen)
            nseq = wave_parse_l1<kLdsWin>(a.src + (int64_t)gi * a.srcStride, n, lds, a.l1Seq + (int64_t)i * a.l1SeqStride, &lastAnchor, scr);
        if ((threadIdx.x & 63u) == 0) { SeqInfo inf; inf.nseq = nseq; inf.lastAnchor = lastAnchor; inf.total = 0; inf.stored = 0; a.l1Info[i] = inf; }
    }
    if (a.gate && (threadIdx.x & 63u) == 0) atomicMax(a.gate, a.gateSeq);    // (sequence numbers: launch_l1 starts over before they wrap)
}
// One wave that holds its stream back until the parse launch `want` has run its queue dry (or ~1.5 s have passed: never a hang).
// Two parse launches started together share the CUs half and half and stay in step from then on -- nothing of one call then runs
// beside the other's parse; behind this gate the second one moves into the CUs as the first one's workgroups leave them.
__global__ __launch_bounds__(64) void k_parse_gate(const uint32_t* gate, uint32_t want)
{
    for (int it = 0; it < 400000; ++it) {
        if ((int32_t)(__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) break;
        __builtin_amdgcn_s_sleep(127);
    }
}
template <int W> __global__ __launch_bounds__(64 * W) void k_l1_parse(CodecArgs a)
{
    ENC_WAVE_TABLE(lds);
    l1_parse_loop<2>(a, lds);                  // (ten tables fill the CU's LDS: the windows through the lane exchange, lz4_seq_device.inl)
}

// grid (waves per block / 4, blocks of the group): bytes of every chunk of 1024 sequences.  kBack: the level-1 parser's records
// (their catch-up is measured here); false: the HC parsers' records, which carry the final match.
template <bool kBack> __global__ __launch_bounds__(256) void k_l1_sizes(CodecArgs a)
{
    const int i = blockIdx.y, wave = blockIdx.x * 4 + (int)(threadIdx.x >> 6), nW = gridDim.x * 4;
    const int nseq = a.l1Info[i].nseq;
    if (nseq <= 0) return;
    const int gi = a.blk0 + i;
    const uint8_t*  s   = a.src + (int64_t)gi * a.srcStride;
    const uint64_t* seq = a.l1Seq + (int64_t)i * a.l1SeqStride;
    const int nChunks = (nseq + kSeqChunk - 1) / kSeqChunk;
    for (int c = wave; c < nChunks; c += nW) {
        const uint32_t v = seq_emit_sizes<kBack>(s, seq, kBack ? a.l1Bk + (int64_t)i * a.l1SeqStride : nullptr, nseq, c);
        if ((threadIdx.x & 63u) == 0) a.l1ChunkBytes[(int64_t)i * a.l1MaxChunks + c] = v;
    }
}

// one wave per block: where every chunk goes, the block's size, liblz4's limitedOutput verdict; raw mode: result[i] = size or 0;
// records (blk.CompressToBlk, blk/blk.go:69-109): the size word (stored iff the encoder returned 0) and, without block
// checksums, the record length
__global__ __launch_bounds__(256) void k_l1_scan(CodecArgs a)
{
    const int i = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= a.nBlocks) return;
    const int gi = a.blk0 + i;
    const int n  = block_len(a, gi);
    SeqInfo inf = a.l1Info[i];
    const int cap = a.rawMode ? (a.dstCap ? a.dstCap[gi] : a.dstCapAll) : a.bsz;     // records: capacity == bsz (blk/blk.go:73)
    int total = 0;
    const bool broken = inf.nseq == kSeqEngineFailed;                                 // (an HC search phase that gave up: no result)
    if (inf.nseq >= 0)
        total = seq_emit_scan(a.l1ChunkBytes + (int64_t)i * a.l1MaxChunks, a.l1ChunkOff + (int64_t)i * a.l1MaxChunks, inf.nseq, inf.lastAnchor, n, cap);
    if ((threadIdx.x & 63u) == 0) {
        inf.total = total; inf.stored = (total == 0);
        a.l1Info[i] = inf;
        if (broken) a.result[gi] = PLZ4HIP_E_DEVICE;
        else if (a.rawMode) a.result[gi] = total;
        else if (a.bodyOff) {
            // records back to back: every record's length now (the scan behind this kernel places them; the size word is written
            // with the payload, k_l1_write)
            a.result[gi] = (total ? total : n) + 4 + (a.blockChecksum ? 4 : 0);
        } else {
            const int c = total ? total : n;                                          // ErrCompress -> stored raw (blk.go:78-92)
            st32u(a.dst + (int64_t)gi * a.dstStride, total ? ((uint32_t)c & 0x7FFFFFFFu) : (0x80000000u | ((uint32_t)n & 0x7FFFFFFFu)));
            if (!a.blockChecksum) a.result[gi] = c + 4;
        }
    }
}

// grid as k_l1_sizes: every chunk written at its place; a record whose encoder returned 0 gets the plaintext instead
template <bool kBack> __global__ __launch_bounds__(256) void k_l1_write(CodecArgs a)
{
    const int i = blockIdx.y, wave = blockIdx.x * 4 + (int)(threadIdx.x >> 6), nW = gridDim.x * 4;
    const SeqInfo inf = a.l1Info[i];
    const int gi = a.blk0 + i;
    const int n  = block_len(a, gi);
    const uint8_t* s   = a.src + (int64_t)gi * a.srcStride;
    uint8_t*       out = a.dst + (int64_t)gi * a.dstStride + (a.rawMode ? 0 : 4);
    if (a.bodyOff) {
        // the record's place in the frame body; one that does not fit is not written (the caller sees bodyOff[n] > bodyCap)
        const int64_t off = a.bodyOff[gi];
        if (inf.nseq == kSeqEngineFailed || off + (int64_t)((inf.total ? inf.total : n) + 4 + (a.blockChecksum ? 4 : 0)) > a.bodyCap) return;
        out = a.dst + off + 4;
        if (wave == 0 && (threadIdx.x & 63u) == 0)                                    // the size word, stored iff the encoder returned 0 (blk.go:78-92)
            st32u(out - 4, inf.total ? ((uint32_t)inf.total & 0x7FFFFFFFu) : (0x80000000u | ((uint32_t)n & 0x7FFFFFFFu)));
    }
    if (inf.total > 0) {
        const uint64_t* seq = a.l1Seq + (int64_t)i * a.l1SeqStride;
        const int nChunks = (inf.nseq + kSeqChunk - 1) / kSeqChunk;
        for (int c = wave; c < (nChunks ? nChunks : 1); c += nW)
            seq_emit_write<kBack>(s, n, seq, kBack ? a.l1Bk + (int64_t)i * a.l1SeqStride : nullptr, inf.nseq, inf.lastAnchor, c, nChunks ? a.l1ChunkOff[(int64_t)i * a.l1MaxChunks + c] : 0u, out);
    } else if (!a.rawMode && n > 0) {
        const int slice = (((n + nW - 1) / nW) + 15) & ~15;
        const int off = wave * slice;
        if (off < n) wave_copy(out + off, s + off, min_(slice, n - off));
    }
}

// records with block checksums: xxh32 over the payload as stored (blk.go:98-102), one wave per block
__global__ __launch_bounds__(256) void k_l1_finish(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t stagebuf[4][4096];
    const int i = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= a.nBlocks) return;
    const int gi = a.blk0 + i;
    const SeqInfo inf = a.l1Info[i];
    const int c = inf.total ? inf.total : block_len(a, gi);
    uint8_t* rec = a.dst + (int64_t)gi * a.dstStride;
    if (a.bodyOff) {
        const int64_t off = a.bodyOff[gi];
        if (inf.nseq == kSeqEngineFailed || off + (int64_t)c + 8 > a.bodyCap) return;     // (not written: k_l1_write)
        rec = a.dst + off;
    }
    const uint32_t x = wave_xxh32_staged(rec + 4, c, stagebuf[threadIdx.x >> 6]);
    if ((threadIdx.x & 63u) == 0) { st32u(rec + 4 + c, x); if (!a.bodyOff) a.result[gi] = inf.nseq == kSeqEngineFailed ? PLZ4HIP_E_DEVICE : c + 8; }
}

// One level-1 block under a dictionary context and/or after another linked block.  Every input block of such a call has
// 64 KiB of scratch in front of it (host_codec lays the staging out that way): the external segment -- the previous block's
// tail or the dictionary -- is copied there, right before the block, and the full encoder runs in its external-segment mode
// (wave_encode_block_ext).  Only the dictionary-context lookup of blocks <= 4 KiB keeps the one-sequence-per-batch encoder.
__device__ __forceinline__ int encode_block_primed(const uint8_t* s, int n, uint8_t* out, int cap, const DictEnc& dc, uint32_t* lds)
{
    if (dc.mode == kDictCtxLookup) return wave_encode_block_dict(s, n, out, cap, dc, lds);
    int segLen = 0;
    if (dc.mode == kDictLoad || dc.mode == kDictCtxCopy) {
        segLen = dc.dictSize;
        wave_copy(const_cast<uint8_t*>(s) - segLen, dc.dict, segLen);
        WAVE_FENCE();
    }
    return wave_encode_block_ext(s, n, out, cap, dc.mode, segLen, dc.dictTable, lds);
}

// blk.CompressToBlk with a dictionary and/or linked blocks (config 5).  Which stream priming applies to a block follows
// clz4.go:160-179 (StreamIndieCtx) and :224-248 (StreamLinkedCtx) + async/writer.go:412-437 (_genDict).
template <int W> __global__ __launch_bounds__(64 * W) void k_encode_rec_dict(CodecArgs a)
{
    ENC_WAVE_TABLE(lds);
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int      n   = block_len(a, i);
        const uint8_t* s   = a.src + (int64_t)i * a.srcStride;
        uint8_t*       rec = a.dst + (int64_t)i * a.dstStride;
        DictEnc dc{nullptr, 0, kDictFreshPrefix, nullptr};
        const uint8_t* tail = nullptr; int tailLen = -1;
        if (a.linked) {
            if (i > 0) { const int pl = block_len(a, i - 1); tailLen = pl < 65536 ? pl : 65536; tail = a.src + (int64_t)(i - 1) * a.srcStride + (pl - tailLen); }
            else if (a.prevTailLen >= 0) { tail = a.prevTail; tailLen = a.prevTailLen; }
        }
        if (tailLen >= 0) {                                   // linked block after another block: LZ4_loadDict(previous tail)
            dc.mode = tailLen >= 8 ? kDictLoad : kDictNonePrefix;
            if (tailLen >= 8) { dc.dict = tail; dc.dictSize = tailLen; }
        } else if (a.dict != nullptr || a.dictLen >= 0) {     // a dictionary context is attached (possibly an empty one)
            if (a.dictLen >= 8) { dc.dict = a.dict; dc.dictSize = a.dictLen; dc.dictTable = a.dictTable; dc.mode = n > 4096 ? kDictCtxCopy : kDictCtxLookup; }
            else dc.mode = kDictNonePrefix;
        }                                                     // else: linked frame start without dictionary -> kDictFreshPrefix
        int      c    = encode_block_primed(s, n, rec + 4, a.bsz, dc, lds);
        uint32_t word = (uint32_t)c & 0x7FFFFFFFu;
        if (c == 0) { wave_copy(rec + 4, s, n); c = n; word = 0x80000000u | ((uint32_t)n & 0x7FFFFFFFu); }
        int len = c + 4;
        if (a.blockChecksum) {
            WAVE_FENCE();
            const uint32_t x = wave_xxh32(rec + 4, c);
            if ((threadIdx.x & 63u) == 0) st32u(rec + 4 + c, x);
            len += 4;
        }
        if ((threadIdx.x & 63u) == 0) { st32u(rec, word); a.result[i] = len; }
    }
}

// Raw LZ4 blocks with a dictionary context (StreamIndieCtx): the block API with WithBlockDictionary (plz4_block.go:48-53).
template <int W> __global__ __launch_bounds__(64 * W) void k_encode_raw_dict(CodecArgs a)
{
    ENC_WAVE_TABLE(lds);
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int n = block_len(a, i);
        DictEnc dc{nullptr, 0, kDictNonePrefix, nullptr};
        if (a.dictLen >= 8) { dc.dict = a.dict; dc.dictSize = a.dictLen; dc.dictTable = a.dictTable; dc.mode = n > 4096 ? kDictCtxCopy : kDictCtxLookup; }
        const int cap = a.dstCap ? a.dstCap[i] : a.dstCapAll;
        const int r = encode_block_primed(a.src + (int64_t)i * a.srcStride, n, a.dst + (int64_t)i * a.dstStride, cap, dc, lds);
        if ((threadIdx.x & 63u) == 0) a.result[i] = r;
    }
}

// One record against an optional dictionary; shared by the independent and the linked decode kernels.
__device__ __forceinline__ void decode_one_record(const CodecArgs& a, int i, const uint8_t* dict, int dictLen, int* rOut, int* stOut, bool* stored, uint8_t* dl)
{
    const uint8_t* rec    = a.recOff ? a.src + a.recOff[i] : a.src + (int64_t)i * a.srcStride;
    const int64_t  recLen = a.recOff ? a.recOff[i + 1] - a.recOff[i] : (int64_t)a.srcLen[i];
    uint8_t*       out    = a.dst + (int64_t)i * a.dstStride;
    const uint32_t word   = plz4_readfirstlane(ld32u(rec));
    const int      sz     = (int)(word & 0x7FFFFFFFu);
    int st = PLZ4HIP_BLK_OK, r = 0;
    *stored = false;
    if (sz > a.bsz || (int64_t)sz + 4 + (a.blockChecksum ? 4 : 0) > recLen) {
        st = PLZ4HIP_BLK_SIZE_OVERFLOW;
    } else {
        if (a.blockChecksum) {
            const uint32_t want = plz4_readfirstlane(ld32u(rec + 4 + sz));
            if (wave_xxh32(rec + 4, sz) != want) st = PLZ4HIP_BLK_HASH_MISMATCH;
        }
        if (st == PLZ4HIP_BLK_OK) {
            if (word & 0x80000000u) {
                if (sz > a.dstCapAll) st = PLZ4HIP_BLK_SIZE_OVERFLOW;
                else { wave_copy(out, rec + 4, sz); r = sz; *stored = true; }
            } else {
                r = wave_decode_block<true>(rec + 4, sz, out, a.dstCapAll, dict, dictLen, dl);    // the batches are assembled in LDS, as in k_decode_rec
                if (r < 0) st = PLZ4HIP_BLK_CORRUPT;
            }
        }
    }
    *rOut = r; *stOut = st;
}

// Independent blocks with a dictionary (indieDecompressorWithDict, compress/decompress.go:42-58, linked == false).
__global__ __launch_bounds__(64) void k_decode_rec_dict(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t dl[kDecLdsBytes];
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        int r, st; bool stored;
        decode_one_record(a, i, a.dict, a.dictLen, &r, &st, &stored, dl);
        if ((threadIdx.x & 63u) == 0) { a.result[i] = r; a.status[i] = st; }
    }
}

// Linked blocks: a serial chain, one wave.  The window follows compress.DictT.Update (compress/dict.go:28-41) and is
// NOT updated by stored blocks (sync/reader.go:75-78, async/reader.go:149-163) -- the reference's behaviour, kept.
__global__ __launch_bounds__(64) void k_decode_rec_linked(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t dl[kDecLdsBytes];
    // one wave per chain (frame); chains are independent of each other ("replicas"), the blocks of one chain are not
    for (int ch = next_block(a.queue); ch < a.nChains; ch = next_block(a.queue)) {
        const int first = a.chainFirst ? a.chainFirst[ch] : 0;
        const int last  = a.chainFirst ? a.chainFirst[ch + 1] : a.nBlocks;
        uint8_t* const win0 = a.window + (size_t)ch * 131072;
        uint8_t* winA = win0; uint8_t* winB = win0 + 65536;
        int winLen = a.windowLen[ch];
        bool dead = false;
        for (int i = first; i < last; ++i) {
            int r = 0, st = PLZ4HIP_BLK_CORRUPT; bool stored = false;
            if (!dead) decode_one_record(a, i, winA, winLen, &r, &st, &stored, dl);
            if ((threadIdx.x & 63u) == 0) { a.result[i] = r; a.status[i] = st; }
            if (st != PLZ4HIP_BLK_OK) { dead = true; continue; }             // first error ends the stream
            if (stored) continue;
            const uint8_t* out = a.dst + (int64_t)i * a.dstStride;
            WAVE_FENCE();
            if (r >= 65536) { wave_copy(winB, out + (r - 65536), 65536); winLen = 65536; }
            else {
                int keep = winLen;
                if (winLen + r > 65536) keep = 65536 - r;
                wave_copy(winB, winA + (winLen - keep), keep);
                wave_copy(winB + keep, out, r);
                winLen = keep + r;
            }
            WAVE_FENCE();
            uint8_t* t = winA; winA = winB; winB = t;
        }
        // leave the live window in the first half for the next call
        if (winA != win0) { WAVE_FENCE(); wave_copy(win0, winA, winLen); }
        if ((threadIdx.x & 63u) == 0) a.windowLen[ch] = winLen;
    }
}

__global__ __launch_bounds__(64) void k_decode_raw_dict(CodecArgs a)
{
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int cap = a.dstCap ? a.dstCap[i] : a.dstCapAll;
        const int r = wave_decode_block(a.src + (int64_t)i * a.srcStride, a.srcLen[i], a.dst + (int64_t)i * a.dstStride, cap, a.dict, a.dictLen);
        if ((threadIdx.x & 63u) == 0) a.result[i] = r;
    }
}

// HC levels 2..12 (config 4 = level 12): LZ4_compress_HC per block; mode 0 = raw LZ4 block, 1 = frame record.
__device__ __forceinline__ HcWork hc_work_of(const CodecArgs& a)
{
    uint8_t* ws = a.hcWork + (size_t)blockIdx.x * kHcWorkBytes;
    HcWork w; w.hash = (uint32_t*)ws; w.chain = (uint16_t*)(ws + kHcHashEntries * 4);
    w.opt = (HcOpt*)(ws + kHcHashEntries * 4 + kHcChainEntries * 2); w.pre = nullptr; w.rank = nullptr; w.list = nullptr;
    return w;
}
// How block i of an HC call is primed: clz4.StreamCtxHC (clz4.go:181-209, dictionary + independent blocks),
// clz4.StreamLinkedCtxHC (:250-283) + async/writer.go:412-437 (_genDict: the previous block's last <= 64 KiB).  In the
// dictionary / linked modes every input block has 64 KiB of scratch in front of it (host_codec lays the staging out that
// way): an external segment is copied there so that it lies right before the block, which is what hc_compress(kHcExt) reads.
__device__ __forceinline__ HcDict hc_dict_of(const CodecArgs& a, int i, int n, const uint8_t* s, bool rawApi)
{
    HcDict d; d.mode = kHcNone; d.len = 0; d.bytes = nullptr; d.hash = nullptr; d.chain = nullptr;
    const uint8_t* seg = nullptr; int segLen = -1;
    if (!rawApi && a.linked) {
        if (i > 0) { const int pl = block_len(a, i - 1); segLen = pl < 65536 ? pl : 65536; seg = a.src + (int64_t)(i - 1) * a.srcStride + (pl - segLen); }
        else if (a.prevTailLen >= 0) { seg = a.prevTail; segLen = a.prevTailLen; }
    }
    if (segLen < 0 && (a.dict != nullptr || a.dictLen >= 0)) {           // a dictionary context is attached (lz4hc.c:1438-1461)
        const int dl = a.dictLen > 0 ? a.dictLen : 0;
        if (n > 4096) { seg = a.dict; segLen = dl; }                      // its state is copied: same as LZ4_loadDictHC + setExternalDict
        else { d.mode = kHcCtx; d.len = dl; d.bytes = a.dict; d.hash = a.hcDictHash; d.chain = a.hcDictChain; }
    }
    if (segLen >= 0) {
        d.mode = kHcExt; d.len = segLen;
        if (segLen > 0) { wave_copy(const_cast<uint8_t*>(s) - segLen, seg, segLen); WAVE_FENCE(); }
    }
    return d;
}
// independent block without dictionary, the i-th of its group: its chain (and lists) were built up front when the call's
// launcher found room for them
__device__ __forceinline__ HcWork hc_with_pre(HcWork w, const CodecArgs& a, int i)
{
    w.pre = a.h12Chain ? a.h12Chain + (int64_t)i * a.h12ChainStride : nullptr;
    w.rank = (a.h12Chain && a.h12Rank) ? a.h12Rank + (int64_t)i * a.h12ChainStride : nullptr;
    w.list = (a.h12Chain && a.h12Rank) ? a.h12List + (int64_t)i * (a.h12ChainStride + 8) + 8 : nullptr;
    return w;
}
__global__ __launch_bounds__(64) void k_encode_raw_hc(CodecArgs a)
{
    const HcWork w = hc_work_of(a);
    const bool dictMode = a.hcEx != 0;
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int i = a.blk0 + g;                                        // (the chain / lists workspace is the group's: index g)
        const int cap = a.dstCap ? a.dstCap[i] : a.dstCapAll;
        const int n = block_len(a, i);
        const uint8_t* s = a.src + (int64_t)i * a.srcStride;
        int r;
        if (a.hcPfx && a.hcPfx[i] >= 0) continue;                          // (done on the list path)
        if (dictMode) r = hc_compress(s, n, a.dst + (int64_t)i * a.dstStride, cap, a.level, w, hc_dict_of(a, i, n, s, true));
        else          r = hc_compress(s, n, a.dst + (int64_t)i * a.dstStride, cap, a.level, hc_with_pre(w, a, g));
        if ((threadIdx.x & 63u) == 0) a.result[i] = r;
    }
}
__global__ __launch_bounds__(64) void k_encode_rec_hc(CodecArgs a)
{
    const HcWork w = hc_work_of(a);
    const bool exMode = a.hcEx != 0;
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int      i   = a.blk0 + g;
        const int      n   = block_len(a, i);
        const uint8_t* s   = a.src + (int64_t)i * a.srcStride;
        uint8_t*       rec = a.dst + (int64_t)i * a.dstStride;
        int c;                                                           // capacity == bsz (blk.go:73); indie.go:80-88
        if (a.hcPfx && a.hcPfx[i] >= 0) continue;                          // (done on the list path)
        if (exMode) c = hc_compress(s, n, rec + 4, a.bsz, a.level, w, hc_dict_of(a, i, n, s, false));
        else        c = hc_compress(s, n, rec + 4, a.bsz, a.level, hc_with_pre(w, a, g));
        uint32_t word = (uint32_t)c & 0x7FFFFFFFu;
        WAVE_FENCE();
        if (c == 0) { wave_copy(rec + 4, s, n); c = n; word = 0x80000000u | ((uint32_t)n & 0x7FFFFFFFu); }
        int len = c + 4;
        if (a.blockChecksum) {
            WAVE_FENCE();
            const uint32_t x = wave_xxh32(rec + 4, c);
            if ((threadIdx.x & 63u) == 0) st32u(rec + 4 + c, x);
            len += 4;
        }
        if ((threadIdx.x & 63u) == 0) { st32u(rec, word); a.result[i] = len; }
    }
}

// HC levels 3..12 with a dictionary and/or linked blocks on the list path (round 4): how every block of the call is primed
// (hc_dict_of: clz4.go:181-209, :250-283), its external segment copied right in front of it, its length noted.  One wave per block.
__global__ __launch_bounds__(64) void k_hc_ext_prep(CodecArgs a)
{
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int n = block_len(a, i);
        const uint8_t* s = a.src + (int64_t)i * a.srcStride;
        const HcDict d = hc_dict_of(a, i, n, s, a.rawMode != 0);
        // (no segment and no context -- the first block of a linked frame without a dictionary -- is an empty segment)
        const int pfx = d.mode == kHcCtx ? -1 : (d.mode == kHcExt ? d.len : 0);
        if ((threadIdx.x & 63u) == 0) a.hcPfx[i] = pfx;
    }
}
// the segment length of block i of an HC call (0: none), and the positions in front of the block that are never inserted
__device__ __forceinline__ int hc_pfx_of(const CodecArgs& a, int i) { return a.hcPfx ? a.hcPfx[i] : 0; }

// ---------------------------------------------------------------------------------------------- HC level 12, three phases
// Phase 1a: how many positions every hash has, then where its run starts in the list (exclusive prefix sum).  One 16-wave
// workgroup per block, the 32768 counters in LDS.
__global__ __launch_bounds__(1024) void k_hc12_hist(CodecArgs a)
{
    __shared__ uint32_t hist[kHcHashEntries];
    __shared__ uint32_t part[1024];
    __shared__ int cur;
    const int tid = (int)threadIdx.x;
    for (;;) {
        if (tid == 0) cur = (int)atomicAdd(a.queue, 1u);
        __syncthreads();
        const int g = cur;
        if (g >= a.nBlocks) break;
        const int i = a.blk0 + g;
        const int pfx = hc_pfx_of(a, i);                                   // (an external segment in front of the block: positions count from its first byte)
        if (pfx < 0) { __syncthreads(); continue; }
        const int n = pfx + block_len(a, i);
        const uint8_t* const src = a.src + (int64_t)i * a.srcStride - pfx;
        for (int h = tid; h < kHcHashEntries; h += 1024) hist[h] = 0u;
        __syncthreads();
        const int nIns = n >= 4 ? n - 3 : 0;
        const int skipLo = pfx > 3 ? pfx - 3 : 0;                           // the segment's last three positions are never inserted (lz4hc.c:1660-1678)
        for (int p = tid; p < nIns; p += 1024) if (p < skipLo || p >= pfx) atomicAdd(&hist[hc12_hash(ld32u(src + p))], 1u);
        __syncthreads();
        uint32_t sum = 0;
        for (int k = 0; k < 32; ++k) sum += hist[tid * 32 + k];
        part[tid] = sum;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t v = (tid >= d) ? part[tid - d] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        uint32_t run = part[tid] - sum;
        uint32_t* const off = a.h12Offsets + (size_t)g * kHcHashEntries;
        for (int k = 0; k < 32; ++k) { const uint32_t c = hist[tid * 32 + k]; off[tid * 32 + k] = run; run += c; }
        __syncthreads();
    }
}

// Phase 1b: chain, rank and list of every block of the group.  One wave per workgroup, two 64 KiB tables in LDS.
__global__ __launch_bounds__(64) void k_hc12_chain(CodecArgs a)
{
    __shared__ uint32_t lastT[kHcHashEntries / 2];
    __shared__ uint32_t curT[kHcHashEntries / 2];
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int i = a.blk0 + g;
        const int pfx = hc_pfx_of(a, i);
        if (pfx < 0) continue;
        const int n = pfx + block_len(a, i);
        int nPad = (n + 1 + 1023) & ~1023;
        if (nPad > a.h12ChainStride) nPad = (int)a.h12ChainStride;
        hc12_build_lists(a.src + (int64_t)i * a.srcStride - pfx, n, a.h12Offsets + (size_t)g * kHcHashEntries,
                         a.h12Chain + (int64_t)g * a.h12ChainStride, a.h12Rank + (int64_t)g * a.h12ChainStride,
                         a.h12List + (int64_t)g * (a.h12ChainStride + 8) + 8, nPad, lastT, curT, pfx > 3 ? pfx - 3 : 0, pfx);
    }
}

// The chain alone, for levels 3..11 (HcWork::pre): one pass per block with the whole 128 KiB table in LDS.
__global__ __launch_bounds__(64) void k_hc_chain(CodecArgs a)
{
    __shared__ uint32_t tab[kHcHashEntries];
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int i = a.blk0 + g;
        const int n = block_len(a, i);
        int nPad = (n + 1 + 1023) & ~1023;
        if (nPad > a.h12ChainStride) nPad = (int)a.h12ChainStride;
        hc12_build_chain(a.src + (int64_t)i * a.srcStride, n, a.h12Chain + (int64_t)g * a.h12ChainStride, nPad, tab);
    }
}

// Phase 2: F(p) for the positions of every block of the group.  One 16-wave workgroup per block at a time; the 64 KiB of source
// behind the positions in flight (and what lies ahead of them) sit in a 128 KiB ring in LDS, fed 1 KiB at a time; the chains
// come from the lists of phase 1.  Every lane runs one position (Hc12Walk); a loop trip runs each phase's code once for the
// lanes that are in it; a lane that has finished takes the next position from the block's queue, so lanes never wait for the
// longest search of their wave.
struct Hc12Ctl { int qNext; int loaded; int skipUntil; int lock; int waveMin[16]; int cur; };

__device__ __forceinline__ int h12_wave_min(int v)
{
    const int big = 0x7FFFFFFF;
    int t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x111, 0xF, 0xF, false); v = v < t ? v : t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x112, 0xF, 0xF, false); v = v < t ? v : t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x114, 0xF, 0xF, false); v = v < t ? v : t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x118, 0xF, 0xF, false); v = v < t ? v : t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x142, 0xA, 0xF, false); v = v < t ? v : t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x143, 0xC, 0xF, false); v = v < t ? v : t;
    return __builtin_amdgcn_readlane(v, 63);
}
#define H12_LD(x)     (*(volatile int*)&(x))
#define H12_ST(x, v)  (*(volatile int*)&(x) = (v))
#define H12_ORDER()   __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup")

__global__ __launch_bounds__(1024) void k_hc12_search(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kHc12SrcRing + kHc12SrcRingPad];
    __shared__ Hc12Ctl ctl;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (;;) {
        if (tid == 0) ctl.cur = (int)atomicAdd(a.queue, 1u);
        __syncthreads();
        const int g = ctl.cur;
        if (g >= a.nBlocks) break;
        const int i = a.blk0 + g;
        const int n = block_len(a, i);
        Hc12Tabs t;
        t.src = a.src + (int64_t)i * a.srcStride;
        t.chain = a.h12Chain + (int64_t)g * a.h12ChainStride;
        t.rank = a.h12Rank + (int64_t)g * a.h12ChainStride;
        t.list = a.h12List + (int64_t)g * (a.h12ChainStride + 8) + 8;
        Hc12F* const F = a.h12F + (int64_t)g * a.h12FStride;
        const int nPos = n - kMfLimit + 1 > 0 ? n - kMfLimit + 1 : 0;                // positions the parser can search
        if (tid == 0) { ctl.qNext = 0; ctl.loaded = 0; ctl.skipUntil = 0; ctl.lock = 0; }
        if (tid < 16) ctl.waveMin[tid] = 0;
        __syncthreads();

        Hc12SrcRing sw; sw.ring = ring; sw.g = t.src; sw.hi = 0;
        STAT_DECL;                              // diagnostics build: cycles and lane counts per section (scripts/stats_probe12.py)
        const unsigned long long tS0 = STAT_NOW(); (void)tS0;
        Hc12Walk<Hc12SrcRing> L;
        L.phase = kPhIdle;                      // kPhIdle: wants a position; kPhWait: holds p, its bytes are not in the ring yet
        int p = 0;
        unsigned idleTrips = 0;
        auto finish = [&]() {                   // the lane's search has ended: publish F(p)
            const Hc12F f = L.result();
            F[p] = f;
            if (f.len > kHc12Sufficient + 8) atomicMax(&ctl.skipUntil, p + f.len - kHc12Sufficient);
        };
        for (;;) {
            const uint64_t mF = __builtin_amdgcn_ballot_w64(L.phase == kPhFilter);
            const uint64_t mC = __builtin_amdgcn_ballot_w64(L.phase == kPhCount);
            const uint64_t mS = __builtin_amdgcn_ballot_w64(L.phase == kPhScan);
            const uint64_t mK = __builtin_amdgcn_ballot_w64(L.phase == kPhRank);
            const uint64_t mP = __builtin_amdgcn_ballot_w64(L.phase == kPhPattern);
            const uint64_t mI = __builtin_amdgcn_ballot_w64(L.phase == kPhIdle || L.phase == kPhWait);
            const bool busy = (mF | mC | mS | mK | mP) != 0;
            const unsigned long long tq0 = STAT_NOW(); (void)tq0;
            if (mI && (__builtin_popcountll(mI) >= a.h12Idle || !busy)) { SSTAT(13, 1);
                // (1) this wave's lower bound on the positions it holds or may still take: published BEFORE it takes new ones
                const int q0 = H12_LD(ctl.qNext);
                int m = h12_wave_min((L.phase != kPhIdle && L.phase != kPhDone) ? p : 0x7FFFFFFF);
                if (q0 < m) m = q0;
                if (lane == 0) H12_ST(ctl.waveMin[wv], m);
                H12_ORDER();
                // (2) keep the ring ahead of the queue: one wave at a time loads the next KiBs.  The ring must still hold the
                // 64 KiB behind every position in flight or yet to be taken: loaded - ring <= min(p) - 65535.
                int loaded = H12_LD(ctl.loaded);
                if (loaded < n && loaded < q0 + 8192) {
                    int got = 0;
                    if (lane == 0) got = atomicCAS(&ctl.lock, 0, 1) == 0;
                    if (__builtin_amdgcn_readfirstlane(got)) {
                        loaded = H12_LD(ctl.loaded);
                        const int qn = H12_LD(ctl.qNext);               // first the queue head, then the waves' bounds
                        H12_ORDER();
                        int mn = h12_wave_min(lane < 16 ? H12_LD(ctl.waveMin[lane]) : 0x7FFFFFFF);
                        if (qn < mn) mn = qn;
                        while (loaded < n && loaded < qn + 16384 && loaded + 1024 <= mn + 65536) {
                            const int at = loaded + lane * 16;
                            uint8_t* d = ring + (at & (kHc12SrcRing - 1));
                            if (at + 16 <= n) {
                                const uint4 v = *(const uint4*)(t.src + at);
                                *(uint4*)d = v;
                                if ((at & (kHc12SrcRing - 1)) == 0) *(uint4*)(ring + kHc12SrcRing) = v;           // the mirror behind the ring's end
                            } else for (int k = 0; k < 16; ++k) if (at + k < n) { d[k] = t.src[at + k]; if ((at & (kHc12SrcRing - 1)) == 0) ring[kHc12SrcRing + k] = d[k]; }
                            loaded += 1024;
                        }
                        H12_ORDER();
                        if (lane == 0) { H12_ST(ctl.loaded, loaded); H12_ORDER(); atomicExch(&ctl.lock, 0); }
                    }
                }
                // (3) idle lanes take the next positions
                const uint64_t idle = __builtin_amdgcn_ballot_w64(L.phase == kPhIdle);
                if (idle) {
                    const int first = __builtin_ctzll(idle);
                    int base = 0;
                    if (lane == first) base = atomicAdd(&ctl.qNext, __builtin_popcountll(idle));
                    base = __builtin_amdgcn_readlane(base, first);
                    if (L.phase == kPhIdle) {
                        p = base + __builtin_popcountll(idle & ((1ull << lane) - 1));
                        L.phase = p < nPos ? kPhWait : kPhDone;
                        // (the block's last positions -- no wide loads there -- are the parser's)
                        if (p < nPos && p + 32 > n) { Hc12F f; f.len = kHc12NotComputed; f.off = 0; F[p] = f; L.phase = kPhIdle; }
                    }
                }
                // (4) a lane starts its position once the bytes around it are in the ring; positions inside a match longer than
                // the parser's "sufficient" length are left to the parser (it jumps over them, lz4hc.c:1871-1882)
                loaded = H12_LD(ctl.loaded);
                const int skip = H12_LD(ctl.skipUntil);
                sw.hi = (uint32_t)(loaded < n ? loaded : n);
                if (L.phase == kPhWait) {
                    if (p < skip) { Hc12F f; f.len = kHc12NotComputed; f.off = 0; F[p] = f; L.phase = kPhIdle; }
                    else if (p + 64 <= loaded || loaded >= n) L.init(t, sw, n, p, (uint32_t)t.chain[p]);
                }
            }
            if (!__builtin_amdgcn_ballot_w64(L.phase != kPhDone)) break;
            SSTAT(12, STAT_NOW() - tq0);
            sw.hi = (uint32_t)min(H12_LD(ctl.loaded), n);
            // One phase's code per section.  A section costs the same whether 3 or 60 lanes are in it, so it runs when enough
            // lanes have gathered; when no phase has that many, the fullest one runs (nothing ever waits for good).
            {
                const int cF = __builtin_popcountll(mF), cC = __builtin_popcountll(mC), cS = __builtin_popcountll(mS),
                          cK = __builtin_popcountll(mK), cP = __builtin_popcountll(mP);
                int top = cF; top = cC > top ? cC : top; top = cS > top ? cS : top; top = cK > top ? cK : top; top = cP > top ? cP : top;
                const int need = top < a.h12Gather ? top : a.h12Gather;
                unsigned long long tx = STAT_NOW(); (void)tx;
                if (cF && cF >= need) {
                    SSTAT(0, 1); SSTAT(1, cF);
                    const bool fl = L.phase == kPhFilter;
                    if (__builtin_amdgcn_ballot_w64(fl && !L.is_near())) { if (fl && !L.is_near() && L.filter_trip<false>(t, sw)) finish(); }
                    if (fl && L.is_near() && L.filter_trip<true>(t, sw)) finish();
                    SSTAT(2, STAT_NOW() - tx); tx = STAT_NOW();
                }
                if (cC && cC >= need) { SSTAT(3, 1); SSTAT(4, cC); if (L.phase == kPhCount) L.count_trip(t, sw); SSTAT(5, STAT_NOW() - tx); tx = STAT_NOW(); }
                if (cS && cS >= need) { SSTAT(6, 1); SSTAT(7, cS); if (L.phase == kPhScan && L.scan_trip(t)) finish(); SSTAT(8, STAT_NOW() - tx); tx = STAT_NOW(); }
                if (cK && cK >= need) { SSTAT(9, 1); SSTAT(10, cK); if (L.phase == kPhRank) L.rank_trip(t); SSTAT(11, STAT_NOW() - tx); tx = STAT_NOW(); }
                if (cP && cP >= need) { if (L.phase == kPhPattern && L.pattern_trip(t, sw)) finish(); }
            }
            if (!busy) {
                __builtin_amdgcn_s_sleep(2);
                if (++idleTrips > (1u << 27)) { if (lane == 0) atomicExch(a.h12Err, 1); break; }      // never seen; bounds every spin
            } else idleTrips = 0;
            SSTAT(14, 1);
        }
        SSTAT(15, STAT_NOW() - tS0); SSTAT(16, 1);
        SSTAT_FLUSH();
        __syncthreads();
    }
}

// Phase 3: the parser + writer + record framing, one wave per block.  The first kH12OptLds entries of the price table are in LDS.
constexpr int kH12OptLds = 1024;
__global__ __launch_bounds__(64) void k_hc12_parse(CodecArgs a)
{
    __shared__ Hc12Ent  oEnt[kH12OptLds];
    __shared__ uint64_t seqs[64];
    Hc12Ws w;
    w.ent = oEnt; w.nl = kH12OptLds; w.seq = seqs;
    {
        uint8_t* gws = a.h12Ws + (size_t)blockIdx.x * kHc12WsGlobalBytes;
        w.gprice = (int*)gws; w.glitlen = (int*)(gws + kHc12OptEntries * 4); w.gmloff = (uint32_t*)(gws + kHc12OptEntries * 8);
    }
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int i = a.blk0 + g;
        const int      n   = block_len(a, i);
        const uint8_t* s   = a.src + (int64_t)i * a.srcStride;
        const Hc12F*   F   = a.h12F + (int64_t)g * a.h12FStride;
        const uint16_t* ch = a.h12Chain + (int64_t)g * a.h12ChainStride;
        if (plz4_readfirstlane(*(volatile int32_t*)a.h12Err)) {               // the search kernel gave up on this group (spin guard):
            if ((threadIdx.x & 63u) == 0) a.result[i] = PLZ4HIP_E_DEVICE;     // its F is not to be trusted -- an engine failure, not a result
            continue;
        }
        if (a.rawMode) {
            const int cap = a.dstCap ? a.dstCap[i] : a.dstCapAll;
            const int r = hc12_parse(s, n, a.dst + (int64_t)i * a.dstStride, cap, F, ch, w);
            if ((threadIdx.x & 63u) == 0) a.result[i] = r;
            continue;
        }
        uint8_t* rec = a.dst + (int64_t)i * a.dstStride;
        int c = hc12_parse(s, n, rec + 4, a.bsz, F, ch, w);              // capacity == bsz (blk.go:73); indie.go:80-88
        uint32_t word = (uint32_t)c & 0x7FFFFFFFu;
        WAVE_FENCE();
        if (c == 0) { wave_copy(rec + 4, s, n); c = n; word = 0x80000000u | ((uint32_t)n & 0x7FFFFFFFu); }
        int len = c + 4;
        if (a.blockChecksum) {
            WAVE_FENCE();
            const uint32_t x = wave_xxh32(rec + 4, c);
            if ((threadIdx.x & 63u) == 0) st32u(rec + 4 + c, x);
            len += 4;
        }
        if ((threadIdx.x & 63u) == 0) { st32u(rec, word); a.result[i] = len; }
    }
}

// Levels 3..9 on independent blocks (lz4hc_lazy_device.inl), the chain and the lists being there.  k_hc_lazy: one wave per
// (block, segment) walks its segment and writes records; k_hc_stitch: one wave per block joins the segments' walks into the
// one walk from position 0; k_hc_gather lays the pieces out as one array; the emit kernels of level 1 (k_l1_sizes<false> ..
// k_l1_finish) make the block from it.
__device__ __forceinline__ HcWork lz_work(const CodecArgs& a, int g)
{
    HcWork w; w.hash = nullptr; w.chain = nullptr; w.opt = nullptr; w.pre = nullptr; w.rank = nullptr; w.list = nullptr;
    if (a.level >= 10) w.opt = hc_work_of(a).opt;                           // levels 10..11: the wave's price table (its slot of the HC workspace)
    return hc_with_pre(w, a, g);
}
// kD: the call's blocks have external segments in front of them (a.hcPfx: dictionary / linked blocks, every level 3..12; lz4hc.c:
// 1438-1461, :1626-1720); chain and lists then cover segment + block.  A kernel of its own, so that the independent blocks'
// kernel does not carry the segment rules (and level 12's parameters).
template <bool kD> __global__ __launch_bounds__(64) void k_hc_lazy(CodecArgs a)
{
    const int items = a.nBlocks * a.lzSegs;
    const bool broken = plz4_readfirstlane(*(volatile int32_t*)a.h12Err) != 0;
    for (int it = next_block(a.queue); it < items; it = next_block(a.queue)) {
        const int g = it / a.lzSegs, j = it - g * a.lzSegs;                  // (a block's segments are taken one after the other)
        const int i = a.blk0 + g;
        const int n = block_len(a, i);
        const int pfx = kD ? hc_pfx_of(a, i) : 0;
        if (broken || n < 0 || n > a.l1MaxLen || pfx < 0) continue;
        const int segs = lz_segments(n, a.lzSegs, a.lzMinSeg);
        if (j >= segs) continue;
        hc_lazy_segment<kD>(a.src + (int64_t)i * a.srcStride, n, a.level, lz_work(a, g), segs, j,
                            a.lzRec + (int64_t)g * a.lzRecStride, a.lzMeta + (int64_t)g * a.lzSegs, a.lzStarts + (int64_t)g * a.lzSegs * kLzStarts, pfx);
    }
}
template <bool kD> __global__ __launch_bounds__(64) void k_hc_stitch(CodecArgs a)
{
    const bool broken = plz4_readfirstlane(*(volatile int32_t*)a.h12Err) != 0;
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int i = a.blk0 + g;
        const int n = block_len(a, i);
        LzPiece* const pieces = a.lzPieces + (int64_t)g * 2 * a.lzSegs;
        int lastAnchor = 0, nseq = -1, segs = 0;                             // -1: a block the workspace was not sized for, or one that is not this path's
        const int pfx = kD ? hc_pfx_of(a, i) : 0;
        if (broken) nseq = kSeqEngineFailed;                                 // the search phase gave up: its F is not to be trusted
        else if (n >= 0 && n <= a.l1MaxLen && pfx >= 0) {
            segs = lz_segments(n, a.lzSegs, a.lzMinSeg);
            nseq = hc_lazy_stitch<kD>(a.src + (int64_t)i * a.srcStride, n, a.level, lz_work(a, g), segs,
                                      a.lzRec + (int64_t)g * a.lzRecStride, a.lzBridge + (int64_t)g * a.lzRecStride,
                                      a.lzMeta + (int64_t)g * a.lzSegs, a.lzStarts + (int64_t)g * a.lzSegs * kLzStarts, pieces, &lastAnchor, pfx);
        }
        if ((threadIdx.x & 63u) == 0) {
            for (int k = 2 * segs; k < 2 * a.lzSegs; ++k) pieces[k].cnt = 0;
            SeqInfo inf; inf.nseq = nseq; inf.lastAnchor = lastAnchor; inf.total = 0; inf.stored = 0; a.l1Info[g] = inf;
        }
    }
}
// grid (workgroups per block, blocks of the group)
__global__ __launch_bounds__(256) void k_hc_gather(CodecArgs a)
{
    const int g = blockIdx.y;
    const LzPiece* const pieces = a.lzPieces + (int64_t)g * 2 * a.lzSegs;
    uint64_t* const seq = a.l1Seq + (int64_t)g * a.l1SeqStride;
    const int t = blockIdx.x * 256 + (int)threadIdx.x, nT = gridDim.x * 256;
    for (int k = 0; k < 2 * a.lzSegs; ++k) {
        const LzPiece pc = pieces[k];
        for (int e = t; e < pc.cnt; e += nT) seq[pc.dst + e] = pc.src[e];
    }
}

// Level 12 up to 4 MiB: its parser (the price DP over F, lz4hc12_device.inl) walked in segments and stitched like levels 3..11,
// records out.  The price table's LDS part is 256 entries here (4 KiB: windows beyond that spill to the wave's global slot), so
// that registers, not LDS, bound the waves per CU.
constexpr int kH12SegLds = 256;
__device__ __forceinline__ Hc12Ws h12_seg_ws(const CodecArgs& a, Hc12Ent* ent, uint64_t* seqs)
{
    Hc12Ws w; w.ent = ent; w.nl = kH12SegLds; w.seq = seqs;
    uint8_t* gws = a.h12Ws + (size_t)blockIdx.x * kHc12WsGlobalBytes;
    w.gprice = (int*)gws; w.glitlen = (int*)(gws + kHc12OptEntries * 4); w.gmloff = (uint32_t*)(gws + kHc12OptEntries * 8);
    return w;
}
__global__ __launch_bounds__(64) void k_hc12_seg(CodecArgs a)
{
    __shared__ Hc12Ent oEnt[kH12SegLds];
    __shared__ uint64_t seqs[64];
    const Hc12Ws w = h12_seg_ws(a, oEnt, seqs);
    const int items = a.nBlocks * a.lzSegs;
    const bool broken = plz4_readfirstlane(*(volatile int32_t*)a.h12Err) != 0;
    for (int it = next_block(a.queue); it < items; it = next_block(a.queue)) {
        const int g = it / a.lzSegs, j = it - g * a.lzSegs;
        const int i = a.blk0 + g;
        const int n = block_len(a, i);
        if (broken || n < 0 || n > a.l1MaxLen) continue;
        const int segs = lz_segments(n, a.lzSegs, a.lzMinSeg);
        if (j >= segs) continue;
        hc12_segment(a.src + (int64_t)i * a.srcStride, n, a.h12F + (int64_t)g * a.h12FStride, a.h12Chain + (int64_t)g * a.h12ChainStride, w, segs, j,
                     a.lzRec + (int64_t)g * a.lzRecStride, a.lzMeta + (int64_t)g * a.lzSegs, a.lzStarts + (int64_t)g * a.lzSegs * kLzStarts);
    }
}
__global__ __launch_bounds__(64) void k_hc12_stitch(CodecArgs a)
{
    __shared__ Hc12Ent oEnt[kH12SegLds];
    __shared__ uint64_t seqs[64];
    const Hc12Ws w = h12_seg_ws(a, oEnt, seqs);
    const bool broken = plz4_readfirstlane(*(volatile int32_t*)a.h12Err) != 0;
    for (int g = next_block(a.queue); g < a.nBlocks; g = next_block(a.queue)) {
        const int i = a.blk0 + g;
        const int n = block_len(a, i);
        LzPiece* const pieces = a.lzPieces + (int64_t)g * 2 * a.lzSegs;
        int lastAnchor = 0, nseq = -1, segs = 0;
        if (broken) nseq = kSeqEngineFailed;                                 // the search kernel gave up (spin guard): its F is not to be trusted
        else if (n >= 0 && n <= a.l1MaxLen) {
            segs = lz_segments(n, a.lzSegs, a.lzMinSeg);
            nseq = hc12_stitch(a.src + (int64_t)i * a.srcStride, n, a.h12F + (int64_t)g * a.h12FStride, a.h12Chain + (int64_t)g * a.h12ChainStride, w, segs,
                               a.lzRec + (int64_t)g * a.lzRecStride, a.lzBridge + (int64_t)g * a.lzRecStride,
                               a.lzMeta + (int64_t)g * a.lzSegs, a.lzStarts + (int64_t)g * a.lzSegs * kLzStarts, pieces, &lastAnchor);
        }
        if ((threadIdx.x & 63u) == 0) {
            for (int k = 2 * segs; k < 2 * a.lzSegs; ++k) pieces[k].cnt = 0;
            SeqInfo inf; inf.nseq = nseq; inf.lastAnchor = lastAnchor; inf.total = 0; inf.stored = 0; a.l1Info[g] = inf;
        }
    }
}

// Level 2 on independent blocks: the batch walk over the two tables (hc_mid_parse, lz4hc_lazy_device.inl), one wave per block,
// the tables in the wave's slot of the HC workspace; records out, the emit kernels of level 1 behind it.
// kD: blocks behind external segments (a.hcPfx; dictionary / linked blocks): the tables start from LZ4MID_fillHTable over the segment
template <bool kD> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hc_mid(CodecArgs a)
{
    uint32_t* const tabs = (uint32_t*)(a.hcWork + (size_t)blockIdx.x * kHcWorkBytes);
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        const int gi = a.blk0 + i;
        const int n  = block_len(a, gi);
        const int pfx = kD ? hc_pfx_of(a, gi) : 0;
        int lastAnchor = 0, nseq = -1;                                       // -1: a block the workspace was not sized for, or one that is not this path's
        if (n >= 0 && n <= a.l1MaxLen && pfx >= 0)
            nseq = hc_mid_parse<kD>(a.src + (int64_t)gi * a.srcStride, n, tabs, tabs + 16384, a.l1Seq + (int64_t)i * a.l1SeqStride, &lastAnchor, pfx);
        if ((threadIdx.x & 63u) == 0) { SeqInfo inf; inf.nseq = nseq; inf.lastAnchor = lastAnchor; inf.total = 0; inf.stored = 0; a.l1Info[i] = inf; }
    }
}

// clz4.NewDictCtxHC (clz4.go:122-147): workgroup 0 builds the level-2 (lz4mid) tables of a dictionary, workgroup 1 the
// hash-chain tables every other level shares.  tabs = 2 x kHcWorkBytes.
__global__ __launch_bounds__(64) void k_hc_dict_prime(const uint8_t* dict, int len, uint8_t* tabs)
{
    uint8_t* ws = tabs + (size_t)blockIdx.x * kHcWorkBytes;
    HcWork w; w.hash = (uint32_t*)ws; w.chain = (uint16_t*)(ws + kHcHashEntries * 4);
    w.opt = (HcOpt*)(ws + kHcHashEntries * 4 + kHcChainEntries * 2); w.pre = nullptr; w.rank = nullptr; w.list = nullptr;
    hc_prime_dict(dict, len, blockIdx.x == 0 ? 2 : 3, w);
}

// ---- LZ4_decompress_safe of a few blocks by the whole chip (lz4_dx_device.inl).  grid (segments, blocks) for the per-segment
// kernels, (chunks of 1024 output bytes, blocks) for the per-byte ones; a block that any stage flags is decoded by k_decode_raw behind.
__device__ __forceinline__ int dx_cap(const CodecArgs& a, int b) { return a.dstCap ? a.dstCap[b] : a.dstCapAll; }
__device__ __forceinline__ const uint8_t* dx_src(const CodecArgs& a, int b) { return a.src + (a.dxSrcOff ? a.dxSrcOff[b] : (int64_t)b * a.srcStride); }
__device__ __forceinline__ int dx_len(const CodecArgs& a, int b) { return a.dxLen ? a.dxLen[b] : a.srcLen[b]; }
__global__ __launch_bounds__(64) void k_dx_tables(CodecArgs a)
{
    const int j = blockIdx.x, b = blockIdx.y;
    const int n = dx_len(a, b);
    {   // the pointer table starts as the identity: this workgroup's share of it
        const int64_t share = (a.dxPtrStride + gridDim.x - 1) / gridDim.x;
        uint32_t* const ptr = a.dxPtr + (int64_t)b * a.dxPtrStride;
        const int64_t lo = (int64_t)j * share, hi = lo + share < a.dxPtrStride ? lo + share : a.dxPtrStride;
        for (int64_t p = lo + (threadIdx.x & 63u); p < hi; p += 64) ptr[p] = (uint32_t)p;
    }
    if (n <= 0 || (int64_t)j * kDxSeg >= n || n > a.dxTStride - 64) return;           // (a block beyond what the tables were sized for is flagged by the stitch)
    dx_segment_table(dx_src(a, b), n, j, a.dxT + (int64_t)b * a.dxTStride);
}
__global__ __launch_bounds__(64) void k_dx_stitch(CodecArgs a)
{
    const int b = blockIdx.x;
    const int n = dx_len(a, b);
    const int nseg = dx_segments(n);
    int bad = (nseg > a.dxMaxSeg || n > a.dxTStride - 64 || dx_cap(a, b) > a.dxPtrStride - 64) ? 1 : dx_stitch(dx_src(a, b), n, dx_cap(a, b), a.dxT + (int64_t)b * a.dxTStride,
                                                a.dxUnits + (int64_t)b * a.dxMaxSeg, nseg);
    if ((threadIdx.x & 63u) == 0) {
        DxInfo inf; inf.bad = bad; inf.outLen = 0; inf.tailFrom = dx_tail_from(nseg); inf.pad = 0;
        for (int r = 0; r <= kDxRounds; ++r) inf.moved[r] = 0;
        a.dxInfo[b] = inf;
    }
}
__global__ __launch_bounds__(64) void k_dx_fill(CodecArgs a)
{
    const int j = blockIdx.x, b = blockIdx.y;
    DxInfo* const inf = a.dxInfo + b;
    if (inf->bad || j > inf->tailFrom) return;
    const int jt = inf->tailFrom;
    const DxUnit* const units = a.dxUnits + (int64_t)b * a.dxMaxSeg;
    const DxUnit u = units[j];
    if (j < jt && u.ip < 0) return;
    const int64_t r = wave_dx_fill(dx_src(a, b), dx_len(a, b), a.dst + (int64_t)b * a.dstStride, dx_cap(a, b),
                                   a.dxPtr + (int64_t)b * a.dxPtrStride, u.ip, u.op, u.stop, j == jt);
    bool ok = r >= 0;
    if (ok && j < jt) {                                              // where this unit stops is where the next one starts
        int k = j + 1; while (k < jt && units[k].ip < 0) ++k;
        ok = units[k].op == (int)r;
    }
    if ((threadIdx.x & 63u) == 0) { if (!ok) atomicOr(&inf->bad, 1); else if (j == jt) inf->outLen = (int)r; }
}
__global__ __launch_bounds__(256) void k_dx_jump(CodecArgs a)
{
    const int b = blockIdx.y, r = a.dxRound;
    DxInfo* const inf = a.dxInfo + b;
    if (inf->bad || (r > 0 && !inf->moved[r - 1])) return;
    const int outLen = inf->outLen, p0 = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 256;
    if (p0 >= outLen) return;
    if (dx_jump(a.dxPtr + (int64_t)b * a.dxPtrStride, p0, outLen) && (threadIdx.x & 63u) == 0) inf->moved[r] = 1u;
}
__global__ __launch_bounds__(256) void k_dx_gather(CodecArgs a)
{
    const int b = blockIdx.y;
    const DxInfo* const inf = a.dxInfo + b;
    if (inf->bad) return;
    const int outLen = inf->outLen, p0 = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 256;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // (records: the payload's checksum is verified beside this path, k_dx_rec_hash; frame.go:114-127 rejects before it decodes)
        const bool hashBad = a.dxHashBad && a.dxHashBad[b];
        a.result[b] = hashBad ? 0 : outLen;
        if (a.status) a.status[b] = hashBad ? PLZ4HIP_BLK_HASH_MISMATCH : PLZ4HIP_BLK_OK;
    }
    if (p0 >= outLen) return;
    dx_gather(a.dst + (int64_t)b * a.dstStride, a.dxPtr + (int64_t)b * a.dxPtrStride, p0, outLen);
}

// records on the few-block path: which records are compressed payloads of a sane size (FrameReader._read's checks, blk/frame.go:
// 79-85; everything else -- stored blocks, size overflows -- is the one-wave kernel's, which runs for the flagged blocks behind)
__global__ __launch_bounds__(256) void k_dx_rec_prep(CodecArgs a, int64_t* srcOff, int32_t* len)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nBlocks) return;
    const int64_t off = a.recOff ? a.recOff[i] : (int64_t)i * a.srcStride;
    const int64_t recLen = a.recOff ? a.recOff[i + 1] - a.recOff[i] : (int64_t)a.srcLen[i];
    int n = -1;
    if (recLen >= 4) {
        const uint32_t word = ld32u(a.src + off);
        const int sz = (int)(word & 0x7FFFFFFFu);
        if (!(word & 0x80000000u) && sz <= a.bsz && (int64_t)sz + 4 + (a.blockChecksum ? 4 : 0) <= recLen) n = sz;
    }
    srcOff[i] = off + 4; len[i] = n;
}
// ... and their block checksums (frame.go:114-127), one wave per record, on a stream of its own beside the decode
__global__ __launch_bounds__(64) void k_dx_rec_hash(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t stagebuf[4096];
    const int b = blockIdx.x;
    const int n = a.dxLen[b];
    int bad = 0;
    if (n >= 0) {
        const uint8_t* p = a.src + a.dxSrcOff[b];
        bad = wave_xxh32_staged(p, n, stagebuf) != plz4_readfirstlane(ld32u(p + n));
    }
    if ((threadIdx.x & 63u) == 0) a.dxHashBad[b] = bad;
}

__global__ __launch_bounds__(64) void k_decode_raw(CodecArgs a)
{
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        if (a.dxInfo && !a.dxInfo[i].bad) continue;                          // (answered by the few-block path)
        const int n   = a.srcLen[i];
        const int cap = a.dstCap ? a.dstCap[i] : a.dstCapAll;
        const int r   = wave_decode_block(a.src + (int64_t)i * a.srcStride, n, a.dst + (int64_t)i * a.dstStride, cap);
        if ((threadIdx.x & 63u) == 0) a.result[i] = r;
    }
}

// FrameReader._read's per-block checks + BlkT.Decompress on the device (blk/frame.go:54-127, blk.go:50-61).
// Record i starts at src + recOff[i] when recOff is given, else at src + i*srcStride with srcLen[i] bytes.
// kDx: behind the few-block path (launch_decode): only the records that path has flagged.  A template parameter so that the
// kernels of the bulk path -- k_decode_rec, k_l1_duplex: bound by their scalar registers -- do not carry the test.
template <bool kDx = false>
__device__ __forceinline__ void decode_rec_loop(const CodecArgs& a, uint8_t* dl)
{
    for (int i = next_block(a.queue); i < a.nBlocks; i = next_block(a.queue)) {
        if (kDx && !a.dxInfo[i].bad) continue;                               // (answered by the few-block path)
        const uint8_t* rec    = a.recOff ? a.src + a.recOff[i] : a.src + (int64_t)i * a.srcStride;
        const int64_t  recLen = a.recOff ? a.recOff[i + 1] - a.recOff[i] : (int64_t)a.srcLen[i];
        uint8_t*       out    = a.dst + (int64_t)i * a.dstStride;
        const uint32_t word   = plz4_readfirstlane(ld32u(rec));
        const int      sz     = (int)(word & 0x7FFFFFFFu);
        int st = PLZ4HIP_BLK_OK, r = 0;
        if (sz > a.bsz || (int64_t)sz + 4 + (a.blockChecksum ? 4 : 0) > recLen) {
            st = PLZ4HIP_BLK_SIZE_OVERFLOW;
        } else {
            if (a.blockChecksum) {
                const uint32_t want = plz4_readfirstlane(ld32u(rec + 4 + sz));
                if (wave_xxh32(rec + 4, sz) != want) st = PLZ4HIP_BLK_HASH_MISMATCH;
            }
            if (st == PLZ4HIP_BLK_OK) {
                if (word & 0x80000000u) {                                    // stored block: straight copy
                    const int cap = a.dstCapAll;
                    if (sz > cap) { st = PLZ4HIP_BLK_SIZE_OVERFLOW; }
                    else { wave_copy(out, rec + 4, sz); r = sz; }
                } else {
                    r = wave_decode_block<true>(rec + 4, sz, out, a.dstCapAll, nullptr, 0, dl);
                    if (r < 0) st = PLZ4HIP_BLK_CORRUPT;
                }
            }
        }
        if ((threadIdx.x & 63u) == 0) { a.result[i] = r; a.status[i] = st; }
    }
}
#if defined(PLZ4_EXP_DEC_EU)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 6))) void k_decode_rec(CodecArgs a)
#else
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_decode_rec(CodecArgs a)
#endif
{
    __shared__ __attribute__((aligned(16))) uint8_t dl[kDecLdsBytes];       // the vector path assembles each batch's output here
    decode_rec_loop<false>(a, dl);
}
__global__ __launch_bounds__(64) void k_decode_rec_dx(CodecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t dl[kDecLdsBytes];
    decode_rec_loop<true>(a, dl);
}

// The level-1 parse of one call and the record decode of another in ONE launch (plz4hip_dev_duplex_records).  The parser is a
// serial chain per block that leaves half of a SIMD's VALU issue slots idle at the 2.5 waves per SIMD its LDS tables allow
// (profiles/r03_summary.json: 51 %), and the decoder is bound by exactly those slots (85-92 %) and needs 1.1 KiB of LDS per
// wave.  Two kernels on two streams do not share a CU that way: the parser's workgroup takes all of a CU's LDS.  So a
// workgroup here is P parser waves (a 16 KiB table each) + D decoder waves -- by default 1 + 1, nine workgroups per CU: 9
// parsers and 9 decoders -- and the parser waves run at a raised priority: the decoders issue when a parser waits.
// Each role pulls block ids from its own queue until it is empty; the waves of a workgroup never synchronise.
// (P parser waves + D decoder waves per workgroup: experiment switch PLZ4HIP_DUPLEX="P,D"; PRIO: the parser waves' s_setprio,
// PLZ4HIP_DUPLEX_PRIO)
constexpr int kWinScratch = 256;          // per parser wave: the dwords its lanes park for the windows of the batch after next
constexpr int duplex_wgs_per_cu(int P, int D) { return (160 * 1024) / (P * (kHashBytes + kWinScratch) + D * kDecLdsBytes); }
#if defined(PLZ4_EXP_DUPLEX_EU)
constexpr int duplex_waves_per_eu(int P, int D) { return PLZ4_EXP_DUPLEX_EU; }       // EXPERIMENT: forced occupancy target
#else
constexpr int duplex_waves_per_eu(int P, int D) { return ((P + D) * duplex_wgs_per_cu(P, D) + 3) / 4; }
#endif
template <int P, int D, int PRIO> __global__ __launch_bounds__(64 * (P + D))
__attribute__((amdgpu_waves_per_eu(duplex_waves_per_eu(P, D), duplex_waves_per_eu(P, D))))
void k_l1_duplex(CodecArgs a, CodecArgs d)
{
    __shared__ uint32_t tabS[P][kHashBytes / 4];
    __shared__ __attribute__((aligned(16))) uint8_t dlS[D][kDecLdsBytes];
    __shared__ __attribute__((aligned(16))) uint8_t winS[P][kWinScratch];
    const int w = plz4_readfirstlane((int)(threadIdx.x >> 6));
    if (w < P) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        l1_parse_loop<1>(a, tabS[w], winS[w]);
    } else {
        decode_rec_loop<false>(d, dlS[w - P]);
    }
}

__global__ __launch_bounds__(64) void k_xxh32(const uint8_t* base, int64_t stride, const int32_t* len, uint32_t* out,
                                              int n, uint32_t* queue)
{
    for (int i = next_block(queue); i < n; i = next_block(queue)) {
        const uint32_t x = wave_xxh32(base + (int64_t)i * stride, len[i]);
        if ((threadIdx.x & 63u) == 0) out[i] = x;
    }
}

// Streaming content checksum (xxh32.XXHZero): one wave writes the buffers base + i*stride (lens[i] bytes each, or `single`
// bytes at base when lens is null) into the stream state, in order.  lens may be a result array: entries <= 0 are skipped.
__global__ __launch_bounds__(64) void k_xxh32_stream(XxhStream* st, const uint8_t* base, int64_t stride, const int32_t* lens, int n, int64_t single, int reset)
{
    if (reset) wave_xxh32_stream_reset(st);
    if (!lens) { wave_xxh32_stream_update(st, base, single); return; }
    for (int i = 0; i < n; ++i) {
        const int len = plz4_readfirstlane(lens[i]);
        if (len > 0) wave_xxh32_stream_update(st, base + (int64_t)i * stride, len);
    }
}

// Exclusive prefix sum int32 -> int64, one workgroup.
__global__ __launch_bounds__(1024) void k_scan(const int32_t* __restrict__ len, int64_t* __restrict__ off, int n)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(t * per, n), hi = min(lo + per, n);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += len[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int64_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = part[t] - s;
    for (int i = lo; i < hi; ++i) { off[i] = run; run += len[i]; }
    if (t == 1023) off[n] = part[1023];
}

// ... continued behind an earlier part: off[0] holds where this part starts (the total the scan of the part before left there);
// lengths that are not positive (an engine failure code) count as 0
__global__ __launch_bounds__(1024) void k_scan_from(const int32_t* __restrict__ len, int64_t* __restrict__ off, int n, int first)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(t * per, n), hi = min(lo + per, n);
    const int64_t base = first ? 0 : off[0];
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += len[i] > 0 ? len[i] : 0;
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int64_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = base + part[t] - s;
    for (int i = lo; i < hi; ++i) { off[i] = run; run += len[i] > 0 ? len[i] : 0; }
    if (t == 1023) off[n] = base + part[1023];
}

// Bytes to hand back per block of a host call: the result if positive (sizes), nothing for 0 / error codes.
// plz4hip_dev_compress: the lengths live on the device and the workspaces are sized from the caller's maxLen.  A private copy with
// every length outside [0, maxLen] replaced by 0 is what the kernels see; such a block's result becomes PLZ4HIP_E_ARG afterwards.
__global__ __launch_bounds__(256) void k_check_len(const int32_t* __restrict__ srcLen, int maxLen, int n, int32_t* __restrict__ lenOut, int32_t* __restrict__ result, int after)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool bad = srcLen[i] < 0 || srcLen[i] > maxLen;
    if (!after) lenOut[i] = bad ? 0 : srcLen[i];
    else if (bad) result[i] = PLZ4HIP_E_ARG;
}

__global__ __launch_bounds__(256) void k_out_len(const int32_t* __restrict__ res, int32_t* __restrict__ len, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) len[i] = res[i] > 0 ? res[i] : 0;
}

// Record mover: dst + dstOff[i] <- src + (srcOff ? srcOff[i] : i*stride), len[i] bytes.  Compaction of the staging
// area into the frame body is the srcOff == nullptr case.  grid = (n, slices); 256 threads move 16 bytes each per
// step (unaligned on both sides is fine on gfx950: global_load/store_dwordx4).
__global__ __launch_bounds__(256) void k_move_records(const uint8_t* __restrict__ src, const int64_t* __restrict__ srcOff,
                                                      int64_t stride, const int32_t* __restrict__ len,
                                                      const int64_t* __restrict__ dstOff, uint8_t* __restrict__ dst,
                                                      int64_t dstCap)
{
    const int      i = blockIdx.x;
    const int      n = len[i];
    if (n <= 0 || dstOff[i] < 0 || dstOff[i] + n > dstCap) return;      // (an overflow shows to the caller as recOff[n] > bodyCap)
    const uint8_t* s = src + (srcOff ? srcOff[i] : (int64_t)i * stride);
    uint8_t*       d = dst + dstOff[i];
    const int full = n & ~15;
    for (int p = (blockIdx.y * 256 + threadIdx.x) * 16; p < full; p += gridDim.y * 256 * 16)
        *(v16u_t*)(d + p) = *(const v16u_t*)(s + p);
    if (blockIdx.y == 0 && (int)threadIdx.x < (n - full)) d[full + threadIdx.x] = s[full + threadIdx.x];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ context
struct plz4hip_xxh32_stream;
struct plz4hip_ctx {
    int          device = 0;
    hipStream_t  stream = nullptr;
    std::mutex   mu;
    std::mutex   errMu;                // guards err alone: argument checks report before `mu` is taken
    std::string  err;
    uint32_t*    d_queues = nullptr;   // ring of work-queue counters
    int          qslot = 0;
    int          cus = 0;
    int          encWaves = 0, decWaves = 0;
    // host-API staging (grown on demand): a ring of chunks in flight, each with pinned host memory, device memory and a stream
    // level 1 in stages: sequence records + chunk tables of one group of blocks (launch_l1)
    struct L1Ws { uint8_t* d = nullptr; size_t bytes = 0; };
    struct HostSlot { uint8_t* h = nullptr; size_t hcap = 0; uint8_t* d = nullptr; size_t dcap = 0; hipStream_t s = nullptr;
                      hipEvent_t evData = nullptr, evHash = nullptr; bool hashBusy = false; L1Ws l1; };
    plz4hip_xxh32_stream* contentHash = nullptr;   // plz4hip_ctx_set_content_hash
    static constexpr int kSlots = 3;
    HostSlot     slot[kSlots];
    uint8_t*     d_hc = nullptr;   int hcWaves = 0;     // HC workspace, one slot per resident HC wave (allocated on first use)
    // level 12 in three phases: chain + search results of one group of blocks, the parser's table overflow, an error flag
    uint8_t*     d_h12 = nullptr;  size_t h12Bytes = 0;  int h12ParseWaves = 0;  int h12SegWaves = 0;  int hcLazyWaves = 0;  size_t h12ErrOff = 0;
    // Both HC workspaces belong to one job at a time: the stream of the last HC job and an event recorded behind it; an HC
    // job on another stream waits for that event on the device (no host block).
    hipEvent_t   hcDone = nullptr; hipStream_t hcStream = nullptr; bool hcPending = false;
    hipStream_t  hashStream = nullptr;   // the streaming content checksum runs here, beside the codec kernels
    // the level-1 workspaces of the device-resident calls (the host-buffer calls have one per staging slot): one job at a time
    // each, ordered across streams on the device like the HC workspaces.  Two of them, so that calls on two streams need not
    // wait for each other: the emit kernels of one call then run beside the parse of the next (the second one is only
    // allocated when a second stream shows up while the first workspace is another stream's; PLZ4HIP_L1_WORKSPACES=1: one)
    static constexpr int kL1Shared = 2;
    L1Ws         l1[kL1Shared];
    hipEvent_t   l1Done[kL1Shared] = {nullptr, nullptr}; hipStream_t l1Stream[kL1Shared] = {nullptr, nullptr}; bool l1Pending[kL1Shared] = {false, false};
    // The parse kernel of a staged call fills the device by itself (persistent workgroups around all of a CU's LDS).  Two of them
    // started together on two streams share the CUs half and half and stay in step, so that nothing of one call runs beside the
    // other's parse.  They are therefore ordered on the device: a call's parse on another stream than the last one's waits (behind
    // k_parse_gate) until that one has run its queue dry; it then moves into the CUs as the first one's workgroups leave, and the
    // emit kernels of the first call run beside it.
    int          l1Refused = 0;                // calls left before a refused second workspace is asked for again
    uint32_t*    d_gate = nullptr; uint32_t gateSeq = 0; hipStream_t gateStream = nullptr; bool gatePending = false; int gateBlocks = 0;
    // levels 3..11 on independent blocks: the list builder of the next group of blocks runs on this stream beside the walk of the
    // current one (launch_hc)
    hipStream_t  hcBuildStream = nullptr; hipEvent_t evHcFork = nullptr, evHcHist = nullptr, evHcChain[2] = {nullptr, nullptr}, evHcFree[2] = {nullptr, nullptr};
    // HC levels 3..12 with a dictionary / linked blocks on the list path: the blocks' segment lengths (k_hc_ext_prep); one HC job at a time
    int32_t*     d_hcPfx = nullptr; int hcPfxCap = 0; int hcLazyExWaves = 0;
    // LZ4_decompress_safe of a few blocks by the whole chip (lz4_dx_device.inl): tables of one job at a time, ordered across streams
    uint8_t*     d_dx = nullptr; size_t dxBytes = 0;
    hipEvent_t   dxDone = nullptr; hipStream_t dxStream = nullptr; bool dxPending = false;
    hipStream_t  dxHashStream = nullptr; hipEvent_t evDxFork = nullptr, evDxHash = nullptr;   // records: the block checksums beside the decode
    // plz4hip_dev_compress: the sanitised block lengths of the last call.  One job at a time like the workspaces: a call on another
    // stream waits (on the device) for the event behind the last job's kernels before it overwrites the copy.
    int32_t*     d_lenCopy = nullptr; int lenCopyCap = 0;
    hipEvent_t   lenDone = nullptr; hipStream_t lenStream = nullptr; bool lenPending = false;
};

// == clz4.DictCtx (clz4.go:96-120): a private device copy of the last 64 KiB of the dictionary + the LZ4_loadDictSlow table.
// == xxh32.XXHZero (xxh32zero.go:58-86, :204-235): the stream state in device memory + an event behind its last update
struct plz4hip_xxh32_stream { XxhStream* d_state = nullptr; hipEvent_t done = nullptr; };

struct plz4hip_dict {
    uint8_t*  d_bytes = nullptr;  int len = 0;      // len < 8: the dictionary is dropped by liblz4 (lz4.c:1613-1615)
    uint32_t* d_table = nullptr;
    uint8_t*  d_hc = nullptr;                       // clz4.DictCtxHC: [level-2 tables][hash-chain tables], kHcWorkBytes each
    std::vector<uint8_t> h_bytes;
};

namespace {

constexpr int kQueueSlots = 4096;         // work-queue counters, reused round-robin: far more than launches that can be in flight

// LZ4_loadDict_internal(_ld_slow) on the host (lz4.c:1587-1646): the table a dictionary context carries.
void build_dict_table_slow(const uint8_t* p, int n, uint32_t* tab)
{
    memset(tab, 0, 4096 * sizeof(uint32_t));
    if (n < 8) return;
    auto hash5 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return (uint32_t)(((v << 24) * 889523592379ull) >> 52); };
    const uint32_t cur = 65536, base = cur - (uint32_t)n;
    for (int i = 0; i <= n - 8; i += 3) tab[hash5(p + i)] = base + (uint32_t)i;                 // later entries overwrite
    const uint32_t limit = cur - 65536;
    for (int i = 0; i <= n - 8; i++) { const uint32_t h = hash5(p + i); if (tab[h] <= limit) tab[h] = base + (uint32_t)i; }
}

int fail(plz4hip_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        std::lock_guard<std::mutex> g(c->errMu);
        c->err = buf;
    }
    return code;
}

// Entry points run on the ctx's device and leave the calling thread's current device as they found it.
struct DeviceGuard {
    int prev = -1; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) err = hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};
#define ENTER_DEVICE(ctx) DeviceGuard dg_((ctx)->device); HIPCHK((ctx), dg_.err)

#define HIPCHK(ctx, call)                                                         \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((ctx), PLZ4HIP_E_DEVICE, #call, e_); } while (0)

uint32_t* next_queue(plz4hip_ctx* c, hipStream_t s, hipError_t* e)
{
    uint32_t* q = c->d_queues + c->qslot;
    c->qslot = (c->qslot + 1) % kQueueSlots;
    *e = hipMemsetAsync(q, 0, sizeof(uint32_t), s);
    return q;
}

// Zero a few bytes of device memory and wait for it -- on the ctx's own stream.  (A plain hipMemset would be the process's first
// use of the NULL stream, which then takes one of the four hardware queues: the staging slots of the host-buffer calls end up
// sharing queues and their chunks stop overlapping -- 2560 blocks through host memory 470 -> 790 ms, scripts/host_rate_ab.py.)
inline hipError_t zero_sync(plz4hip_ctx* c, void* p, size_t n)
{
    hipError_t e = hipMemsetAsync(p, 0, n, c->stream);
    return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}

// ... and a small blocking copy the same way (hipMemcpy is a NULL-stream operation as well)
inline hipError_t copy_sync(plz4hip_ctx* c, void* dst, const void* src, size_t n, hipMemcpyKind kind)
{
    hipError_t e = hipMemcpyAsync(dst, src, n, kind, c->stream);
    return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_slot(plz4hip_ctx* c, int i, size_t hostBytes, size_t devBytes)
{
    plz4hip_ctx::HostSlot& sl = c->slot[i];
    if (!sl.s && hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking) != hipSuccess) return fail(c, PLZ4HIP_E_DEVICE, "hipStreamCreate");
    if (hostBytes > sl.hcap) {
        if (sl.h) hipHostFree(sl.h);
        sl.h = nullptr; sl.hcap = 0;
        const size_t want = round_up(hostBytes + (hostBytes >> 3), 1 << 20);
        if (hipHostMalloc((void**)&sl.h, want, hipHostMallocDefault) != hipSuccess) return fail(c, PLZ4HIP_E_NOMEM, "hipHostMalloc");
        sl.hcap = want;
    }
    if (devBytes > sl.dcap) {
        if (sl.d) hipFree(sl.d);
        sl.d = nullptr; sl.dcap = 0;
        const size_t want = round_up(devBytes + (devBytes >> 3), 1 << 20);
        if (hipMalloc((void**)&sl.d, want) != hipSuccess) return fail(c, PLZ4HIP_E_NOMEM, "hipMalloc");
        sl.dcap = want;
    }
    return PLZ4HIP_OK;
}

// memcpy of many buffers on a few host threads (one thread cannot feed PCIe); small jobs stay on the caller's thread
template <class F> void parallel_blocks(int n, size_t totalBytes, F&& fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? (hw < 8 ? hw : 8) : 4);
    if (totalBytes < ((size_t)8 << 20) || n < 2 || nt < 2) { for (int i = 0; i < n; ++i) fn(i); return; }
    if (nt > n) nt = n;
    std::atomic<int> next{0};
    auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

int grid_for(int nBlocks, int resident) { return nBlocks < resident ? nBlocks : resident; }
// workgroups of the level-1 encoder kernels (kEncWavesPerWg waves each) for nBlocks blocks and `resident` resident waves
// Launch of a level-1 encoder kernel: ten-wave workgroups when the call has more blocks than nine-per-CU one-wave workgroups
// could hold at once, one-wave workgroups otherwise.
#define ENC_LAUNCH(kernel, nBlocks, c, s, a) do { \
        const int waves_ = grid_for((nBlocks), (c)->encWaves); \
        if (waves_ > 9 * (c)->cus) hipLaunchKernelGGL(kernel<kEncWavesPerWg>, dim3((waves_ + kEncWavesPerWg - 1) / kEncWavesPerWg), dim3(64 * kEncWavesPerWg), 0, s, a); \
        else hipLaunchKernelGGL(kernel<1>, dim3(waves_), dim3(64), 0, s, a); \
    } while (0)

bool is_hc_level(int level) { return level >= 2 && level <= 12; }        // every row of the level table (lz4hc.c:92-106): mid, hash chain, optimal

int ensure_hc(plz4hip_ctx* c)
{
    if (c->d_hc) return PLZ4HIP_OK;
    // The HC parsers are one dependent memory access after another: the more waves a CU holds, the more of that latency
    // is hidden.  Fill every wave slot the register budget allows (workspace: 320 KiB per wave); PLZ4HIP_HC_WAVES_PER_CU
    // overrides for experiments.
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_encode_rec_hc, 64, 0) != hipSuccess || per < 1) per = 8;
    {   // (level 2's batch walk keeps its tables in these slots as well and holds more waves per CU than the one-thread parsers)
        int perMid = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perMid, k_hc_mid<false>, 64, 0) == hipSuccess && perMid > per) per = perMid;
    }
    if (const char* v = getenv("PLZ4HIP_HC_WAVES_PER_CU")) { const int w = atoi(v); if (w >= 1 && w <= per) per = w; }
    const int waves = c->cus * per;
    if (hipMalloc((void**)&c->d_hc, (size_t)waves * kHcWorkBytes) != hipSuccess) return fail(c, PLZ4HIP_E_NOMEM, "HC workspace");
    c->hcWaves = waves;
    return PLZ4HIP_OK;
}

// ---- HC jobs and their workspaces.  hc_enter: called before an HC job is enqueued on `s` -- if the previous HC job went to a
// different stream, `s` waits for it on the device.  hc_leave: marks the end of the job on `s`.
int hc_enter(plz4hip_ctx* c, hipStream_t s)
{
    if (c->hcPending && c->hcStream != s) HIPCHK(c, hipStreamWaitEvent(s, c->hcDone, 0));
    return PLZ4HIP_OK;
}
int hc_leave(plz4hip_ctx* c, hipStream_t s)
{
    if (!c->hcDone) HIPCHK(c, hipEventCreateWithFlags(&c->hcDone, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->hcDone, s));
    c->hcPending = true; c->hcStream = s;
    return PLZ4HIP_OK;
}

// What an HC call may take for its per-block workspaces (chains, lists, search results) when PLZ4HIP_HC_BUDGET_GIB does not say:
// a quarter of what is free, at most 64 GiB.  A library call that by default walks off with most of the card is not a drop-in
// (round 2 took three quarters); callers that own the GPU raise the budget, and plz4hip_ctx_trim gives the memory back.  More
// blocks per group is faster for these kernels (they live on blocks in flight), so a dedicated machine should set it.
size_t hc_default_budget(size_t freeBytes) { const size_t q = freeBytes / 4, cap = (size_t)64 << 30; return q < cap ? q : cap; }

// Level 12 on independent blocks without dictionary runs in three phases per group of blocks (lz4hc12_device.inl):
// chain -> search results -> parser.  Per block the group workspace holds the chain (2 B per position) and F (8 B per
// position); the group size follows from the memory set aside (PLZ4HIP_HC12_GROUP overrides, for tests).
struct H12Plan { int64_t chainStride, fStride; size_t perBlock; int group; size_t offRank, offList, offOffsets, offF, offWs, offErr, total;
                 int maxChunks; size_t offInfo, offChunkB, offChunkO;       // (the emit stage's small arrays: levels 3..9)
                 int64_t recStride; size_t offRec, offBridge, offMeta, offStarts, offPieces; };   // (segments: levels 3..9)
constexpr int kLzMaxSegs = 512;
int plan_h12(plz4hip_ctx* c, int nBlocks, int maxLen, H12Plan* pl, bool lazy, bool needF)
{
    pl->chainStride = (int64_t)round_up((size_t)maxLen + 1, 1024);
    pl->fStride = (int64_t)round_up((size_t)(maxLen > 11 ? maxLen - 11 : 1), 64);
    if (!needF) pl->fStride = (int64_t)round_up((size_t)maxLen / 4 + 64, 64);        // no search results there: only the block's final records
    pl->maxChunks = (int)((round_up((size_t)maxLen / 4 + 3, 64) + kSeqChunk - 1) / kSeqChunk);
    pl->perBlock = (size_t)pl->chainStride * 2 + (size_t)pl->chainStride * 4 + ((size_t)pl->chainStride + 8) * 4 + (size_t)kHcHashEntries * 4 + (size_t)pl->fStride * 8
                 + sizeof(SeqInfo) + (size_t)pl->maxChunks * 8;
    pl->recStride = (int64_t)round_up((size_t)maxLen / 4 + (size_t)kLzMaxSegs * 80, 64);
    if (lazy) pl->perBlock += (size_t)pl->recStride * 16 + (size_t)kLzMaxSegs * (sizeof(LzSegMeta) + kLzStarts * 8 + 2 * sizeof(LzPiece));
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return fail(c, PLZ4HIP_E_DEVICE, "hipMemGetInfo");
    // hc_default_budget of what is free (and whatever the ctx holds already) unless PLZ4HIP_HC_BUDGET_GIB says more: the parser runs
    // one wave per block, so the more blocks a group has (up to three waves per SIMD: 3072), the better its latency is hidden --
    // 4096 blocks in two groups of 2048 instead of four of 1024: 1418 -> 1664 MiB/s.  plz4hip_ctx_trim gives the memory back.
    size_t budget = hc_default_budget(freeB + c->h12Bytes);
    if (const char* v = getenv("PLZ4HIP_HC_BUDGET_GIB")) { const long g = atol(v); if (g >= 1) budget = (size_t)g << 30; }
    if (budget < c->h12Bytes) budget = c->h12Bytes;
    int64_t grp = (int64_t)(budget / pl->perBlock);
    if (const char* v = getenv("PLZ4HIP_HC12_GROUP")) { const int gv = atoi(v); if (gv >= 1) grp = gv; }
    if (grp < 1) grp = 1;
    if (grp > nBlocks) grp = nBlocks;
    if (!c->h12ParseWaves) {
        int per = 0, perSeg = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_hc12_parse, 64, 0) != hipSuccess || per < 1) per = 8;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perSeg, k_hc12_seg, 64, 0) != hipSuccess || perSeg < 1) perSeg = 8;
        c->h12SegWaves = c->cus * perSeg;
        c->h12ParseWaves = c->cus * (per > perSeg ? per : perSeg);           // (sizes the waves' global price-table slots: the larger of the two grids)
    }
    for (;;) {
        {   // groups of equal size: what is allocated is what one of them needs, whatever the budget would allow
            const int64_t nGroups = (nBlocks + grp - 1) / grp;
            pl->group = (int)((nBlocks + nGroups - 1) / nGroups);
        }
        // [0, 256): the error flag (zeroed when the workspace is allocated, sticky); the chains start behind it
        pl->offRank = 256 + round_up((size_t)pl->group * (size_t)pl->chainStride * 2, 256);
        pl->offList = pl->offRank + round_up((size_t)pl->group * (size_t)pl->chainStride * 4, 256);
        pl->offOffsets = pl->offList + round_up((size_t)pl->group * ((size_t)pl->chainStride + 8) * 4, 256);
        pl->offF = pl->offOffsets + round_up((size_t)pl->group * (size_t)kHcHashEntries * 4, 256);
        pl->offWs = pl->offF + round_up((size_t)pl->group * (size_t)pl->fStride * 8, 256);
        pl->offErr = 0;
        pl->offInfo = pl->offWs + round_up((size_t)c->h12ParseWaves * kHc12WsGlobalBytes, 256);
        pl->offChunkB = pl->offInfo + round_up((size_t)pl->group * sizeof(SeqInfo), 256);
        pl->offChunkO = pl->offChunkB + round_up((size_t)pl->group * (size_t)pl->maxChunks * 4, 256);
        pl->offRec = pl->offChunkO + round_up((size_t)pl->group * (size_t)pl->maxChunks * 4, 256);
        pl->total = pl->offRec;
        if (lazy) {
            pl->offBridge = pl->offRec + round_up((size_t)pl->group * (size_t)pl->recStride * 8, 256);
            pl->offMeta = pl->offBridge + round_up((size_t)pl->group * (size_t)pl->recStride * 8, 256);
            pl->offStarts = pl->offMeta + round_up((size_t)pl->group * kLzMaxSegs * sizeof(LzSegMeta), 256);
            pl->offPieces = pl->offStarts + round_up((size_t)pl->group * kLzMaxSegs * kLzStarts * 8, 256);
            pl->total = pl->offPieces + round_up((size_t)pl->group * kLzMaxSegs * 2 * sizeof(LzPiece), 256);
        }
        if (getenv("PLZ4HIP_VERBOSE"))
            fprintf(stderr, "plz4hip: level 12, %d blocks: free %zu MiB, held %zu MiB, groups of %d (%zu MiB)\n", nBlocks, freeB >> 20, c->h12Bytes >> 20, pl->group, pl->total >> 20);
        if (pl->total <= c->h12Bytes) break;
        if (c->hcPending) HIPCHK(c, hipEventSynchronize(c->hcDone));          // nothing may still use the old workspace
        if (c->d_h12) hipFree(c->d_h12);
        c->d_h12 = nullptr; c->h12Bytes = 0;
        if (hipMalloc((void**)&c->d_h12, pl->total) == hipSuccess) {
            c->h12Bytes = pl->total;
            HIPCHK(c, zero_sync(c, c->d_h12, 256));
            break;
        }
        (void)hipGetLastError();
        c->d_h12 = nullptr;
        if (grp <= 1) return fail(c, PLZ4HIP_E_NOMEM, "level-12 workspace");
        grp = (grp + 1) / 2;                                                  // refused: smaller groups
    }
    return PLZ4HIP_OK;
}

// (blocks of 8 MiB and more -- raw block API only -- keep the one-thread-per-block kernel: the writer packs positions into 23 bits)
bool use_h12(const CodecArgs& a, int maxLen) { return a.level >= 12 && !a.hcEx && maxLen < (1 << 23) && getenv("PLZ4HIP_HC12_OFF") == nullptr; }
// levels 3..11, independent blocks up to 4 MiB: segments walked at once, stitched, record emit (lz4hc_lazy_device.inl)
// (PLZ4HIP_HC12_LAZY=1, an experiment switch: level 12 on the same walk -- its searches made where the optimal parser asks for them,
// wave-wide, instead of every position's search up front, lz4hc12_device.inl)
bool use_lazy(const CodecArgs& a, int maxLen) { return a.level >= 3 && (a.level <= 11 || getenv("PLZ4HIP_HC12_LAZY") != nullptr) && !a.hcEx && maxLen > 0 && maxLen <= kSeqMaxBlock && getenv("PLZ4HIP_HC_LAZY_OFF") == nullptr; }

int launch_l1(plz4hip_ctx* c, hipStream_t s, CodecArgs a, int nb, int maxLen, int rawMode, plz4hip_ctx::L1Ws* ws, bool* midDeclined = nullptr,
              const CodecArgs* rider = nullptr);

// Enqueue one HC call of nb blocks (a: everything but queue / workspace filled in) on s.  rawMode: LZ4 blocks, else records.
int launch_hc_body(plz4hip_ctx* c, hipStream_t s, CodecArgs a, int nb, int maxLen, int rawMode, bool* forked);
int launch_hc(plz4hip_ctx* c, hipStream_t s, CodecArgs a, int nb, int maxLen, int rawMode)
{
    if (int rc = hc_enter(c, s)) return rc;
    bool forked = false;
    const int rc = launch_hc_body(c, s, a, nb, maxLen, rawMode, &forked);
    // Whatever was enqueued uses the ctx's workspaces, also when the body left early: the call's stream joins the builder's
    // stream (a completed body has done so group by group) and the event behind the job is recorded in any case, so that a
    // following trim / destroy / HC job waits for kernels that may still run.
    if (rc != PLZ4HIP_OK && forked && c->hcBuildStream && c->evHcFork
        && hipEventRecord(c->evHcFork, c->hcBuildStream) == hipSuccess) (void)hipStreamWaitEvent(s, c->evHcFork, 0);
    const int rc2 = hc_leave(c, s);
    return rc != PLZ4HIP_OK ? rc : rc2;
}
int launch_hc_body(plz4hip_ctx* c, hipStream_t s, CodecArgs a, int nb, int maxLen, int rawMode, bool* forked)
{
    hipError_t e;
    const auto prep_ext = [&]() -> int {                                    // the blocks' segments laid out, their lengths noted (k_hc_ext_prep)
        if (nb > c->hcPfxCap) {
            if (c->hcPending) HIPCHK(c, hipEventSynchronize(c->hcDone));
            if (c->d_hcPfx) hipFree(c->d_hcPfx);
    if (c->dxPending) hipEventSynchronize(c->dxDone);
    if (c->d_dx) hipFree(c->d_dx);
    if (c->dxDone) hipEventDestroy(c->dxDone);
    if (c->evDxFork) hipEventDestroy(c->evDxFork);
    if (c->evDxHash) hipEventDestroy(c->evDxHash);
            c->d_hcPfx = nullptr; c->hcPfxCap = 0;
            if (hipMalloc((void**)&c->d_hcPfx, (size_t)nb * 4 + 1024) != hipSuccess) return fail(c, PLZ4HIP_E_NOMEM, "HC segment lengths");
            c->hcPfxCap = nb + 256;
        }
        a.hcPfx = c->d_hcPfx; a.rawMode = rawMode; a.nBlocks = nb; a.blk0 = 0;
        a.queue = next_queue(c, s, &e); HIPCHK(c, e);
        hipLaunchKernelGGL(k_hc_ext_prep, dim3(grid_for(nb, c->cus * 8)), dim3(64), 0, s, a);
        return PLZ4HIP_OK;
    };
    const auto ctx_blocks = [&](const CodecArgs& a0) -> int {
        // the blocks <= 4 KiB under a dictionary context (hcPfx < 0; the emit stage has laid them down as stored records in the
        // meantime): the one-thread parser over the context's two sets of tables writes them now
        if (a0.dict == nullptr && a0.dictLen < 0) return PLZ4HIP_OK;
        if (int rc = ensure_hc(c)) return rc;
        CodecArgs x = a0; x.hcWork = c->d_hc; x.blk0 = 0; x.nBlocks = nb; x.h12Chain = nullptr; x.h12Rank = nullptr; x.h12List = nullptr;
        x.queue = next_queue(c, s, &e); HIPCHK(c, e);
        if (rawMode) hipLaunchKernelGGL(k_encode_raw_hc, dim3(grid_for(nb, c->hcWaves)), dim3(64), 0, s, x);
        else         hipLaunchKernelGGL(k_encode_rec_hc, dim3(grid_for(nb, c->hcWaves)), dim3(64), 0, s, x);
        HIPCHK(c, hipGetLastError());
        return PLZ4HIP_OK;
    };
    const bool extOk = a.hcEx && maxLen > 0 && maxLen <= kSeqMaxBlock && getenv("PLZ4HIP_HC_EXT_OFF") == nullptr;
    if (a.level == 2 && (!a.hcEx || extOk) && maxLen > 0 && maxLen <= kSeqMaxBlock && getenv("PLZ4HIP_HC_MID_OFF") == nullptr) {
        // level 2, blocks up to 4 MiB: the staged call of level 1 with the level-2 walk as its parser (behind external segments too)
        if (int rc = ensure_hc(c)) return rc;
        a.hcWork = c->d_hc;
        if (a.hcEx) { if (int rc = prep_ext()) return rc; }
        bool declined = false;
        if (int rc = launch_l1(c, s, a, nb, maxLen, rawMode, nullptr, &declined)) return rc;
        if (!declined) return a.hcEx ? ctx_blocks(a) : PLZ4HIP_OK;
        a.hcPfx = nullptr;                                                   // (no workspace: every block on the one-thread parser)
    }
    // levels 3..12 with a dictionary and/or linked blocks, blocks up to 4 MiB (round 4): the same route as the lazy levels -- chain and
    // lists over segment + block, the walk in segments, stitched, records through the emit stage -- with the segment rules of the
    // reference kept in the finders (lz4hc_lazy_device.inl, kD); level 12 walks its optimal parser there as well.  Only the blocks
    // <= 4 KiB under a dictionary context (usingDictCtxHc: two sets of tables) keep the one-thread parser, launched behind.
    const bool lazyEx = extOk && a.level >= 3;
    const bool lazy = use_lazy(a, maxLen) || lazyEx;
    if (lazy || use_h12(a, maxLen)) {
        // level 12 up to 4 MiB: its parser in segments as well (records; larger raw blocks keep the one-wave parser that writes bytes)
        const bool seg12 = !lazy && maxLen > 0 && maxLen <= kSeqMaxBlock && getenv("PLZ4HIP_HC12_SEG_OFF") == nullptr;
        H12Plan pl;
        if (int rc = plan_h12(c, nb, maxLen + (lazyEx ? 65536 : 0), &pl, lazy || seg12, !lazy)) return rc;
        if (lazyEx) { if (int rc = prep_ext()) return rc; }
        a.h12Chain = (uint16_t*)(c->d_h12 + 256); a.h12ChainStride = pl.chainStride;
        a.h12Rank = (uint32_t*)(c->d_h12 + pl.offRank); a.h12List = (uint32_t*)(c->d_h12 + pl.offList);
        a.h12Offsets = (uint32_t*)(c->d_h12 + pl.offOffsets);
        a.h12F = (Hc12F*)(c->d_h12 + pl.offF); a.h12FStride = pl.fStride;
        a.h12Ws = c->d_h12 + pl.offWs; a.h12Err = (int32_t*)(c->d_h12 + pl.offErr); c->h12ErrOff = pl.offErr;
        a.rawMode = rawMode;
        if (lazy || seg12) {
            if (lazy && a.level >= 10) { if (int rc = ensure_hc(c)) return rc; a.hcWork = c->d_hc; }
            if (!c->hcLazyWaves) {
                int per = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_hc_lazy<false>, 64, 0) != hipSuccess || per < 1) per = 8;
                c->hcLazyWaves = c->cus * per;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_hc_lazy<true>, 64, 0) != hipSuccess || per < 1) per = 8;
                c->hcLazyExWaves = c->cus * per;
            }
            a.l1MaxLen = maxLen; a.l1Seq = (uint64_t*)a.h12F; a.l1SeqStride = pl.fStride; a.l1MaxChunks = pl.maxChunks; a.l1Bk = nullptr;
            a.l1Info = (SeqInfo*)(c->d_h12 + pl.offInfo);
            a.l1ChunkBytes = (uint32_t*)(c->d_h12 + pl.offChunkB); a.l1ChunkOff = (uint32_t*)(c->d_h12 + pl.offChunkO);
            a.lzRec = (uint64_t*)(c->d_h12 + pl.offRec); a.lzBridge = (uint64_t*)(c->d_h12 + pl.offBridge); a.lzRecStride = pl.recStride;
            a.lzMeta = (LzSegMeta*)(c->d_h12 + pl.offMeta); a.lzStarts = (uint64_t*)(c->d_h12 + pl.offStarts); a.lzPieces = (LzPiece*)(c->d_h12 + pl.offPieces);
            a.lzMinSeg = 8192;
            if (const char* v = getenv("PLZ4HIP_HC_MIN_SEG")) { const int m = atoi(v); if (m >= 64) a.lzMinSeg = m; }     // tests: many segments in small blocks
        }
        a.h12Gather = 12; a.h12Idle = 16;
        if (const char* v = getenv("PLZ4HIP_HC12_GATHER")) a.h12Gather = atoi(v);
        if (const char* v = getenv("PLZ4HIP_HC12_IDLE")) a.h12Idle = atoi(v);
        int nGroups = (nb + pl.group - 1) / pl.group;
        int per = (nb + nGroups - 1) / nGroups;                           // groups of equal size (<= pl.group)
        HIPCHK(c, hipMemsetAsync(a.h12Err, 0, 4, s));                      // the give-up flag is per call (k_hc12_parse reports it per block)
        // Levels 3..11: the list builder is one wave per CU around 128 KiB of LDS (0.3 s per 2048 blocks with the chip all but idle),
        // the walk needs no LDS.  A call of enough blocks therefore runs in at least four groups over the two halves of the
        // workspace, the builder of group g+1 on a second stream beside the walk of group g (which leaves it a wave slot per CU);
        // the histogram of group g+1 (1024-thread workgroups that would not find room beside the walk) runs between two walks.
        int overlapMin = 2048;
        if (const char* v = getenv("PLZ4HIP_HC_OVERLAP_MIN")) overlapMin = atoi(v);           // tests: the pipeline on a handful of blocks
        const bool overlap = lazy && nb >= overlapMin && nb >= 2 && pl.group >= 2 && getenv("PLZ4HIP_HC_OVERLAP_OFF") == nullptr;   // (two groups' worth of workspace)
        if (overlap) {
            int want = 4;
            if (const char* v = getenv("PLZ4HIP_HC_OVERLAP_GROUPS")) { want = atoi(v); if (want < 2) want = 2; }
            int tgt = pl.group / 2 < (nb + want - 1) / want ? pl.group / 2 : (nb + want - 1) / want;
            if (tgt < 1) tgt = 1;
            nGroups = (nb + tgt - 1) / tgt;
            per = (nb + nGroups - 1) / nGroups;
            if (!c->hcBuildStream) HIPCHK(c, hipStreamCreateWithFlags(&c->hcBuildStream, hipStreamNonBlocking));
            for (hipEvent_t* ev : {&c->evHcFork, &c->evHcHist, &c->evHcChain[0], &c->evHcChain[1], &c->evHcFree[0], &c->evHcFree[1]})
                if (!*ev) HIPCHK(c, hipEventCreateWithFlags(ev, hipEventDisableTiming));
        }
        // the workspace pointers of the half a group uses: every per-block array moved on by k blocks
        const auto half = [&](CodecArgs x, int k) {
            x.h12Chain += (int64_t)k * x.h12ChainStride; x.h12Rank += (int64_t)k * x.h12ChainStride; x.h12List += (int64_t)k * (x.h12ChainStride + 8);
            x.h12Offsets += (size_t)k * kHcHashEntries; x.h12F += (int64_t)k * x.h12FStride; x.l1Seq += (int64_t)k * x.l1SeqStride;
            x.l1Info += k; x.l1ChunkBytes += (int64_t)k * x.l1MaxChunks; x.l1ChunkOff += (int64_t)k * x.l1MaxChunks;
            x.lzRec += (int64_t)k * x.lzRecStride; x.lzBridge += (int64_t)k * x.lzRecStride;
            x.lzMeta += (int64_t)k * kLzMaxSegs; x.lzStarts += (int64_t)k * kLzMaxSegs * kLzStarts; x.lzPieces += (int64_t)k * 2 * kLzMaxSegs;
            return x;
        };
        const auto build = [&](hipStream_t st, CodecArgs x, bool hist, bool chain) -> int {
            hipError_t e2;
            if (hist)  { x.queue = next_queue(c, st, &e2); HIPCHK(c, e2); hipLaunchKernelGGL(k_hc12_hist, dim3(grid_for(x.nBlocks, c->cus)), dim3(1024), 0, st, x); }
            if (chain) { x.queue = next_queue(c, st, &e2); HIPCHK(c, e2); hipLaunchKernelGGL(k_hc12_chain, dim3(grid_for(x.nBlocks, c->cus)), dim3(64), 0, st, x); }
            return PLZ4HIP_OK;
        };
        const CodecArgs a0 = a;
        // The groups.  With the builder beside the walk only the first group's builder is exposed, so the groups start small and
        // double (a builder takes about half the time of the walk over the same blocks: the next group's is done when this
        // group's walk is) until they have the full size.
        std::vector<int> gStart, gSize;
        for (int at = 0, sz = overlap ? (per / 8 > 0 ? per / 8 : 1) : per; at < nb; ) {
            const int n_ = nb - at < sz ? nb - at : sz;
            gStart.push_back(at); gSize.push_back(n_);
            at += n_;
            sz = sz * 2 < per ? sz * 2 : per;
        }
        nGroups = (int)gStart.size();
        if (overlap) {
            hipStream_t sb = c->hcBuildStream;
            HIPCHK(c, hipEventRecord(c->evHcFork, s));
            HIPCHK(c, hipStreamWaitEvent(sb, c->evHcFork, 0));
            *forked = true;
            CodecArgs x = half(a0, 0); x.blk0 = gStart[0]; x.nBlocks = gSize[0];
            if (int rc = build(sb, x, true, true)) return rc;
            HIPCHK(c, hipEventRecord(c->evHcChain[0], sb));
            if (nGroups > 1) {
                x = half(a0, per); x.blk0 = gStart[1]; x.nBlocks = gSize[1];
                if (int rc = build(sb, x, true, false)) return rc;
                HIPCHK(c, hipEventRecord(c->evHcHist, sb));
            }
        }
        for (int gi = 0; gi < nGroups; ++gi) {
            const int g0 = gStart[gi], ng = gSize[gi];
            if (overlap) a = half(a0, (gi & 1) * per);
            a.blk0 = g0; a.nBlocks = ng;
            if (overlap) {
                hipStream_t sb = c->hcBuildStream;
                HIPCHK(c, hipStreamWaitEvent(s, c->evHcChain[gi & 1], 0));
                if (gi + 1 < nGroups) {
                    HIPCHK(c, hipStreamWaitEvent(s, c->evHcHist, 0));           // the next group's histogram is done: its builder starts now
                    CodecArgs x = half(a0, ((gi + 1) & 1) * per); x.blk0 = gStart[gi + 1]; x.nBlocks = gSize[gi + 1];
                    if (int rc = build(sb, x, false, true)) return rc;
                    HIPCHK(c, hipEventRecord(c->evHcChain[(gi + 1) & 1], sb));
                }
            } else {
                if (int rc = build(s, a, true, true)) return rc;
            }
            a.queue = next_queue(c, s, &e); HIPCHK(c, e);
            if (!lazy) hipLaunchKernelGGL(k_hc12_search, dim3(grid_for(ng, c->cus)), dim3(1024), 0, s, a);
            a.queue = next_queue(c, s, &e); HIPCHK(c, e);
            if (!lazy && !seg12) hipLaunchKernelGGL(k_hc12_parse, dim3(grid_for(ng, c->h12ParseWaves)), dim3(64), 0, s, a);
            else {
                // segments per block: as many as there are 8 KiB pieces, at most 512.  Measured at 4096 blocks of 4 MiB, level 3 /
                // level 9: 16 segments 7522 / 4870, 32: 7734 / 4986, 64: 7908 / 5090, 128: 8071 / 5691, 256: 8279 / 5958, 512: 8602 /
                // 6139, 1024: 8642 / 6159 MiB/s.  Far more walks than the chip holds waves: the queue hands neighbouring segments of
                // one block to neighbouring waves, which then share the block's source and lists in L2, and short items even out.
                a.lzSegs = kLzMaxSegs;
                if (const char* v = getenv("PLZ4HIP_HC_SEGS")) a.lzSegs = atoi(v);
                a.lzSegs = a.lzSegs < 1 ? 1 : (a.lzSegs > kLzMaxSegs ? kLzMaxSegs : a.lzSegs);
                if (seg12) {
                    hipLaunchKernelGGL(k_hc12_seg, dim3(grid_for(ng * a.lzSegs, c->h12SegWaves)), dim3(64), 0, s, a);
                    a.queue = next_queue(c, s, &e); HIPCHK(c, e);
                    hipLaunchKernelGGL(k_hc12_stitch, dim3(grid_for(ng, c->h12SegWaves)), dim3(64), 0, s, a);
                } else {
                const int resident = lazyEx ? c->hcLazyExWaves : c->hcLazyWaves;
                int lzWaves = a.level >= 10 && c->hcWaves < resident ? c->hcWaves : resident;   // (a price table per wave)
                if (overlap && lzWaves > 2 * c->cus) lzWaves -= c->cus;                    // a wave slot (and its registers) per CU for the builder
                if (lazyEx) hipLaunchKernelGGL(k_hc_lazy<true>, dim3(grid_for(ng * a.lzSegs, lzWaves)), dim3(64), 0, s, a);
                else        hipLaunchKernelGGL(k_hc_lazy<false>, dim3(grid_for(ng * a.lzSegs, lzWaves)), dim3(64), 0, s, a);
                a.queue = next_queue(c, s, &e); HIPCHK(c, e);
                if (lazyEx) hipLaunchKernelGGL(k_hc_stitch<true>, dim3(grid_for(ng, lzWaves)), dim3(64), 0, s, a);
                else        hipLaunchKernelGGL(k_hc_stitch<false>, dim3(grid_for(ng, lzWaves)), dim3(64), 0, s, a);
                }
                hipLaunchKernelGGL(k_hc_gather, dim3(ng >= 1024 ? 4 : 16, ng), dim3(256), 0, s, a);
                int wg = (16384 / ng) / 4;                                  // emit: waves per block so that a small call still spreads over the chip
                if (wg > (pl.maxChunks + 3) / 4) wg = (pl.maxChunks + 3) / 4;
                if (wg < 1) wg = 1;
                hipLaunchKernelGGL(k_l1_sizes<false>, dim3(wg, ng), dim3(256), 0, s, a);
                hipLaunchKernelGGL(k_l1_scan, dim3((ng + 3) / 4), dim3(256), 0, s, a);
                hipLaunchKernelGGL(k_l1_write<false>, dim3(wg, ng), dim3(256), 0, s, a);
                if (!rawMode && a.blockChecksum) hipLaunchKernelGGL(k_l1_finish, dim3((ng + 3) / 4), dim3(256), 0, s, a);
            }
            HIPCHK(c, hipGetLastError());
            if (overlap && gi + 2 < nGroups) {
                // this half is free again: the histogram of the group after next may overwrite it
                hipStream_t sb = c->hcBuildStream;
                HIPCHK(c, hipEventRecord(c->evHcFree[gi & 1], s));
                HIPCHK(c, hipStreamWaitEvent(sb, c->evHcFree[gi & 1], 0));
                CodecArgs x = half(a0, (gi & 1) * per); x.blk0 = gStart[gi + 2]; x.nBlocks = gSize[gi + 2];
                if (int rc = build(sb, x, true, false)) return rc;
                HIPCHK(c, hipEventRecord(c->evHcHist, sb));
            }
        }
        if (lazyEx) { if (int rc = ctx_blocks(a0)) return rc; }
    } else {
        if (int rc = ensure_hc(c)) return rc;
        a.hcWork = c->d_hc; a.nBlocks = nb;
        a.h12Chain = nullptr;
        a.blk0 = 0;
        int per = nb;                     // blocks per launch
        bool lists = false;
        size_t chainBytes = 0, rankBytes = 0, listBytes = 0;
        if (a.level >= 3 && !a.hcEx && maxLen >= 4096 && getenv("PLZ4HIP_HC_PRE_OFF") == nullptr) {
            // levels 3..12, independent blocks: the chain of every block built up front (2 B per position): the parsers then run
            // without their 4 M dependent table updates per 4 MiB block.  Levels 4..12 (8 candidates and more per search)
            // also get the per-hash lists (another 8 B per position): their searches then look at up to 63 candidates per
            // round (hc_find_wider_lists).  A call whose blocks do not fit the memory set aside runs in groups of equal size.
            const int64_t stride = (int64_t)round_up((size_t)maxLen + 1, 1024);
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return fail(c, PLZ4HIP_E_DEVICE, "hipMemGetInfo");
            size_t budget = hc_default_budget(freeB + c->h12Bytes);      // the chain alone
            // the lists are worth more memory than that: these kernels live on the number of blocks in flight (7 waves per SIMD
            // fit), and 4096 blocks of 4 MiB with their lists (40 MiB each) are 160 GiB: a machine that is there for this sets PLZ4HIP_HC_BUDGET_GIB;
            // plz4hip_ctx_trim gives it back.  What the ctx already holds is used in any case.
            size_t budgetLists = hc_default_budget(freeB + c->h12Bytes);
            if (const char* v = getenv("PLZ4HIP_HC_BUDGET_GIB")) { const long g = atol(v); if (g >= 1) budget = budgetLists = (size_t)g << 30; }
            if (budget < c->h12Bytes) budget = c->h12Bytes;
            if (budgetLists < c->h12Bytes) budgetLists = c->h12Bytes;
            const size_t slack = 256 + 4 * 256;
            const size_t chainPer = (size_t)stride * 2;
            const size_t listsPer = chainPer + (size_t)stride * 4 + ((size_t)stride + 8) * 4 + (size_t)kHcHashEntries * 4;
            const auto groups_of = [&](size_t room, size_t perBlock) {          // blocks per launch, groups of equal size
                int64_t g = room > slack ? (int64_t)((room - slack) / perBlock) : 0;
                if (const char* v = getenv("PLZ4HIP_HC_GROUP")) { const int gv = atoi(v); if (gv >= 1 && gv < g) g = gv; }
                if (g < 1) return 0;
                if (g > nb) g = nb;
                const int n = (int)((nb + g - 1) / g);
                return (nb + n - 1) / n;
            };
            const int perLists = (a.level >= 4 && getenv("PLZ4HIP_HC_LISTS_OFF") == nullptr) ? groups_of(budgetLists, listsPer) : 0;
            const int perChain = groups_of(budget, chainPer);
            // a group has to keep the chip busy: below 2048 blocks in flight the lists lose to the chain alone over more blocks
            // at levels 4..9; the optimal parser's levels gain an order of magnitude and take them in any case
            lists = perLists >= 1 && (perLists == nb || perLists >= (a.level >= 10 ? 256 : 2048) || perChain < 1);
            if (lists && perLists > c->h12Bytes / listsPer) {             // try the large request first: refused -> the chain alone
                if (c->hcPending) HIPCHK(c, hipEventSynchronize(c->hcDone));
                if (c->d_h12) hipFree(c->d_h12);
                c->d_h12 = nullptr; c->h12Bytes = 0;
                const size_t need = slack + (size_t)perLists * listsPer;
                if (hipMalloc((void**)&c->d_h12, need) != hipSuccess) { (void)hipGetLastError(); c->d_h12 = nullptr; lists = false; }
                else { c->h12Bytes = need; HIPCHK(c, zero_sync(c, c->d_h12, 256)); }
            }
            const int perPre = lists ? perLists : perChain;
            if (getenv("PLZ4HIP_VERBOSE"))
                fprintf(stderr, "plz4hip: HC level %d, %d blocks: free %zu MiB, held %zu MiB, lists %d, %d blocks per group\n", a.level, nb,
                        freeB >> 20, c->h12Bytes >> 20, (int)lists, perPre);
            if (perPre >= 1) {
                per = perPre;
                const size_t need = slack + (size_t)per * (lists ? listsPer : chainPer);
                if (need > c->h12Bytes) {
                    if (c->hcPending) HIPCHK(c, hipEventSynchronize(c->hcDone));
                    if (c->d_h12) hipFree(c->d_h12);
                    c->d_h12 = nullptr; c->h12Bytes = 0;
                    if (hipMalloc((void**)&c->d_h12, need) != hipSuccess) { c->d_h12 = nullptr; return fail(c, PLZ4HIP_E_NOMEM, "HC chain workspace"); }
                    c->h12Bytes = need;
                    HIPCHK(c, zero_sync(c, c->d_h12, 256));
                }
                chainBytes = round_up((size_t)per * chainPer, 256);
                rankBytes = round_up((size_t)per * (size_t)stride * 4, 256);
                listBytes = round_up((size_t)per * ((size_t)stride + 8) * 4, 256);
                a.h12Chain = (uint16_t*)(c->d_h12 + 256); a.h12ChainStride = stride;
                a.h12Rank = nullptr; a.h12List = nullptr; a.h12Offsets = nullptr;
                if (lists) {
                    a.h12Rank = (uint32_t*)(c->d_h12 + 256 + chainBytes);
                    a.h12List = (uint32_t*)(c->d_h12 + 256 + chainBytes + rankBytes);
                    a.h12Offsets = (uint32_t*)(c->d_h12 + 256 + chainBytes + rankBytes + listBytes);
                }
            }
        }
        for (int g0 = 0; g0 < nb; g0 += per) {
            const int ng = nb - g0 < per ? nb - g0 : per;
            a.blk0 = g0; a.nBlocks = ng;
            if (a.h12Chain && lists) {
                a.queue = next_queue(c, s, &e); HIPCHK(c, e);
                hipLaunchKernelGGL(k_hc12_hist, dim3(grid_for(ng, c->cus)), dim3(1024), 0, s, a);
                a.queue = next_queue(c, s, &e); HIPCHK(c, e);
                hipLaunchKernelGGL(k_hc12_chain, dim3(grid_for(ng, c->cus)), dim3(64), 0, s, a);
            } else if (a.h12Chain) {
                a.queue = next_queue(c, s, &e); HIPCHK(c, e);
                hipLaunchKernelGGL(k_hc_chain, dim3(grid_for(ng, c->cus)), dim3(64), 0, s, a);
            }
            a.queue = next_queue(c, s, &e); HIPCHK(c, e);
            if (rawMode) hipLaunchKernelGGL(k_encode_raw_hc, dim3(grid_for(ng, c->hcWaves)), dim3(64), 0, s, a);
            else         hipLaunchKernelGGL(k_encode_rec_hc, dim3(grid_for(ng, c->hcWaves)), dim3(64), 0, s, a);
            HIPCHK(c, hipGetLastError());
        }
    }
    return PLZ4HIP_OK;
}

// Enqueue one level-1 call of nb independent blocks without dictionary (a: everything but queue / workspace filled in) on s.
// rawMode: LZ4 blocks (result = bytes or 0), else records.  Blocks up to 4 MiB run in stages (lz4_seq_device.inl): the
// workspace holds 8 bytes per possible sequence -- 2 bytes per input byte -- of one group of blocks; a call that does not fit
// the memory set aside (half of what is free; PLZ4HIP_L1_BUDGET_GIB) runs in groups of equal size.  ws: the workspace to use
// (a staging slot's own, always used from that slot's stream) or null for the ctx's, which is ordered across streams.
// Larger blocks (raw block API only) and calls that do not say how long their blocks are keep the fused kernels.
// midDeclined: the call is a level-2 call (k_hc_mid instead of k_l1_parse, records without catch-up); set when no workspace could
// be had, the caller then runs its one-kernel path.
// rider: the record decode of another call (queue filled in) that shares the parse kernel's launch (k_l1_duplex); a call that
// runs in groups carries it in its first group, one that takes the fused kernels launches it by itself.
int launch_l1(plz4hip_ctx* c, hipStream_t s, CodecArgs a, int nb, int maxLen, int rawMode, plz4hip_ctx::L1Ws* ws, bool* midDeclined,
              const CodecArgs* rider)
{
    hipError_t e;
    const bool mid = midDeclined != nullptr;
    a.rawMode = rawMode; a.blk0 = 0; a.nBlocks = nb;
    const bool shared = (ws == nullptr);
    int wsi = 0;
    if (shared) {
        // the workspace this stream used last; else one whose last job is over (no second allocation for streams that merely take
        // turns); else one that does not exist yet; else the first one (and the wait below)
        int nws = plz4hip_ctx::kL1Shared;
        if (const char* v = getenv("PLZ4HIP_L1_WORKSPACES")) { if (atoi(v) == 1) nws = 1; }
        int mine = -1, idle = -1, fresh = -1;
        for (int i = 0; i < nws; ++i) {
            if (c->l1[i].d && c->l1Stream[i] == s && mine < 0) mine = i;
            if (c->l1[i].d && idle < 0 && (!c->l1Pending[i] || hipEventQuery(c->l1Done[i]) == hipSuccess)) idle = i;   // (its last job is over)
            if (!c->l1[i].d && fresh < 0) fresh = i;
        }
        (void)hipGetLastError();                                                          // (hipErrorNotReady of the query)
        wsi = mine >= 0 ? mine : (idle >= 0 ? idle : (fresh >= 0 ? fresh : 0));
        ws = &c->l1[wsi];
    }
    bool fused = maxLen <= 0 || maxLen > kSeqMaxBlock || getenv("PLZ4HIP_L1_FUSED") != nullptr;
    const size_t seqStride = round_up((size_t)(maxLen > 0 ? maxLen : 0) / 4 + 3, 64);     // + the dump entry (lz4_seq_device.inl)
    const int    maxChunks = (int)((seqStride + kSeqChunk - 1) / kSeqChunk);
    const size_t perBlock  = sizeof(SeqInfo) + (size_t)maxChunks * 8 + seqStride * 9;
    const auto need_for = [&](int per) { return round_up((size_t)per * sizeof(SeqInfo), 256) + 2 * round_up((size_t)per * maxChunks * 4, 256) + (size_t)per * seqStride * 9; };
    int per = nb;
    if (shared && !fused && wsi != 0 && !ws->d && c->l1Refused > 0) { c->l1Refused--; wsi = 0; ws = &c->l1[0]; }   // (refused a moment ago)
    if (shared && !fused && wsi != 0 && !ws->d) {
        // a SECOND workspace is there for overlap: it is taken whole or not at all (a smaller one would cut this call into groups,
        // which costs more than waiting for the first workspace does)
        if (hipMalloc((void**)&ws->d, need_for(nb)) == hipSuccess) ws->bytes = need_for(nb);
        else {
            (void)hipGetLastError();
            ws->d = nullptr; ws->bytes = 0; wsi = 0; ws = &c->l1[0];
            c->l1Refused = 8;                                                    // (the next calls do not ask again)
            if (getenv("PLZ4HIP_VERBOSE")) fprintf(stderr, "plz4hip: level 1: no room for a second workspace of %zu MiB, sharing the first\n", need_for(nb) >> 20);
        }
    }
    if (!fused && need_for(nb) > ws->bytes) {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return fail(c, PLZ4HIP_E_DEVICE, "hipMemGetInfo");
        size_t budget = (freeB + ws->bytes) / 2;
        if (const char* v = getenv("PLZ4HIP_L1_BUDGET_GIB")) { const long g = atol(v); if (g >= 1) budget = (size_t)g << 30; }
        if (const char* v = getenv("PLZ4HIP_L1_BUDGET_MIB")) { const long g = atol(v); if (g >= 1) budget = (size_t)g << 20; }   // tests: force groups
        if (budget < ws->bytes) budget = ws->bytes;
        int64_t grp = (int64_t)(budget / (perBlock + 64));
        if (grp > nb) grp = nb;
        while (grp >= 1) {
            const int nGroups = (int)((nb + grp - 1) / grp);
            per = (nb + nGroups - 1) / nGroups;
            if (need_for(per) <= ws->bytes) break;
            if (shared && c->l1Pending[wsi]) HIPCHK(c, hipEventSynchronize(c->l1Done[wsi]));     // nothing may still use the old workspace
            else if (!shared) HIPCHK(c, hipStreamSynchronize(s));
            if (ws->d) hipFree(ws->d);
            ws->d = nullptr; ws->bytes = 0;
            if (hipMalloc((void**)&ws->d, need_for(per)) == hipSuccess) { ws->bytes = need_for(per); break; }
            (void)hipGetLastError();
            ws->d = nullptr;
            grp = grp / 2;                                                              // refused: smaller groups
        }
        if (grp < 1) fused = true;                                                      // not even one block: the fused kernels need no workspace
        if (getenv("PLZ4HIP_VERBOSE"))
            fprintf(stderr, "plz4hip: level 1, %d blocks of <= %d: free %zu MiB, workspace %zu MiB, groups of %d%s\n", nb, maxLen, freeB >> 20, ws->bytes >> 20, per, fused ? " (fused)" : "");
    }
    if (fused && mid) { *midDeclined = true; return PLZ4HIP_OK; }
    if (fused && a.bodyOff) return fail(c, PLZ4HIP_E_NOMEM, "records straight into a frame body need the staged call's workspace (blocks up to 4 MiB)");
    if (fused) {
        a.queue = next_queue(c, s, &e); HIPCHK(c, e);
        if (rawMode) ENC_LAUNCH(k_encode_raw, nb, c, s, a); else ENC_LAUNCH(k_encode_rec, nb, c, s, a);
        if (rider) hipLaunchKernelGGL(k_decode_rec, dim3(grid_for(rider->nBlocks, c->decWaves)), dim3(64), 0, s, *rider);
        HIPCHK(c, hipGetLastError());
        return PLZ4HIP_OK;
    }
    if (shared && c->l1Pending[wsi] && c->l1Stream[wsi] != s) HIPCHK(c, hipStreamWaitEvent(s, c->l1Done[wsi], 0));
    a.l1MaxLen = maxLen; a.l1SeqStride = (int64_t)seqStride; a.l1MaxChunks = maxChunks;
    a.l1Info = (SeqInfo*)ws->d;
    a.l1ChunkBytes = (uint32_t*)(ws->d + round_up((size_t)per * sizeof(SeqInfo), 256));
    a.l1ChunkOff   = (uint32_t*)((uint8_t*)a.l1ChunkBytes + round_up((size_t)per * maxChunks * 4, 256));
    a.l1Seq        = (uint64_t*)((uint8_t*)a.l1ChunkOff + round_up((size_t)per * maxChunks * 4, 256));
    a.l1Bk         = (uint8_t*)(a.l1Seq + (size_t)per * seqStride);
    for (int g0 = 0; g0 < nb; g0 += per) {
        const int ng = nb - g0 < per ? nb - g0 : per;
        a.blk0 = g0; a.nBlocks = ng;
        a.queue = next_queue(c, s, &e); HIPCHK(c, e);
        const bool signals = !mid && !getenv("PLZ4HIP_EXP_NO_GATE");            // (the level-1 parse and the duplex launch)
        if (signals) {
            // (only behind a launch that fills the device: the parse launches of the host-buffer calls' chunks -- a third of the
            // wave slots each -- are meant to share it, and behind the gate they would run one after the other: 2560 blocks through
            // host memory 470 -> 680 ms)
            if (c->gatePending && c->gateStream != s && c->gateBlocks >= 8 * c->cus)
                hipLaunchKernelGGL(k_parse_gate, dim3(1), dim3(64), 0, s, (const uint32_t*)c->d_gate, c->gateSeq);
            if (c->gateSeq >= 0x7FFF0000u) {                                    // (once in two billion launches: start over)
                HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, zero_sync(c, c->d_gate, sizeof(uint32_t))); c->gateSeq = 0;
            }
            a.gate = c->d_gate; a.gateSeq = ++c->gateSeq;
            c->gatePending = true; c->gateStream = s; c->gateBlocks = ng;
        } else {
            a.gate = nullptr; c->gatePending = false;
        }
        if (mid && a.hcPfx) hipLaunchKernelGGL(k_hc_mid<true>, dim3(grid_for(ng, c->hcWaves)), dim3(64), 0, s, a);
        else if (mid) hipLaunchKernelGGL(k_hc_mid<false>, dim3(grid_for(ng, c->hcWaves)), dim3(64), 0, s, a);
        else if (rider && g0 == 0) {
            // workgroups: what the device holds at once, unless neither role has that much to do
            int P = 1, D = 1, prio = 3;
            if (const char* ev = getenv("PLZ4HIP_DUPLEX")) { if (sscanf(ev, "%d,%d", &P, &D) != 2) { P = 1; D = 1; } }
            if (const char* ev = getenv("PLZ4HIP_DUPLEX_PRIO")) prio = atoi(ev);
#define DUPLEX_CASE(PP, DD) \
            if (P == PP && D == DD) { \
                int wgs = ((ng + PP - 1) / PP > (rider->nBlocks + DD - 1) / DD) ? (ng + PP - 1) / PP : (rider->nBlocks + DD - 1) / DD; \
                const int hold = c->cus * duplex_wgs_per_cu(PP, DD); \
                if (wgs > hold) wgs = hold; \
                if (prio) hipLaunchKernelGGL((k_l1_duplex<PP, DD, 3>), dim3(wgs), dim3(64 * (PP + DD)), 0, s, a, *rider); \
                else      hipLaunchKernelGGL((k_l1_duplex<PP, DD, 0>), dim3(wgs), dim3(64 * (PP + DD)), 0, s, a, *rider); \
                launched = true; \
            }
            bool launched = false;
            DUPLEX_CASE(1, 1) DUPLEX_CASE(1, 2) DUPLEX_CASE(3, 4) DUPLEX_CASE(9, 7)      // (9+5, 9+3, 3+2 measured as well: DESIGN 3.1b)
#undef DUPLEX_CASE
            if (!launched) return fail(c, PLZ4HIP_E_ARG, "PLZ4HIP_DUPLEX: not a built combination");
        }
#if defined(PLZ4_EXP_TABBITS)
        else {   // EXPERIMENT: W waves per workgroup, as many workgroups as the device holds
            int W = 10; if (const char* ev = getenv("PLZ4HIP_EXP_W")) W = atoi(ev);
#define EXP_CASE(WW) if (W == WW) { int per = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_l1_parse<WW>, 64 * WW, 0); \
                if (getenv("PLZ4HIP_VERBOSE")) fprintf(stderr, "exp: W=%d workgroups per CU %d\n", WW, per); \
                int wgs = c->cus * per; if (wgs * WW > ng) wgs = (ng + WW - 1) / WW; \
                hipLaunchKernelGGL(k_l1_parse<WW>, dim3(wgs), dim3(64 * WW), 0, s, a); }
            EXP_CASE(4) EXP_CASE(5) EXP_CASE(6) EXP_CASE(8) EXP_CASE(9) EXP_CASE(10) EXP_CASE(12) EXP_CASE(13) EXP_CASE(14) EXP_CASE(16)
#undef EXP_CASE
        }
#else
        else ENC_LAUNCH(k_l1_parse, ng, c, s, a);
#endif
        // emit: waves per block so that a small call still spreads over the chip
        int wg = (16384 / ng) / 4;
        if (wg > (maxChunks + 3) / 4) wg = (maxChunks + 3) / 4;
        if (wg < 1) wg = 1;
        if (mid) hipLaunchKernelGGL(k_l1_sizes<false>, dim3(wg, ng), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_l1_sizes<true>, dim3(wg, ng), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_l1_scan, dim3((ng + 3) / 4), dim3(256), 0, s, a);
        if (a.bodyOff) hipLaunchKernelGGL(k_scan_from, dim3(1), dim3(1024), 0, s, (const int32_t*)(a.result + g0), a.bodyOff + g0, ng, g0 == 0 ? 1 : 0);
        if (mid) hipLaunchKernelGGL(k_l1_write<false>, dim3(wg, ng), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_l1_write<true>, dim3(wg, ng), dim3(256), 0, s, a);
        if (!rawMode && a.blockChecksum) hipLaunchKernelGGL(k_l1_finish, dim3((ng + 3) / 4), dim3(256), 0, s, a);
        HIPCHK(c, hipGetLastError());
    }
    if (shared) {
        if (!c->l1Done[wsi]) HIPCHK(c, hipEventCreateWithFlags(&c->l1Done[wsi], hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->l1Done[wsi], s));
        c->l1Pending[wsi] = true; c->l1Stream[wsi] = s;
    }
    return PLZ4HIP_OK;
}

// Enqueue the decode of nb blocks on s (a: everything but the queue filled in) -- raw blocks (LZ4_decompress_safe per block) or, with
// `records`, frame records (FrameReader._read + BlkT.Decompress, blk/frame.go:54-127, blk.go:50-61).  maxIn / maxOut: upper bounds of
// the blocks' compressed and decoded sizes known to the host (the lengths themselves may live on the device).
// A call of few blocks is cut across the chip (lz4_dx_device.inl: 1 x 4 MiB in about a millisecond of kernels instead of the
// 66 ms one wavefront needs; the records' block checksums are verified beside it on a stream of their own); blocks that path will
// not answer for -- anything but a plainly valid compressed block -- and calls of many blocks, where one wave per block is the
// better use of the chip, run the one-wave kernels.  PLZ4HIP_DX_MAX_BLOCKS (default 128; 0: off).
int launch_decode(plz4hip_ctx* c, hipStream_t s, CodecArgs a, int nb, int64_t maxIn, int64_t maxOut, bool records)
{
    hipError_t e;
    int dxMax = 128;
    if (const char* v = getenv("PLZ4HIP_DX_MAX_BLOCKS")) dxMax = atoi(v);
    bool dx = nb <= dxMax && maxIn >= 16384 && maxIn <= (int64_t)(6 << 20) && maxOut >= 1;
    a.dxInfo = nullptr; a.dxSrcOff = nullptr; a.dxLen = nullptr; a.dxHashBad = nullptr;
    if (dx) {
        const int64_t outB = maxOut < kDxMaxOut ? maxOut : kDxMaxOut;
        const size_t tStride = round_up((size_t)maxIn + 64, 64), pStride = round_up((size_t)outB + 64, 1024);
        const int maxSeg = (int)((maxIn + kDxSeg - 1) / kDxSeg);
        const size_t offPtr = round_up((size_t)nb * tStride * 8, 256), offUnits = offPtr + round_up((size_t)nb * pStride * 4, 256);
        const size_t offInfo = offUnits + round_up((size_t)nb * maxSeg * sizeof(DxUnit), 256), offRec = offInfo + round_up((size_t)nb * sizeof(DxInfo), 256);
        const size_t need = offRec + round_up((size_t)nb * 16, 256);
        if (c->dxPending && c->dxStream != s) HIPCHK(c, hipStreamWaitEvent(s, c->dxDone, 0));
        if (need > c->dxBytes) {
            if (c->dxPending) HIPCHK(c, hipEventSynchronize(c->dxDone));
            if (c->d_dx) hipFree(c->d_dx);
            c->d_dx = nullptr; c->dxBytes = 0;
            if (hipMalloc((void**)&c->d_dx, need) != hipSuccess) { (void)hipGetLastError(); c->d_dx = nullptr; dx = false; }   // no room: one wave per block
            else c->dxBytes = need;
        }
        if (dx) {
            a.dxT = (uint64_t*)c->d_dx; a.dxTStride = (int64_t)tStride;
            a.dxPtr = (uint32_t*)(c->d_dx + offPtr); a.dxPtrStride = (int64_t)pStride;
            a.dxUnits = (DxUnit*)(c->d_dx + offUnits); a.dxMaxSeg = maxSeg;
            a.dxInfo = (DxInfo*)(c->d_dx + offInfo);
            const bool hashed = records && a.blockChecksum;
            if (records) {
                int64_t* so = (int64_t*)(c->d_dx + offRec); int32_t* ln = (int32_t*)(so + nb);
                hipLaunchKernelGGL(k_dx_rec_prep, dim3((nb + 255) / 256), dim3(256), 0, s, a, so, ln);
                a.dxSrcOff = so; a.dxLen = ln;
                if (hashed) {
                    a.dxHashBad = ln + nb;
                    // (the ctx's own stream, idle otherwise: one stream more would push the staging slots' streams onto shared
                    // hardware queues -- the runtime has four -- and cost the large host calls their overlap: 412 -> 610 ms per
                    // 2304 blocks, measured)
                    c->dxHashStream = c->stream;
                    if (!c->evDxFork) HIPCHK(c, hipEventCreateWithFlags(&c->evDxFork, hipEventDisableTiming));
                    if (!c->evDxHash) HIPCHK(c, hipEventCreateWithFlags(&c->evDxHash, hipEventDisableTiming));
                    HIPCHK(c, hipEventRecord(c->evDxFork, s));
                    HIPCHK(c, hipStreamWaitEvent(c->dxHashStream, c->evDxFork, 0));
                    hipLaunchKernelGGL(k_dx_rec_hash, dim3(nb), dim3(64), 0, c->dxHashStream, a);
                    HIPCHK(c, hipEventRecord(c->evDxHash, c->dxHashStream));
                }
            }
            const int chunks = (int)((outB + 1023) / 1024);
            hipLaunchKernelGGL(k_dx_tables, dim3(maxSeg, nb), dim3(64), 0, s, a);
            hipLaunchKernelGGL(k_dx_stitch, dim3(nb), dim3(64), 0, s, a);
            hipLaunchKernelGGL(k_dx_fill, dim3(maxSeg, nb), dim3(64), 0, s, a);
            for (int r = 0; r < kDxRounds; ++r) { a.dxRound = r; hipLaunchKernelGGL(k_dx_jump, dim3(chunks, nb), dim3(256), 0, s, a); }
            if (hashed) HIPCHK(c, hipStreamWaitEvent(s, c->evDxHash, 0));
            hipLaunchKernelGGL(k_dx_gather, dim3(chunks, nb), dim3(256), 0, s, a);
            HIPCHK(c, hipGetLastError());
        }
    }
    a.queue = next_queue(c, s, &e); HIPCHK(c, e);
    if (records && dx) hipLaunchKernelGGL(k_decode_rec_dx, dim3(grid_for(nb, c->decWaves)), dim3(64), 0, s, a);
    else if (records)  hipLaunchKernelGGL(k_decode_rec, dim3(grid_for(nb, c->decWaves)), dim3(64), 0, s, a);
    else               hipLaunchKernelGGL(k_decode_raw, dim3(grid_for(nb, c->decWaves)), dim3(64), 0, s, a);
    HIPCHK(c, hipGetLastError());
    if (dx) {
        if (!c->dxDone) HIPCHK(c, hipEventCreateWithFlags(&c->dxDone, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->dxDone, s));
        c->dxPending = true; c->dxStream = s;
    }
    return PLZ4HIP_OK;
}

}  // namespace

extern "C" {

int plz4hip_abi_version(void) { return PLZ4HIP_ABI_VERSION; }

#if defined(PLZ4_STATS)
// diagnostics build only: read-and-reset the device counters
int plz4hip_debug_stats(unsigned long long* out24)
{
    unsigned long long zero[24] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return PLZ4HIP_E_DEVICE;
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(plz4_stats), sizeof zero) != hipSuccess) return PLZ4HIP_E_DEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(plz4_stats), zero, sizeof zero) != hipSuccess) return PLZ4HIP_E_DEVICE;
    return PLZ4HIP_OK;
}
#endif

int plz4hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return PLZ4HIP_E_DEVICE;
    return n;
}

int plz4hip_compress_bound(int n) { return ((unsigned)n > 0x7E000000u) ? 0 : n + n / 255 + 16; }

int64_t plz4hip_dev_stage_stride(int bsz) { return (int64_t)round_up((size_t)bsz + 8, 16); }

int plz4hip_ctx_create(int device, plz4hip_ctx** out)
{
    if (!out) return PLZ4HIP_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return PLZ4HIP_E_DEVICE;
    plz4hip_ctx* c = new (std::nothrow) plz4hip_ctx();
    if (!c) return PLZ4HIP_E_NOMEM;
    c->device = device;
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_queues, kQueueSlots * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_gate, 256);
    // (on the ctx's own stream: a hipMemset would be this process's first use of the NULL stream, which then takes one of the
    // four hardware queues -- the staging slots of the host-buffer calls end up sharing queues and their chunks stop overlapping:
    // 2560 blocks 470 -> 790 ms, found with scripts/host_rate_ab.py)
    if (e == hipSuccess) e = zero_sync(c, c->d_gate, 256);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device);
    int encPer = 0, decPer = 0;
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&encPer, k_encode_rec<kEncWavesPerWg>, 64 * kEncWavesPerWg, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&decPer, k_decode_rec, 64, 0);
    if (e != hipSuccess) {
        fprintf(stderr, "plz4hip_ctx_create: %s\n", hipGetErrorString(e));
        if (c->d_queues) hipFree(c->d_queues);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return PLZ4HIP_E_DEVICE;
    }
    if (encPer < 1) encPer = 1;
    if (decPer < 1) decPer = 1;
    if (encPer > 1) encPer = 1;                      // one 160 KiB workgroup is all of a CU's LDS
    c->encWaves = c->cus * encPer * kEncWavesPerWg;
    c->decWaves = c->cus * decPer;
    *out = c;
    return PLZ4HIP_OK;
}

void plz4hip_ctx_destroy(plz4hip_ctx* c)
{
    if (!c) return;
    DeviceGuard dg(c->device);
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->d_queues) hipFree(c->d_queues);
    for (auto& sl : c->slot) {
        if (sl.s) { hipStreamSynchronize(sl.s); hipStreamDestroy(sl.s); }
        if (sl.evData) hipEventDestroy(sl.evData);
        if (sl.evHash) hipEventDestroy(sl.evHash);
        if (sl.h) hipHostFree(sl.h);
        if (sl.d) hipFree(sl.d);
        if (sl.l1.d) hipFree(sl.l1.d);
    }
    if (c->d_hc) hipFree(c->d_hc);
    if (c->d_h12) hipFree(c->d_h12);
    for (int i = 0; i < plz4hip_ctx::kL1Shared; ++i) {
        if (c->l1Pending[i]) hipEventSynchronize(c->l1Done[i]);
        if (c->l1[i].d) hipFree(c->l1[i].d);
        if (c->l1Done[i]) hipEventDestroy(c->l1Done[i]);
    }
    if (c->d_lenCopy) hipFree(c->d_lenCopy);
    if (c->d_hcPfx) hipFree(c->d_hcPfx);
    if (c->lenDone) hipEventDestroy(c->lenDone);
    if (c->d_gate) hipFree(c->d_gate);
    if (c->hcDone) hipEventDestroy(c->hcDone);
    if (c->hashStream) { hipStreamSynchronize(c->hashStream); hipStreamDestroy(c->hashStream); }
    if (c->hcBuildStream) { hipStreamSynchronize(c->hcBuildStream); hipStreamDestroy(c->hcBuildStream); }
    for (hipEvent_t ev : {c->evHcFork, c->evHcHist, c->evHcChain[0], c->evHcChain[1], c->evHcFree[0], c->evHcFree[1]}) if (ev) hipEventDestroy(ev);
    delete c;
}

// Give back what the host-buffer calls and the HC levels keep between calls: pinned host + device staging, HC workspaces.
int plz4hip_ctx_trim(plz4hip_ctx* c)
{
    if (!c) return PLZ4HIP_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    if (c->hcPending) { HIPCHK(c, hipEventSynchronize(c->hcDone)); c->hcPending = false; }
    for (auto& sl : c->slot) {
        if (sl.s) HIPCHK(c, hipStreamSynchronize(sl.s));
        if (sl.h) { hipHostFree(sl.h); sl.h = nullptr; sl.hcap = 0; }
        if (sl.d) { hipFree(sl.d); sl.d = nullptr; sl.dcap = 0; }
        if (sl.l1.d) { hipFree(sl.l1.d); sl.l1.d = nullptr; sl.l1.bytes = 0; }
    }
    for (int i = 0; i < plz4hip_ctx::kL1Shared; ++i) {
        if (c->l1Pending[i]) { HIPCHK(c, hipEventSynchronize(c->l1Done[i])); c->l1Pending[i] = false; }
        if (c->l1[i].d) { hipFree(c->l1[i].d); c->l1[i].d = nullptr; c->l1[i].bytes = 0; }
        c->l1Stream[i] = nullptr;
    }
    if (c->d_hc) { hipFree(c->d_hc); c->d_hc = nullptr; c->hcWaves = 0; }
    if (c->d_h12) { hipFree(c->d_h12); c->d_h12 = nullptr; c->h12Bytes = 0; }
    if (c->dxPending) { HIPCHK(c, hipEventSynchronize(c->dxDone)); c->dxPending = false; }
    if (c->d_dx) { hipFree(c->d_dx); c->d_dx = nullptr; c->dxBytes = 0; }
    return PLZ4HIP_OK;
}

const char* plz4hip_last_error(const plz4hip_ctx* c)
{
    if (!c) return "null ctx";
    static thread_local std::string copy;          // the caller's own copy: valid until this thread asks again
    std::lock_guard<std::mutex> g(const_cast<plz4hip_ctx*>(c)->errMu);
    copy = c->err;
    return copy.c_str();
}

int plz4hip_dev_resident_waves(plz4hip_ctx* c, int decode) { return c ? (decode ? c->decWaves : c->encWaves) : PLZ4HIP_E_ARG; }

// ---------------------------------------------------------------------------------------- device-resident API
int plz4hip_dev_compress(plz4hip_ctx* c, int nBlocks, const void* src, int64_t srcStride, const int32_t* srcLen,
                         void* dst, int64_t dstStride, const int32_t* dstCap, int level, int maxLen, int32_t* result, void* stream)
{
    if (!c || nBlocks < 0 || !srcLen || !dstCap || !result) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_compress: bad argument");
    if (level != 1 && !is_hc_level(level)) return fail(c, PLZ4HIP_E_UNSUPPORTED, "levels 1..12 are built");
    if (nBlocks == 0) return PLZ4HIP_OK;
    if (is_hc_level(level) && maxLen <= 0) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_compress: levels 2..12 need maxLen (the largest srcLen[i]; the lengths are on the device)");
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    CodecArgs a{};
    a.src = (const uint8_t*)src; a.srcStride = srcStride; a.srcLen = srcLen;
    a.dst = (uint8_t*)dst; a.dstStride = dstStride; a.dstCap = dstCap;
    a.result = result; a.nBlocks = nBlocks;
    a.dictLen = -1; a.prevTailLen = -1;
    if (maxLen > 0) {
        // the kernels index per-block workspaces sized from maxLen: they read lengths that were checked against it
        if (nBlocks > c->lenCopyCap) {
            HIPCHK(c, hipDeviceSynchronize());                           // (a copy an earlier call still reads is not freed under it)
            if (c->d_lenCopy) hipFree(c->d_lenCopy);
            c->d_lenCopy = nullptr; c->lenCopyCap = 0;
            if (hipMalloc((void**)&c->d_lenCopy, (size_t)nBlocks * 4 + 1024) != hipSuccess) return fail(c, PLZ4HIP_E_NOMEM, "plz4hip_dev_compress: length copy");
            c->lenCopyCap = nBlocks + 256;
        }
        if (c->lenPending && c->lenStream != s) HIPCHK(c, hipStreamWaitEvent(s, c->lenDone, 0));   // the last job's kernels still read the copy
        hipLaunchKernelGGL(k_check_len, dim3((nBlocks + 255) / 256), dim3(256), 0, s, srcLen, maxLen, nBlocks, c->d_lenCopy, result, 0);
        a.srcLen = c->d_lenCopy;
    }
    int rc;
    if (is_hc_level(level)) { a.level = level; rc = launch_hc(c, s, a, nBlocks, maxLen, 1); }
    else rc = launch_l1(c, s, a, nBlocks, maxLen, 1, nullptr);           // (maxLen <= 0: lengths unknown to the host -> fused kernels)
    if (rc == PLZ4HIP_OK && maxLen > 0) {
        hipLaunchKernelGGL(k_check_len, dim3((nBlocks + 255) / 256), dim3(256), 0, s, srcLen, maxLen, nBlocks, c->d_lenCopy, result, 1);
        HIPCHK(c, hipGetLastError());
    }
    if (maxLen > 0) {                                                     // (also after a failed launch: whatever was enqueued reads the copy)
        if (!c->lenDone) HIPCHK(c, hipEventCreateWithFlags(&c->lenDone, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->lenDone, s));
        c->lenPending = true; c->lenStream = s;
    }
    return rc;
}

int plz4hip_dev_decompress(plz4hip_ctx* c, int nBlocks, const void* src, int64_t srcStride, const int32_t* srcLen,
                           void* dst, int64_t dstStride, const int32_t* dstCap, int32_t* result, void* stream)
{
    if (!c || nBlocks < 0 || !srcLen || !dstCap || !result) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_decompress: bad argument");
    if (nBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    CodecArgs a{};
    a.src = (const uint8_t*)src; a.srcStride = srcStride; a.srcLen = srcLen;
    a.dst = (uint8_t*)dst; a.dstStride = dstStride; a.dstCap = dstCap;
    a.result = result; a.nBlocks = nBlocks;
    a.dictLen = -1; a.prevTailLen = -1;
    // (the lengths live on the device: the strides bound them)
    return launch_decode(c, s, a, nBlocks, nBlocks > 1 ? srcStride : (int64_t)(6 << 20), nBlocks > 1 ? dstStride : (int64_t)kDxMaxOut, false);
}

int plz4hip_dev_encode_records(plz4hip_ctx* c, const void* src, int64_t srcBytes, int bsz, int level,
                               int blockChecksum, void* stage, int32_t* recLen, void* stream)
{
    if (!c || srcBytes < 0 || bsz <= 0 || !stage || !recLen) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_encode_records: bad argument");
    if (level != 1 && !is_hc_level(level)) return fail(c, PLZ4HIP_E_UNSUPPORTED, "levels 1..12 are built");
    const int64_t nb64 = (srcBytes + bsz - 1) / bsz;
    if (nb64 > 0x7FFFFFFF) return fail(c, PLZ4HIP_E_ARG, "too many blocks");
    const int nBlocks = (int)nb64;
    if (nBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    CodecArgs a{};
    a.src = (const uint8_t*)src; a.srcStride = bsz; a.srcBytes = srcBytes; a.bsz = bsz;
    a.dst = (uint8_t*)stage; a.dstStride = plz4hip_dev_stage_stride(bsz);
    a.result = recLen; a.nBlocks = nBlocks; a.blockChecksum = blockChecksum;
    a.dictLen = -1; a.prevTailLen = -1;
    if (is_hc_level(level)) {
        a.level = level;
        return launch_hc(c, s, a, nBlocks, bsz, 0);
    }
    return launch_l1(c, s, a, nBlocks, bsz, 0, nullptr);
}

static int move_slices(int maxLen) { const int s = (maxLen + 256 * 16 * 8 - 1) / (256 * 16 * 8); return s < 1 ? 1 : s; }

int plz4hip_dev_compact_records(plz4hip_ctx* c, const void* stage, int64_t stageStride, const int32_t* recLen,
                                int nBlocks, int64_t* recOff, void* body, int64_t bodyCap, void* stream)
{
    if (!c || nBlocks < 0 || !stage || !recLen || !recOff || stageStride <= 0) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_compact_records: bad argument");
    if (body && bodyCap <= 0) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_compact_records: bodyCap");
    if (nBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, recLen, recOff, nBlocks);
    HIPCHK(c, hipGetLastError());
    if (body) {
        hipLaunchKernelGGL(k_move_records, dim3(nBlocks, move_slices((int)stageStride)), dim3(256), 0, s,
                           (const uint8_t*)stage, (const int64_t*)nullptr, stageStride, recLen, (const int64_t*)recOff, (uint8_t*)body, bodyCap);
        HIPCHK(c, hipGetLastError());
    }
    return PLZ4HIP_OK;
}

int plz4hip_dev_scatter_records(plz4hip_ctx* c, const void* src, const int64_t* srcOff, const int32_t* len,
                                const int64_t* dstOff, int n, int maxLen, void* dst, int64_t dstCap, void* stream)
{
    if (!c || n < 0 || !src || !srcOff || !len || !dstOff || !dst || maxLen < 0 || dstCap < 0) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_scatter_records: bad argument");
    if (n == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_move_records, dim3(n, move_slices(maxLen)), dim3(256), 0, s,
                       (const uint8_t*)src, srcOff, (int64_t)0, len, dstOff, (uint8_t*)dst, dstCap);
    HIPCHK(c, hipGetLastError());
    return PLZ4HIP_OK;
}

int plz4hip_dev_decode_records(plz4hip_ctx* c, const void* body, const int64_t* recOff, int nBlocks,
                               int bsz, int blockChecksum, void* dst, int64_t dstStride, int dstCap,
                               int32_t* result, int32_t* status, void* stream)
{
    if (!c || nBlocks < 0 || !body || !recOff || !dst || !result || !status || bsz <= 0) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_decode_records: bad argument");
    if (nBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    CodecArgs a{};
    a.src = (const uint8_t*)body; a.recOff = recOff; a.bsz = bsz;
    a.dst = (uint8_t*)dst; a.dstStride = dstStride; a.dstCapAll = dstCap;
    a.result = result; a.status = status; a.nBlocks = nBlocks; a.blockChecksum = blockChecksum;
    a.dictLen = -1; a.prevTailLen = -1;
    return launch_decode(c, s, a, nBlocks, bsz, dstCap, true);
}

int plz4hip_dev_duplex_records(plz4hip_ctx* c, const void* src, int64_t srcBytes, int bsz, int blockChecksum, void* stage, int32_t* recLen,
                               const void* body, const int64_t* recOff, int nDecBlocks, int decBsz, int decBlockChecksum,
                               void* dst, int64_t dstStride, int dstCap, int32_t* result, int32_t* status, void* stream)
{
    if (!c || srcBytes < 0 || bsz <= 0 || !stage || !recLen) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_duplex_records: bad argument (encode side)");
    if (nDecBlocks < 0 || !body || !recOff || !dst || !result || !status || decBsz <= 0) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_duplex_records: bad argument (decode side)");
    const int64_t nb64 = (srcBytes + bsz - 1) / bsz;
    if (nb64 > 0x7FFFFFFF) return fail(c, PLZ4HIP_E_ARG, "too many blocks");
    const int nBlocks = (int)nb64;
    if (nBlocks == 0 && nDecBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    CodecArgs d{};
    d.src = (const uint8_t*)body; d.recOff = recOff; d.bsz = decBsz;
    d.dst = (uint8_t*)dst; d.dstStride = dstStride; d.dstCapAll = dstCap;
    d.result = result; d.status = status; d.nBlocks = nDecBlocks; d.blockChecksum = decBlockChecksum;
    d.dictLen = -1; d.prevTailLen = -1;
    if (nDecBlocks > 0) { hipError_t e; d.queue = next_queue(c, s, &e); HIPCHK(c, e); }
    if (nBlocks == 0) {
        hipLaunchKernelGGL(k_decode_rec, dim3(grid_for(nDecBlocks, c->decWaves)), dim3(64), 0, s, d);
        HIPCHK(c, hipGetLastError());
        return PLZ4HIP_OK;
    }
    CodecArgs a{};
    a.src = (const uint8_t*)src; a.srcStride = bsz; a.srcBytes = srcBytes; a.bsz = bsz;
    a.dst = (uint8_t*)stage; a.dstStride = plz4hip_dev_stage_stride(bsz);
    a.result = recLen; a.nBlocks = nBlocks; a.blockChecksum = blockChecksum;
    a.dictLen = -1; a.prevTailLen = -1;
    return launch_l1(c, s, a, nBlocks, bsz, 0, nullptr, nullptr, nDecBlocks > 0 ? &d : nullptr);
}

// plz4hip_dev_encode_records + plz4hip_dev_compact_records in one: the records go straight to their place in the frame body
// (the emit stage knows every record's length before it writes a byte: k_l1_scan, then a scan over the blocks).  No staging
// area, no second pass over the records.  Level 1 and 2 (the staged call); recOff[nBlocks] > bodyCap: the records beyond the
// room were not written.
static int encode_body_args(plz4hip_ctx* c, CodecArgs& a, const void* src, int64_t srcBytes, int bsz, int blockChecksum,
                            void* body, int64_t bodyCap, int64_t* recOff, int32_t* recLen, int* nBlocksOut, const char* who)
{
    if (!c || srcBytes < 0 || bsz <= 0 || bsz > kSeqMaxBlock || !body || bodyCap <= 0 || !recOff || !recLen) return fail(c, PLZ4HIP_E_ARG, who);
    const int64_t nb64 = (srcBytes + bsz - 1) / bsz;
    if (nb64 > 0x7FFFFFFF) return fail(c, PLZ4HIP_E_ARG, "too many blocks");
    *nBlocksOut = (int)nb64;
    a.src = (const uint8_t*)src; a.srcStride = bsz; a.srcBytes = srcBytes; a.bsz = bsz;
    a.dst = (uint8_t*)body; a.dstStride = 0; a.bodyOff = recOff; a.bodyCap = bodyCap;
    a.result = recLen; a.nBlocks = *nBlocksOut; a.blockChecksum = blockChecksum;
    a.dictLen = -1; a.prevTailLen = -1;
    return PLZ4HIP_OK;
}
int plz4hip_dev_encode_body(plz4hip_ctx* c, const void* src, int64_t srcBytes, int bsz, int level, int blockChecksum,
                            void* body, int64_t bodyCap, int64_t* recOff, int32_t* recLen, void* stream)
{
    if (level != 1 && level != 2) return fail(c, PLZ4HIP_E_UNSUPPORTED, "plz4hip_dev_encode_body: levels 1 and 2 (the others: plz4hip_dev_encode_records + plz4hip_dev_compact_records)");
    CodecArgs a{}; int nBlocks = 0;
    if (int rc = encode_body_args(c, a, src, srcBytes, bsz, blockChecksum, body, bodyCap, recOff, recLen, &nBlocks, "plz4hip_dev_encode_body: bad argument")) return rc;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if (nBlocks == 0) { HIPCHK(c, hipMemsetAsync(recOff, 0, sizeof(int64_t), s)); return PLZ4HIP_OK; }
    if (level == 2) { a.level = 2; return launch_hc(c, s, a, nBlocks, bsz, 0); }
    return launch_l1(c, s, a, nBlocks, bsz, 0, nullptr);
}
// ... and with the record decode of another batch in the same launch (plz4hip_dev_duplex_records' counterpart)
int plz4hip_dev_duplex_body(plz4hip_ctx* c, const void* src, int64_t srcBytes, int bsz, int blockChecksum,
                            void* body, int64_t bodyCap, int64_t* recOff, int32_t* recLen,
                            const void* decBody, const int64_t* decRecOff, int nDecBlocks, int decBsz, int decBlockChecksum,
                            void* dst, int64_t dstStride, int dstCap, int32_t* result, int32_t* status, void* stream)
{
    CodecArgs a{}; int nBlocks = 0;
    if (int rc = encode_body_args(c, a, src, srcBytes, bsz, blockChecksum, body, bodyCap, recOff, recLen, &nBlocks, "plz4hip_dev_duplex_body: bad argument (encode side)")) return rc;
    if (nDecBlocks < 0 || (nDecBlocks > 0 && (!decBody || !decRecOff || !dst || !result || !status || decBsz <= 0))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_duplex_body: bad argument (decode side)");
    if (nBlocks == 0 && nDecBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    CodecArgs d{};
    d.src = (const uint8_t*)decBody; d.recOff = decRecOff; d.bsz = decBsz;
    d.dst = (uint8_t*)dst; d.dstStride = dstStride; d.dstCapAll = dstCap;
    d.result = result; d.status = status; d.nBlocks = nDecBlocks; d.blockChecksum = decBlockChecksum;
    d.dictLen = -1; d.prevTailLen = -1;
    if (nDecBlocks > 0) { hipError_t e; d.queue = next_queue(c, s, &e); HIPCHK(c, e); }
    if (nBlocks == 0) {
        HIPCHK(c, hipMemsetAsync(recOff, 0, sizeof(int64_t), s));
        hipLaunchKernelGGL(k_decode_rec, dim3(grid_for(nDecBlocks, c->decWaves)), dim3(64), 0, s, d);
        HIPCHK(c, hipGetLastError());
        return PLZ4HIP_OK;
    }
    return launch_l1(c, s, a, nBlocks, bsz, 0, nullptr, nullptr, nDecBlocks > 0 ? &d : nullptr);
}

// ---------------------------------------------------------------------------------------- host-buffer API
// Staging layout on both sides: [int32 lenA[n]] [int32 lenB[n]] [int32 res[n]] [int32 st[n]] [in blocks @inStride] [out blocks @outStride]
namespace {
// Layout of one chunk in a slot (same offsets in the pinned host buffer and in the device buffer):
//   [srcLen n][dstCap n][result n][status n][outLen n][outOff n+1 (int64)] | inputs at inStride | outputs at outStride
// and, device only, the outputs once more back to back (compacted) -- that is what travels back over PCIe.
struct Staging { size_t offA, offB, offRes, offSt, offLen, offOff, offIn, offOut, offPack, total; int64_t inStride, outStride; size_t gap; };

// gap: bytes of scratch in front of every input block (HC with a dictionary / linked blocks: the external segment goes there)
Staging plan(int n, int maxIn, int maxOut, size_t gap = 0)
{
    Staging s{};
    s.gap = gap;
    const size_t arr = round_up((size_t)n * 4, 256);
    s.offA = 0; s.offB = arr; s.offRes = 2 * arr; s.offSt = 3 * arr; s.offLen = 4 * arr;
    s.offOff = 5 * arr;
    s.offIn = s.offOff + round_up((size_t)(n + 1) * 8, 256);
    s.inStride = (int64_t)round_up((size_t)maxIn + gap + 16, 16);
    s.outStride = (int64_t)round_up((size_t)maxOut + 16, 16);
    s.offOut = s.offIn + (size_t)n * s.inStride;
    s.offPack = s.offOut + (size_t)n * s.outStride;
    s.total = s.offPack + (size_t)n * s.outStride;
    return s;
}
}  // namespace

struct DictJob {                       // optional dictionary / linked parameters of a host call
    const plz4hip_dict* dict = nullptr;
    int   linked = 0;
    const void* prevTail = nullptr; int prevTailLen = -1;
    uint8_t* window = nullptr; int* windowLen = nullptr;      // host buffers (64 KiB, one length) per linked decode chain
    int   nChains = 1; const int32_t* chainFirst = nullptr;   // several chains in one call (plz4hip_decode_records_chains)
    bool  any = false;
    int   level = 1;
};

// Host-buffer entry points.  A call is cut into chunks of at most kChunkBytes (6 GiB; PLZ4HIP_HOST_CHUNK_MB) of staging; up to kSlots chunks are in flight,
// each on its own stream: while the GPU works on one, the host threads fill the next and empty the previous, H2D / kernel /
// D2H of different chunks overlap, and kernels of several chunks share the GPU.  Only the bytes produced travel back
// (outputs are compacted on the device first).  A linked chain (block i needs block i-1) stays one chunk.
static int host_codec(plz4hip_ctx* c, int mode /*0 enc raw,1 dec raw,2 enc rec,3 dec rec,4 xxh*/, int nBlocks,
                      const void* const* src, const int32_t* srcLen, void* const* dst, const int32_t* dstCap,
                      int bsz, int blockChecksum, int32_t* result, int32_t* status, const DictJob* dj = nullptr)
{
    if (nBlocks == 0) return PLZ4HIP_OK;
    int maxIn = 0, maxOut = 0;
    for (int i = 0; i < nBlocks; ++i) {
        if (srcLen[i] < 0 || (srcLen[i] > 0 && !src[i])) return fail(c, PLZ4HIP_E_ARG, "bad source block");
        if (srcLen[i] > maxIn) maxIn = srcLen[i];
        const int oc = (mode == 0 || mode == 1) ? dstCap[i] : (mode == 4 ? 4 : bsz + 8);
        if (oc < 0) return fail(c, PLZ4HIP_E_ARG, "negative capacity");
        if (oc > maxOut) maxOut = oc;
    }
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    const bool dictMode = dj && dj->any;
    const bool hcMode = dj && is_hc_level(dj->level);
    plz4hip_xxh32_stream* const hash = (mode == 2 || (mode == 3 && !(dj && dj->nChains > 1))) ? c->contentHash : nullptr;
    if (hash && !c->hashStream) HIPCHK(c, hipStreamCreateWithFlags(&c->hashStream, hipStreamNonBlocking));
    // a linked DECODE is one serial chain (block i needs block i-1's output): one chunk.  A linked ENCODE needs only the
    // previous block's source tail, which the caller's buffers hold: it is cut into chunks like any other call, the first
    // block of a later chunk getting the tail of the block before it as its prevTail.
    const bool chained = dictMode && dj->linked && mode == 3;
    if (dictMode && dj->prevTail && dj->prevTailLen > 65536) return fail(c, PLZ4HIP_E_ARG, "prevTail longer than 64 KiB");

    size_t kChunkBytes = (size_t)4 << 30;      // x kSlots in flight (smaller chunks starve the encoder's wave slots); pinned host + device staging stay allocated until plz4hip_ctx_trim / destroy
    // the HC levels run one chunk at a time (below) and live on the number of blocks in flight: one chunk of three times the
    // size (4 MiB blocks: 1024 instead of 341 per chunk) for the same staging memory
    if (hcMode) kChunkBytes *= plz4hip_ctx::kSlots;
    if (const char* v = getenv("PLZ4HIP_HOST_CHUNK_MB")) { const long mb = atol(v); if (mb > 0) kChunkBytes = (size_t)mb << 20; }   // tests: force many chunks
    const size_t gap = (dictMode && (mode == 0 || mode == 2)) ? 65536 : 0;     // encoders with a dictionary / linked blocks: room for the external segment
    int cb = nBlocks;
    {
        const Staging one = plan(1, maxIn, maxOut, gap);
        const size_t per = (size_t)one.inStride + 2 * (size_t)one.outStride;
        if (!chained && per * (size_t)nBlocks > kChunkBytes) { cb = (int)(kChunkBytes / per); if (cb < 1) cb = 1; }
    }
    const int nChunks = (nBlocks + cb - 1) / cb;
    // HC kernels index one shared workspace by workgroup: never two of them at once
    const int nSlots = hcMode ? 1 : (nChunks < plz4hip_ctx::kSlots ? nChunks : plz4hip_ctx::kSlots);
    const Staging st = plan(cb, maxIn, maxOut, gap);
    // [prevTail 64 KiB][per chain: window 2 x 64 KiB][windowLen per chain][chainFirst]
    const int    nCh = (dictMode && dj->window) ? dj->nChains : 1;
    const size_t offExtra = st.total;
    const size_t offWin = offExtra + 65536, offWinLen = offWin + (size_t)nCh * 131072, offFirst = offWinLen + round_up((size_t)nCh * 4, 256);
    const size_t winBytes = (offFirst + round_up((size_t)(nCh + 1) * 4, 256)) - offWin;
    const size_t slotBytes = st.total + (dictMode ? 65536 + winBytes : 0);
    for (int i = 0; i < nSlots; ++i) if (int rc = ensure_slot(c, i, slotBytes, slotBytes)) return rc;

    auto submit = [&](int k) -> int {
        plz4hip_ctx::HostSlot& sl = c->slot[k % nSlots];
        const int b0 = k * cb, nb = (nBlocks - b0 < cb) ? nBlocks - b0 : cb;
        hipStream_t s = sl.s;
        int32_t* hA = (int32_t*)(sl.h + st.offA);
        int32_t* hB = (int32_t*)(sl.h + st.offB);
        size_t inBytes = 0;
        for (int i = 0; i < nb; ++i) { hA[i] = srcLen[b0 + i]; hB[i] = (mode == 0 || mode == 1) ? dstCap[b0 + i] : 0; inBytes += (size_t)srcLen[b0 + i]; }
        parallel_blocks(nb, inBytes, [&](int i) {
            if (srcLen[b0 + i]) memcpy(sl.h + st.offIn + st.gap + (size_t)i * st.inStride, src[b0 + i], (size_t)srcLen[b0 + i]);
        });
        HIPCHK(c, hipMemcpyAsync(sl.d, sl.h, st.offRes, hipMemcpyHostToDevice, s));                                   // srcLen, dstCap
        HIPCHK(c, hipMemcpyAsync(sl.d + st.offIn, sl.h + st.offIn, (size_t)nb * st.inStride, hipMemcpyHostToDevice, s));
        if (hash && mode == 2) {
            if (!sl.evData) HIPCHK(c, hipEventCreateWithFlags(&sl.evData, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(sl.evData, s));
        }
        hipError_t e; uint32_t* q = next_queue(c, s, &e); HIPCHK(c, e);
        CodecArgs a{};
        a.src = sl.d + st.offIn + st.gap; a.srcStride = st.inStride; a.srcLen = (const int32_t*)(sl.d + st.offA);
        a.dst = sl.d + st.offOut; a.dstStride = st.outStride; a.dstCap = (const int32_t*)(sl.d + st.offB);
        a.result = (int32_t*)(sl.d + st.offRes); a.status = (int32_t*)(sl.d + st.offSt);
        a.queue = q; a.nBlocks = nb; a.bsz = bsz; a.blockChecksum = blockChecksum; a.dstCapAll = bsz + 8;
        a.dictLen = -1; a.prevTailLen = -1;
        if (hcMode) a.level = dj->level;
        if (dictMode) {
            if (dj->dict) { a.dict = dj->dict->d_bytes; a.dictLen = dj->dict->len; a.dictTable = dj->dict->d_table; }
            if (hcMode) {
                a.hcEx = 1;
                if (dj->dict) {
                    const uint8_t* t = dj->dict->d_hc + (dj->level <= 2 ? 0 : (size_t)kHcWorkBytes);
                    a.hcDictHash = (const uint32_t*)t; a.hcDictChain = (const uint16_t*)(t + kHcHashEntries * 4);
                }
            }
            a.linked = dj->linked;
            const void* pt = dj->prevTail; int ptLen = (dj->prevTail && dj->prevTailLen >= 0) ? dj->prevTailLen : -1;
            if (dj->linked && b0 > 0) {                                   // a later chunk of a linked encode
                const int pl = srcLen[b0 - 1]; ptLen = pl < 65536 ? pl : 65536;
                pt = (const uint8_t*)src[b0 - 1] + (pl - ptLen);
            }
            if (ptLen >= 0) {
                if (ptLen) memcpy(sl.h + offExtra, pt, (size_t)ptLen);
                HIPCHK(c, hipMemcpyAsync(sl.d + offExtra, sl.h + offExtra, (size_t)ptLen + 16, hipMemcpyHostToDevice, s));
                a.prevTail = sl.d + offExtra; a.prevTailLen = ptLen;
            }
            if (dj->window) {
                for (int ch = 0; ch < nCh; ++ch) memcpy(sl.h + offWin + (size_t)ch * 131072, dj->window + (size_t)ch * 65536, 65536);
                memcpy(sl.h + offWinLen, dj->windowLen, (size_t)nCh * sizeof(int));
                if (dj->chainFirst) memcpy(sl.h + offFirst, dj->chainFirst, (size_t)(nCh + 1) * sizeof(int32_t));
                HIPCHK(c, hipMemcpyAsync(sl.d + offWin, sl.h + offWin, winBytes, hipMemcpyHostToDevice, s));
                a.window = sl.d + offWin; a.windowLen = (int*)(sl.d + offWinLen);
                a.nChains = nCh; a.chainFirst = dj->chainFirst ? (const int32_t*)(sl.d + offFirst) : nullptr;
            }
        }
        switch (mode) {
        case 0: if (hcMode) { if (int rc = launch_hc(c, s, a, nb, maxIn, 1)) return rc; }
                else if (dictMode) ENC_LAUNCH(k_encode_raw_dict, nb, c, s, a);
                else { if (int rc = launch_l1(c, s, a, nb, maxIn, 1, &sl.l1)) return rc; } break;
        case 1: if (dictMode) hipLaunchKernelGGL(k_decode_raw_dict, dim3(grid_for(nb, c->decWaves)), dim3(64), 0, s, a);
                else { if (int rc = launch_decode(c, s, a, nb, maxIn, maxOut, false)) return rc; } break;
        case 2: a.dstCap = nullptr;
                if (hcMode) { if (int rc = launch_hc(c, s, a, nb, maxIn, 0)) return rc; }
                else if (dictMode) ENC_LAUNCH(k_encode_rec_dict, nb, c, s, a);
                else { if (int rc = launch_l1(c, s, a, nb, maxIn, 0, &sl.l1)) return rc; } break;
        case 3: a.dstCap = nullptr;
                if (dictMode && dj->linked) hipLaunchKernelGGL(k_decode_rec_linked, dim3(grid_for(nCh, c->decWaves)), dim3(64), 0, s, a);
                else if (dictMode) hipLaunchKernelGGL(k_decode_rec_dict, dim3(grid_for(nb, c->decWaves)), dim3(64), 0, s, a);
                else { if (int rc = launch_decode(c, s, a, nb, bsz, bsz + 8, true)) return rc; } break;
        case 4: hipLaunchKernelGGL(k_xxh32, dim3(grid_for(nb, c->decWaves)), dim3(64), 0, s,
                                   (const uint8_t*)a.src, a.srcStride, a.srcLen, (uint32_t*)a.result, nb, q); break;
        }
        HIPCHK(c, hipGetLastError());
        if (hash && (mode == 2 || mode == 3)) {
            // The content checksum of this chunk's plaintext, in block order, on the ctx's hash stream: it starts once the
            // chunk's plaintext is on the device (encode: the copy in; decode: the decode kernel) and runs beside whatever the
            // slot's stream does next; chunks follow each other on the hash stream.
            if (!sl.evData) HIPCHK(c, hipEventCreateWithFlags(&sl.evData, hipEventDisableTiming));
            if (!sl.evHash) HIPCHK(c, hipEventCreateWithFlags(&sl.evHash, hipEventDisableTiming));
            if (mode == 3) HIPCHK(c, hipEventRecord(sl.evData, s));             // (encode: recorded right behind the copy in)
            HIPCHK(c, hipStreamWaitEvent(c->hashStream, sl.evData, 0));
            if (mode == 2) hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(64), 0, c->hashStream, hash->d_state, (const uint8_t*)a.src, a.srcStride, a.srcLen, nb, (int64_t)0, 0);
            else           hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(64), 0, c->hashStream, hash->d_state, (const uint8_t*)a.dst, a.dstStride, (const int32_t*)a.result, nb, (int64_t)0, 0);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(sl.evHash, c->hashStream));
            HIPCHK(c, hipEventRecord(hash->done, c->hashStream));
            sl.hashBusy = true;
        }
        if (mode != 4) {                                       // pack the outputs: sizes -> offsets -> back to back
            int32_t* dLen = (int32_t*)(sl.d + st.offLen); int64_t* dOff = (int64_t*)(sl.d + st.offOff);
            hipLaunchKernelGGL(k_out_len, dim3((nb + 255) / 256), dim3(256), 0, s, (const int32_t*)a.result, dLen, nb);
            hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, (const int32_t*)dLen, dOff, nb);
            hipLaunchKernelGGL(k_move_records, dim3(nb, move_slices((int)st.outStride)), dim3(256), 0, s,
                               (const uint8_t*)(sl.d + st.offOut), (const int64_t*)nullptr, st.outStride, (const int32_t*)dLen,
                               (const int64_t*)dOff, sl.d + st.offPack, (int64_t)nb * st.outStride);
            HIPCHK(c, hipGetLastError());
        }
        // results, status, sizes and offsets now; the packed bytes once their total is known (retire)
        HIPCHK(c, hipMemcpyAsync(sl.h + st.offRes, sl.d + st.offRes, st.offIn - st.offRes, hipMemcpyDeviceToHost, s));
        if (dictMode && dj->window)
            HIPCHK(c, hipMemcpyAsync(sl.h + offWin, sl.d + offWin, offFirst - offWin, hipMemcpyDeviceToHost, s));
        return PLZ4HIP_OK;
    };
    auto retire = [&](int k) -> int {
        plz4hip_ctx::HostSlot& sl = c->slot[k % nSlots];
        const int b0 = k * cb, nb = (nBlocks - b0 < cb) ? nBlocks - b0 : cb;
        HIPCHK(c, hipStreamSynchronize(sl.s));
        if (sl.hashBusy) { HIPCHK(c, hipEventSynchronize(sl.evHash)); sl.hashBusy = false; }     // the slot's buffers are free again
        const int32_t* hRes = (const int32_t*)(sl.h + st.offRes);
        const int32_t* hSt  = (const int32_t*)(sl.h + st.offSt);
        const int32_t* hLen = (const int32_t*)(sl.h + st.offLen);
        const int64_t* hOff = (const int64_t*)(sl.h + st.offOff);
        if (mode != 4 && hOff[nb] > 0) {
            HIPCHK(c, hipMemcpyAsync(sl.h + st.offOut, sl.d + st.offPack, (size_t)hOff[nb], hipMemcpyDeviceToHost, sl.s));
            HIPCHK(c, hipStreamSynchronize(sl.s));
        }
        if (dictMode && dj->window) {
            for (int ch = 0; ch < nCh; ++ch) memcpy(dj->window + (size_t)ch * 65536, sl.h + offWin + (size_t)ch * 131072, 65536);
            memcpy(dj->windowLen, sl.h + offWinLen, (size_t)nCh * sizeof(int));
        }
        for (int i = 0; i < nb; ++i) { result[b0 + i] = hRes[i]; if (status) status[b0 + i] = hSt[i]; }
        if (mode != 4)
            parallel_blocks(nb, (size_t)hOff[nb], [&](int i) {
                if (hLen[i] > 0 && dst[b0 + i]) memcpy(dst[b0 + i], sl.h + st.offOut + (size_t)hOff[i], (size_t)hLen[i]);
            });
        return PLZ4HIP_OK;
    };
    int rc = PLZ4HIP_OK;
    int retired = 0;
    for (int k = 0; k < nChunks && rc == PLZ4HIP_OK; ++k) {
        if (k >= nSlots) { rc = retire(retired++); if (rc) break; }       // its slot is the one chunk k needs
        rc = submit(k);
    }
    for (; retired < nChunks && rc == PLZ4HIP_OK; ++retired) rc = retire(retired);
    if (rc != PLZ4HIP_OK) for (int i = 0; i < nSlots; ++i) if (c->slot[i].s) hipStreamSynchronize(c->slot[i].s);   // nothing left in flight
    if (rc == PLZ4HIP_OK && hcMode && c->d_h12 && dj->level >= 12) {
        // the level-12 search kernel bounds its spins and raises this flag if it ever gives up (never seen): an engine failure,
        // not a result
        int32_t flag = 0;
        HIPCHK(c, copy_sync(c, &flag, c->d_h12 + c->h12ErrOff, 4, hipMemcpyDeviceToHost));
        if (flag) return fail(c, PLZ4HIP_E_DEVICE, "level-12 search kernel gave up (spin guard)");
    }
    return rc;
}

int plz4hip_compress_batch(plz4hip_ctx* c, int nBlocks, const void* const* src, const int32_t* srcLen,
                           void* const* dst, const int32_t* dstCap, int level, int32_t* result)
{
    if (!c || nBlocks < 0 || (nBlocks && (!src || !srcLen || !dst || !dstCap || !result))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_compress_batch: bad argument");
    if (is_hc_level(level)) { DictJob j; j.level = level; return host_codec(c, 0, nBlocks, src, srcLen, dst, dstCap, 0, 0, result, nullptr, &j); }
    if (level != 1) return fail(c, PLZ4HIP_E_UNSUPPORTED, "levels 1..12 are built");
    return host_codec(c, 0, nBlocks, src, srcLen, dst, dstCap, 0, 0, result, nullptr);
}

int plz4hip_decompress_batch(plz4hip_ctx* c, int nBlocks, const void* const* src, const int32_t* srcLen,
                             void* const* dst, const int32_t* dstCap, int32_t* result)
{
    if (!c || nBlocks < 0 || (nBlocks && (!src || !srcLen || !dst || !dstCap || !result))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_decompress_batch: bad argument");
    return host_codec(c, 1, nBlocks, src, srcLen, dst, dstCap, 0, 0, result, nullptr);
}

int plz4hip_xxh32_batch(plz4hip_ctx* c, int n, const void* const* buf, const int32_t* len, uint32_t* out)
{
    if (!c || n < 0 || (n && (!buf || !len || !out))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_xxh32_batch: bad argument");
    return host_codec(c, 4, n, buf, len, nullptr, nullptr, 0, 0, (int32_t*)out, nullptr);
}

int plz4hip_encode_records(plz4hip_ctx* c, int nBlocks, const void* const* src, const int32_t* srcLen,
                           int bsz, int level, int blockChecksum, void* const* rec, int32_t* recLen)
{
    if (!c || nBlocks < 0 || bsz <= 0 || (nBlocks && (!src || !srcLen || !rec || !recLen))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_encode_records: bad argument");
    for (int i = 0; i < nBlocks; ++i) if (srcLen[i] > bsz) return fail(c, PLZ4HIP_E_ARG, "source block larger than block size");
    if (is_hc_level(level)) { DictJob j; j.level = level; return host_codec(c, 2, nBlocks, src, srcLen, rec, nullptr, bsz, blockChecksum, recLen, nullptr, &j); }
    if (level != 1) return fail(c, PLZ4HIP_E_UNSUPPORTED, "levels 1..12 are built");
    return host_codec(c, 2, nBlocks, src, srcLen, rec, nullptr, bsz, blockChecksum, recLen, nullptr);
}

int plz4hip_decode_records(plz4hip_ctx* c, int nBlocks, const void* const* rec, const int32_t* recLen,
                           int bsz, int blockChecksum, void* const* dst, int32_t* result, int32_t* status)
{
    if (!c || nBlocks < 0 || bsz <= 0 || (nBlocks && (!rec || !recLen || !dst || !result || !status))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_decode_records: bad argument");
    for (int i = 0; i < nBlocks; ++i) if (recLen[i] < 4) return fail(c, PLZ4HIP_E_ARG, "record shorter than its size word");
    return host_codec(c, 3, nBlocks, rec, recLen, dst, nullptr, bsz, blockChecksum, result, status);
}


// ---------------------------------------------------------------------------------------- streaming content checksum
int plz4hip_xxh32_stream_create(plz4hip_ctx* c, plz4hip_xxh32_stream** out)
{
    if (!c || !out) return fail(c, PLZ4HIP_E_ARG, "plz4hip_xxh32_stream_create: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    if (!c->hashStream) HIPCHK(c, hipStreamCreateWithFlags(&c->hashStream, hipStreamNonBlocking));
    plz4hip_xxh32_stream* h = new (std::nothrow) plz4hip_xxh32_stream();
    if (!h) return fail(c, PLZ4HIP_E_NOMEM, "plz4hip_xxh32_stream");
    hipError_t e = hipMalloc((void**)&h->d_state, sizeof(XxhStream));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->done, hipEventDisableTiming);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(64), 0, c->hashStream, h->d_state, (const uint8_t*)nullptr, (int64_t)0, (const int32_t*)nullptr, 0, (int64_t)0, 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(h->done, c->hashStream);
    }
    if (e != hipSuccess) { if (h->d_state) hipFree(h->d_state); if (h->done) hipEventDestroy(h->done); delete h; return fail(c, PLZ4HIP_E_DEVICE, "plz4hip_xxh32_stream_create", e); }
    *out = h;
    return PLZ4HIP_OK;
}

void plz4hip_xxh32_stream_destroy(plz4hip_ctx* c, plz4hip_xxh32_stream* h)
{
    if (!h) return;
    DeviceGuard dg(c ? c->device : 0);
    if (c) { std::lock_guard<std::mutex> g(c->mu); if (c->contentHash == h) c->contentHash = nullptr; }
    if (h->done) { hipEventSynchronize(h->done); hipEventDestroy(h->done); }
    if (h->d_state) hipFree(h->d_state);
    delete h;
}

int plz4hip_xxh32_stream_reset(plz4hip_ctx* c, plz4hip_xxh32_stream* h)
{
    if (!c || !h) return fail(c, PLZ4HIP_E_ARG, "plz4hip_xxh32_stream_reset: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(64), 0, c->hashStream, h->d_state, (const uint8_t*)nullptr, (int64_t)0, (const int32_t*)nullptr, 0, (int64_t)0, 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(h->done, c->hashStream));
    return PLZ4HIP_OK;
}

int plz4hip_dev_xxh32_stream_update(plz4hip_ctx* c, plz4hip_xxh32_stream* h, const void* data, int64_t n, void* stream)
{
    if (!c || !h || n < 0 || (n && !data)) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dev_xxh32_stream_update: bad argument");
    if (n == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipStreamWaitEvent(s, h->done, 0));                       // behind the last update, whichever stream it ran on
    hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(64), 0, s, h->d_state, (const uint8_t*)data, (int64_t)0, (const int32_t*)nullptr, 0, n, 0);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(h->done, s));
    return PLZ4HIP_OK;
}

int plz4hip_xxh32_stream_update(plz4hip_ctx* c, plz4hip_xxh32_stream* h, const void* data, int64_t n)
{
    if (!c || !h || n < 0 || (n && !data)) return fail(c, PLZ4HIP_E_ARG, "plz4hip_xxh32_stream_update: bad argument");
    if (n == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    const size_t piece = (size_t)64 << 20;
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, (size_t)n < piece ? (size_t)n : piece));
    int rc = PLZ4HIP_OK;
    for (int64_t o = 0; o < n && rc == PLZ4HIP_OK; o += (int64_t)piece) {
        const size_t k = (size_t)(n - o) < piece ? (size_t)(n - o) : piece;
        hipError_t e = hipMemcpyAsync(d, (const uint8_t*)data + o, k, hipMemcpyHostToDevice, c->hashStream);
        if (e == hipSuccess) { hipLaunchKernelGGL(k_xxh32_stream, dim3(1), dim3(64), 0, c->hashStream, h->d_state, (const uint8_t*)d, (int64_t)0, (const int32_t*)nullptr, 0, (int64_t)k, 0); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipStreamSynchronize(c->hashStream);
        if (e != hipSuccess) rc = fail(c, PLZ4HIP_E_DEVICE, "plz4hip_xxh32_stream_update", e);
    }
    hipFree(d);
    if (rc == PLZ4HIP_OK) HIPCHK(c, hipEventRecord(h->done, c->hashStream));
    return rc;
}

int plz4hip_xxh32_stream_sum(plz4hip_ctx* c, plz4hip_xxh32_stream* h, uint32_t* out)
{
    if (!c || !h || !out) return fail(c, PLZ4HIP_E_ARG, "plz4hip_xxh32_stream_sum: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    HIPCHK(c, hipEventSynchronize(h->done));
    XxhStream st;
    HIPCHK(c, copy_sync(c, &st, h->d_state, sizeof st, hipMemcpyDeviceToHost));
    *out = xxh32_stream_sum(st);
    return PLZ4HIP_OK;
}

int plz4hip_ctx_set_content_hash(plz4hip_ctx* c, plz4hip_xxh32_stream* h)
{
    if (!c) return PLZ4HIP_E_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->contentHash = h;
    return PLZ4HIP_OK;
}

// ---------------------------------------------------------------------------------------- dictionaries / linked blocks
int plz4hip_dict_create(plz4hip_ctx* c, const void* dict, int dictLen, plz4hip_dict** out)
{
    if (!c || !out || dictLen < 0 || (dictLen && !dict)) return fail(c, PLZ4HIP_E_ARG, "plz4hip_dict_create: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    ENTER_DEVICE(c);
    plz4hip_dict* d = new (std::nothrow) plz4hip_dict();
    if (!d) return fail(c, PLZ4HIP_E_NOMEM, "plz4hip_dict");
    const uint8_t* p = (const uint8_t*)dict;
    if (dictLen > 65536) { p += dictLen - 65536; dictLen = 65536; }             // compress/dict.go:43-56, lz4.c:1617
    d->h_bytes.assign(p, p + dictLen);
    d->len = dictLen;
    std::vector<uint32_t> tab(4096);
    build_dict_table_slow(d->h_bytes.data(), dictLen, tab.data());
    hipError_t e = hipMalloc((void**)&d->d_bytes, 65536 + 64);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_table, 4096 * sizeof(uint32_t));
    if (e == hipSuccess && dictLen) e = copy_sync(c, d->d_bytes, d->h_bytes.data(), (size_t)dictLen, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = copy_sync(c, d->d_table, tab.data(), 4096 * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_hc, 2 * (size_t)kHcWorkBytes);
    if (e == hipSuccess) {                                                       // clz4.NewDictCtxHC for both table strategies
        hipLaunchKernelGGL(k_hc_dict_prime, dim3(2), dim3(64), 0, c->stream, (const uint8_t*)d->d_bytes, dictLen, d->d_hc);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) { if (d->d_bytes) hipFree(d->d_bytes); if (d->d_table) hipFree(d->d_table); if (d->d_hc) hipFree(d->d_hc); delete d; return fail(c, PLZ4HIP_E_DEVICE, "plz4hip_dict_create", e); }
    *out = d;
    return PLZ4HIP_OK;
}

void plz4hip_dict_destroy(plz4hip_ctx* c, plz4hip_dict* d)
{
    if (!d) return;
    DeviceGuard dg(c ? c->device : 0);
    if (d->d_bytes) hipFree(d->d_bytes);
    if (d->d_table) hipFree(d->d_table);
    if (d->d_hc) hipFree(d->d_hc);
    delete d;
}

int plz4hip_compress_batch_dict(plz4hip_ctx* c, int nBlocks, const void* const* src, const int32_t* srcLen,
                                void* const* dst, const int32_t* dstCap, int level, const plz4hip_dict* dict, int32_t* result)
{
    if (!c || !dict || nBlocks < 0 || (nBlocks && (!src || !srcLen || !dst || !dstCap || !result))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_compress_batch_dict: bad argument");
    if (level != 1 && !is_hc_level(level)) return fail(c, PLZ4HIP_E_UNSUPPORTED, "levels 1..12 are built");
    DictJob j; j.dict = dict; j.any = true; j.level = level;
    return host_codec(c, 0, nBlocks, src, srcLen, dst, dstCap, 0, 0, result, nullptr, &j);
}

int plz4hip_decompress_batch_dict(plz4hip_ctx* c, int nBlocks, const void* const* src, const int32_t* srcLen,
                                  void* const* dst, const int32_t* dstCap, const plz4hip_dict* dict, int32_t* result)
{
    if (!c || !dict || nBlocks < 0 || (nBlocks && (!src || !srcLen || !dst || !dstCap || !result))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_decompress_batch_dict: bad argument");
    DictJob j; j.dict = dict; j.any = true;
    return host_codec(c, 1, nBlocks, src, srcLen, dst, dstCap, 0, 0, result, nullptr, &j);
}

int plz4hip_encode_records_ex(plz4hip_ctx* c, int nBlocks, const void* const* src, const int32_t* srcLen,
                              int bsz, int level, int blockChecksum, int linked, const plz4hip_dict* dict,
                              const void* prevTail, int prevTailLen, void* const* rec, int32_t* recLen)
{
    if (!c || nBlocks < 0 || bsz <= 0 || (nBlocks && (!src || !srcLen || !rec || !recLen))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_encode_records_ex: bad argument");
    if (level != 1 && !is_hc_level(level)) return fail(c, PLZ4HIP_E_UNSUPPORTED, "levels 1..12 are built");
    for (int i = 0; i < nBlocks; ++i) if (srcLen[i] > bsz) return fail(c, PLZ4HIP_E_ARG, "source block larger than block size");
    if (!linked && !dict) {
        if (is_hc_level(level)) { DictJob j; j.level = level; return host_codec(c, 2, nBlocks, src, srcLen, rec, nullptr, bsz, blockChecksum, recLen, nullptr, &j); }
        return host_codec(c, 2, nBlocks, src, srcLen, rec, nullptr, bsz, blockChecksum, recLen, nullptr);
    }
    DictJob j; j.dict = dict; j.linked = linked; j.prevTail = prevTail; j.prevTailLen = (linked && prevTail) ? prevTailLen : -1; j.any = true; j.level = level;
    return host_codec(c, 2, nBlocks, src, srcLen, rec, nullptr, bsz, blockChecksum, recLen, nullptr, &j);
}

int plz4hip_decode_records_ex(plz4hip_ctx* c, int nBlocks, const void* const* rec, const int32_t* recLen,
                              int bsz, int blockChecksum, int linked, const plz4hip_dict* dict,
                              void* window, int* windowLen, void* const* dst, int32_t* result, int32_t* status)
{
    if (!c || nBlocks < 0 || bsz <= 0 || (nBlocks && (!rec || !recLen || !dst || !result || !status))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_decode_records_ex: bad argument");
    for (int i = 0; i < nBlocks; ++i) if (recLen[i] < 4) return fail(c, PLZ4HIP_E_ARG, "record shorter than its size word");
    if (!linked && !dict) return host_codec(c, 3, nBlocks, rec, recLen, dst, nullptr, bsz, blockChecksum, result, status);
    if (linked && (!window || !windowLen || *windowLen < 0 || *windowLen > 65536)) return fail(c, PLZ4HIP_E_ARG, "linked decode needs the 64 KiB window state");
    DictJob j; j.dict = dict; j.linked = linked; j.any = true;
    if (linked) { j.window = (uint8_t*)window; j.windowLen = windowLen; j.dict = nullptr; }
    return host_codec(c, 3, nBlocks, rec, recLen, dst, nullptr, bsz, blockChecksum, result, status, &j);
}

int plz4hip_decode_records_chains(plz4hip_ctx* c, int nChains, const int32_t* chainFirst, const void* const* rec, const int32_t* recLen,
                                  int bsz, int blockChecksum, void* windows, int32_t* windowLen,
                                  void* const* dst, int32_t* result, int32_t* status)
{
    if (!c || nChains < 0 || bsz <= 0 || (nChains && (!chainFirst || !windows || !windowLen))) return fail(c, PLZ4HIP_E_ARG, "plz4hip_decode_records_chains: bad argument");
    if (nChains == 0) return PLZ4HIP_OK;
    if (chainFirst[0] != 0) return fail(c, PLZ4HIP_E_ARG, "chainFirst[0] must be 0");
    for (int k = 0; k < nChains; ++k) {
        if (chainFirst[k + 1] < chainFirst[k]) return fail(c, PLZ4HIP_E_ARG, "chainFirst must not decrease");
        if (windowLen[k] < 0 || windowLen[k] > 65536) return fail(c, PLZ4HIP_E_ARG, "window length out of range");
    }
    const int nBlocks = chainFirst[nChains];
    if (nBlocks && (!rec || !recLen || !dst || !result || !status)) return fail(c, PLZ4HIP_E_ARG, "plz4hip_decode_records_chains: bad argument");
    if (nBlocks == 0) return PLZ4HIP_OK;
    for (int i = 0; i < nBlocks; ++i) if (recLen[i] < 4) return fail(c, PLZ4HIP_E_ARG, "record shorter than its size word");
    DictJob j; j.linked = 1; j.any = true; j.window = (uint8_t*)windows; j.windowLen = windowLen; j.nChains = nChains; j.chainFirst = chainFirst;
    return host_codec(c, 3, nBlocks, rec, recLen, dst, nullptr, bsz, blockChecksum, result, status, &j);
}

}  // extern "C"
