/*
 * plz4hip.h -- C ABI of the MI355X (gfx950) LZ4 block engine that replaces plz4's cgo -> liblz4 engine.
 *
 * Drop-in boundary.  Today plz4 crosses from Go into C once per block through
 * internal/pkg/clz4/clz4.go (CompressFast :31-45, DecompressSafe :47-60, CompressBound :27-29) underneath
 * compress.Compressor / compress.Decompressor (internal/pkg/compress/compress.go:7-13, decompress.go:14-16),
 * called from blk.CompressToBlk (internal/pkg/blk/blk.go:69-109), BlkT.Decompress (blk.go:50-61) and the
 * raw block API (plz4_block.go:96-172); the per-block checksum is xxh32.ChecksumZero (blk.go:98-102,
 * blk/frame.go:114-127).  One cgo call per 4 MiB block cannot feed a GPU, so this ABI is the same contract
 * in BATCH form and sits where the reference's worker loops sit (async/writer.go:232-282 compressLoop,
 * async/reader.go:192-221 _decompressLoop): N independent blocks in, N per-block results out.
 * A one-block batch behaves exactly like the per-block call it replaces.
 *
 * Value semantics are liblz4's: encode result > 0 = bytes written, 0 = "cannot compress within dst"
 * (clz4.ErrLz4Compress -> zerr.ErrCompress -> the caller stores the block raw, blk.go:78-92);
 * decode result >= 0 = bytes, < 0 = liblz4's error code (clz4.ErrLz4Decompress -> zerr.ErrCorrupted).
 * The function return value is separate and reports engine/runtime failures (PLZ4HIP_E_*); those must NOT be
 * turned into stored blocks by the caller.
 *
 * Plain C, pointers and sizes only.  Host-pointer entry points never retain caller memory after return.
 * Thread-safety: a ctx serialises its own calls with an internal mutex; every entry point runs on the ctx's device and
 * puts the calling thread's current HIP device back before it returns (goroutines migrate between OS threads); use one
 * ctx per writer/reader for concurrency, one per device for several GPUs (section D).  plz4hip_last_error returns the
 * calling thread's own copy of the text (valid until that thread asks again).
 * Memory: the host-buffer calls keep their pinned host + device staging (up to 3 chunks of <= 4 GiB; one of <= 12 GiB at the HC
 * levels, which run a chunk at a time), level 1 its sequence records (2.25 bytes per input byte of the largest group of blocks
 * in one call, at most half of what is free: PLZ4HIP_L1_BUDGET_GIB; a second such workspace when calls arrive on a second stream
 * while the first one is busy, so that the two need not wait for each other -- PLZ4HIP_L1_WORKSPACES=1: never) and the HC levels
 * their workspaces between calls;
 * plz4hip_ctx_trim gives them back.  The HC workspaces are sized by the call: on independent blocks levels 3..11 keep 14.5 bytes
 * per input byte of the blocks in flight (chain, per-hash lists, sequence records of the segments), level 12 22.5 (the search results as well), level 2 2.25, out
 * of a quarter of the device memory that is free when the call arrives, at most 64 GiB; a call whose blocks do not fit runs in
 * groups.  A caller that owns the GPU raises the budget with PLZ4HIP_HC_BUDGET_GIB (these kernels live on blocks in flight).
 * With a dictionary and / or linked blocks the HC levels run the same designs over segment + block (strides for 4 MiB + 64 KiB).
 * A DECODE call of few blocks (decompress_batch, decode_records and their dev_ forms, up to PLZ4HIP_DX_MAX_BLOCKS = 128 blocks of up
 * to 4 MiB + 8 of output; 0 turns it off) is cut across the whole chip and keeps 8 bytes per input byte + 4 per output byte of
 * the call's blocks (about 50 MiB per 4 MiB block) until plz4hip_ctx_trim; results and error codes are LZ4_decompress_safe's
 * either way (a block that path will not answer for is decoded by the one-wavefront decoder inside the call).
 * Other environment switches, for tests and experiments only: PLZ4HIP_HC_EXT_OFF (the one-thread HC parsers for dictionary / linked
 * calls, rounds 1-3), PLZ4HIP_HC12_LAZY (level 12 with its searches made on demand), PLZ4HIP_HC_OVERLAP_OFF / _MIN / _GROUPS (levels 3..11: a call of 2048
 * blocks or more runs in four or more groups, the list builder of the next group on a second stream of the ctx beside the walk
 * of the current one), PLZ4HIP_DUPLEX=P,D / PLZ4HIP_DUPLEX_PRIO (parser + decoder waves per workgroup of the duplex call),
 * PLZ4HIP_HC_SEGS / PLZ4HIP_HC_MIN_SEG (segments a block is walked
 * in at levels 3..11, default up to 512 of at least 8 KiB), PLZ4HIP_HC_LAZY_OFF / PLZ4HIP_HC_MID_OFF / PLZ4HIP_HC12_OFF / PLZ4HIP_L1_FUSED
 * (the one-kernel paths of rounds 1-2 instead), PLZ4HIP_HOST_CHUNK_MB, PLZ4HIP_VERBOSE.
 */
#ifndef PLZ4HIP_H
#define PLZ4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLZ4HIP_ABI_VERSION 2

enum {
    PLZ4HIP_OK            =  0,
    PLZ4HIP_E_ARG         = -1,   /* bad argument */
    PLZ4HIP_E_DEVICE      = -2,   /* HIP runtime / kernel failure (see plz4hip_last_error) */
    PLZ4HIP_E_NOMEM       = -3,   /* host or device allocation failed */
    PLZ4HIP_E_UNSUPPORTED = -4    /* level / mode not built yet */
};

/* Per-block status of the record-level decode (plz4hip_decode_records, plz4hip_dev_decode_records). */
enum {
    PLZ4HIP_BLK_OK            = 0,
    PLZ4HIP_BLK_HASH_MISMATCH = 1,   /* zerr.ErrBlockHash        (blk/frame.go:114-127) */
    PLZ4HIP_BLK_SIZE_OVERFLOW = 2,   /* zerr.ErrBlockSizeOverflow (blk/frame.go:79-81)  */
    PLZ4HIP_BLK_CORRUPT       = 3    /* zerr.ErrDecompress: liblz4 returned < 0 (compress/decompress.go:32-38) */
};

typedef struct plz4hip_ctx plz4hip_ctx;

int         plz4hip_abi_version(void);
int         plz4hip_device_count(void);                       /* <0: PLZ4HIP_E_DEVICE */
int         plz4hip_ctx_create(int device, plz4hip_ctx** out);
void        plz4hip_ctx_destroy(plz4hip_ctx* ctx);
const char* plz4hip_last_error(const plz4hip_ctx* ctx);       /* text of the last PLZ4HIP_E_* on this ctx */
int         plz4hip_ctx_trim(plz4hip_ctx* ctx);                /* release staging buffers and HC workspaces (waits for work in flight) */

/* == clz4.CompressBound (clz4.go:27-29) -> LZ4_compressBound (lz4.h:215).  Pure host arithmetic. */
int plz4hip_compress_bound(int n);

/* ---------------------------------------------------------------------------------------------------------
 * A. Batched Compressor / Decompressor over HOST buffers (what the cgo shim calls).
 *    Replaces clz4.CompressFast (clz4.go:31-45) / clz4.DecompressSafe (clz4.go:47-60) x nBlocks.
 *    level: 1 (== LZ4_compress_fast(accel 1), compress/indie.go:66-74) and 2..12 (== LZ4_compress_HC(level),
 *           compress/indie.go:80-88 -> clz4.CompressHC clz4.go:80-94 -> lz4hc.c:1519: lz4mid at 2, hash chain at 3..9,
 *           optimal parser at 10..12); anything else is PLZ4HIP_E_UNSUPPORTED.  encode_records / dev_encode_records
 *           take the same levels.
 *    result[i]: encode  > 0 bytes written into dst[i], 0 = liblz4 "does not fit dstCap[i]";
 *               decode >= 0 bytes written, < 0 liblz4 error code.
 * ------------------------------------------------------------------------------------------------------- */
int plz4hip_compress_batch(plz4hip_ctx* ctx, int nBlocks,
                           const void* const* src, const int32_t* srcLen,
                           void* const* dst, const int32_t* dstCap,
                           int level, int32_t* result);

int plz4hip_decompress_batch(plz4hip_ctx* ctx, int nBlocks,
                             const void* const* src, const int32_t* srcLen,
                             void* const* dst, const int32_t* dstCap,
                             int32_t* result);

/* == xxh32.ChecksumZero (xxh32zero.go:238-280) x n. */
int plz4hip_xxh32_batch(plz4hip_ctx* ctx, int n, const void* const* buf, const int32_t* len, uint32_t* out);

/* ---------------------------------------------------------------------------------------------------------
 * B. Frame records over HOST buffers: blk.CompressToBlk (blk.go:69-109) / FrameReader._read + BlkT.Decompress
 *    (blk/frame.go:54-127, blk.go:50-61) fused on the device.
 *    encode: rec[i] (>= bsz+8 bytes) receives [LE32 size | 0x80000000 if stored][payload][LE32 xxh32(payload)?];
 *            encoder capacity is bsz (NOT the bound); stored raw iff the encoder returns 0.  recLen[i] = bytes.
 *    decode: rec[i]/recLen[i] is one such record (size word included); dst[i] has bsz+8 bytes like the
 *            reference's pooled block (blk/pool.go:23-26); status[i] = PLZ4HIP_BLK_*, result[i] = plaintext bytes
 *            (or liblz4's negative code when status is PLZ4HIP_BLK_CORRUPT).
 * ------------------------------------------------------------------------------------------------------- */
int plz4hip_encode_records(plz4hip_ctx* ctx, int nBlocks,
                           const void* const* src, const int32_t* srcLen,
                           int bsz, int level, int blockChecksum,
                           void* const* rec, int32_t* recLen);

int plz4hip_decode_records(plz4hip_ctx* ctx, int nBlocks,
                           const void* const* rec, const int32_t* recLen,
                           int bsz, int blockChecksum,
                           void* const* dst, int32_t* result, int32_t* status);

/* ---------------------------------------------------------------------------------------------------------
 * B'. Dictionaries and linked blocks (levels 1..12; SURVEY.md §8a-11, BASELINE config 5).
 *    plz4hip_dict == clz4.DictCtx + clz4.DictCtxHC (clz4.go:96-147: private copy of the last 64 KiB, the LZ4_loadDictSlow
 *                   table for level 1, the LZ4_loadDictHC tables for level 2 (lz4mid) and for levels 3..12 (hash chain)).
 *    *_batch_dict : level 1: clz4.StreamIndieCtx.Compress (clz4.go:160-179); levels 2..12: clz4.StreamCtxHC.Compress
 *                   (clz4.go:191-209 -> LZ4_compress_HC_continue under an attached dictionary, lz4hc.c:1438-1461);
 *                   clz4.DecompressSafeWithDict (clz4.go:62-78) per block -- i.e. CompressBlock/DecompressBlock with
 *                   WithBlockDictionary (plz4_block.go:48-53).
 *    encode_records_ex: `dict` = WithDictionary; `linked` = WithBlockLinked: block i>0 is primed with the last <= 64 KiB of
 *                   src[i-1] (async/writer.go:412-437 -> LZ4_loadDict, clz4.go:224-241; levels 2..12: LZ4_loadDictHC,
 *                   clz4.go:262-283); block 0 with prevTail when the batch continues a frame (prevTail == NULL: block 0
 *                   starts the frame and uses `dict`, if any).  As everywhere, a block the encoder cannot fit into bsz
 *                   bytes comes back as a stored record; NOTE that the reference's linked HC compressor does not: it
 *                   fails the write instead (compress/linked.go:47-49 returns the error without zerr.ErrCompress, so
 *                   blk.CompressToBlk takes no fallback) -- a drop-in caller checks the stored bit of rec[i] for
 *                   linked && level > 1 and reports the error (the host layer and the Go shim do).
 *    decode_records_ex: independent blocks + dict: every block against `dict` (compress/decompress.go:42-58);
 *                   linked: a serial chain over the batch; `window` (64 KiB, caller-owned) / `*windowLen` carry
 *                   compress.DictT (compress/dict.go:5-56) across calls: initialise with the dictionary's last 64 KiB
 *                   (or length 0).  As in the reference, a stored block does not update the window.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct plz4hip_dict plz4hip_dict;
int  plz4hip_dict_create(plz4hip_ctx* ctx, const void* dict, int dictLen, plz4hip_dict** out);
void plz4hip_dict_destroy(plz4hip_ctx* ctx, plz4hip_dict* dict);

int plz4hip_compress_batch_dict(plz4hip_ctx* ctx, int nBlocks, const void* const* src, const int32_t* srcLen,
                                void* const* dst, const int32_t* dstCap, int level, const plz4hip_dict* dict, int32_t* result);
int plz4hip_decompress_batch_dict(plz4hip_ctx* ctx, int nBlocks, const void* const* src, const int32_t* srcLen,
                                  void* const* dst, const int32_t* dstCap, const plz4hip_dict* dict, int32_t* result);

int plz4hip_encode_records_ex(plz4hip_ctx* ctx, int nBlocks, const void* const* src, const int32_t* srcLen,
                              int bsz, int level, int blockChecksum, int linked, const plz4hip_dict* dict,
                              const void* prevTail, int prevTailLen, void* const* rec, int32_t* recLen);
int plz4hip_decode_records_ex(plz4hip_ctx* ctx, int nBlocks, const void* const* rec, const int32_t* recLen,
                              int bsz, int blockChecksum, int linked, const plz4hip_dict* dict,
                              void* window, int* windowLen, void* const* dst, int32_t* result, int32_t* status);
/* Linked decode of SEVERAL frames in one call.  The blocks of one linked frame are a serial chain ("replicas
 * only"), chains of different frames share nothing: chain k = records [chainFirst[k], chainFirst[k+1]) (chainFirst[0] == 0,
 * nChains + 1 entries), one wavefront each.  windows = nChains x 64 KiB, windowLen[nChains]: the compress.DictT state of every
 * chain, in/out, exactly as `window` / `windowLen` of plz4hip_decode_records_ex(linked = 1); per-block result / status likewise
 * (a chain stops at its first bad block, the other chains go on).  No counterpart in the reference: a server that decodes
 * many linked frames (rdr.go:339-341 runs each one on a single goroutine) hands them over together. */
int plz4hip_decode_records_chains(plz4hip_ctx* ctx, int nChains, const int32_t* chainFirst,
                                  const void* const* rec, const int32_t* recLen, int bsz, int blockChecksum,
                                  void* windows, int32_t* windowLen, void* const* dst, int32_t* result, int32_t* status);

/* ---------------------------------------------------------------------------------------------------------
 * B''. Streaming content checksum on the device: xxh32.XXHZero (internal/pkg/xxh32/xxh32zero.go:58-86 Write, :204-235 Sum32)
 *    as the reference feeds it, block after block (async/hash.go:99-111, sync/writer.go:267-271, sync/reader.go:56).
 *    XXH32 has no combine operator: the stream is one serial chain, so this is ONE wavefront that runs beside the codec kernels
 *    (it is there to take the hash off the host's critical path, not to be fast: see DESIGN.md for its rate).
 *    create / reset: empty stream, seed 0.  sum: Sum32 of what was written so far (the stream goes on after it).
 *    update (host bytes) / dev_update (device bytes, enqueued on `stream` behind the stream's previous update).
 *    plz4hip_ctx_set_content_hash(ctx, h): while set (NULL clears it), every plz4hip_encode_records[_ex] call also writes the
 *    plaintext of its blocks, in block order, into h, and every plz4hip_decode_records[_ex] call the plaintext it produced
 *    (blocks whose result is <= 0 contribute nothing; the caller stops at the first bad block anyway) -- without the
 *    plaintext passing through the host again.  A stream is fed from one place at a time.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct plz4hip_xxh32_stream plz4hip_xxh32_stream;
int  plz4hip_xxh32_stream_create(plz4hip_ctx* ctx, plz4hip_xxh32_stream** out);
void plz4hip_xxh32_stream_destroy(plz4hip_ctx* ctx, plz4hip_xxh32_stream* h);
int  plz4hip_xxh32_stream_reset(plz4hip_ctx* ctx, plz4hip_xxh32_stream* h);
int  plz4hip_xxh32_stream_update(plz4hip_ctx* ctx, plz4hip_xxh32_stream* h, const void* data, int64_t n);
int  plz4hip_dev_xxh32_stream_update(plz4hip_ctx* ctx, plz4hip_xxh32_stream* h, const void* devData, int64_t n, void* stream);
int  plz4hip_xxh32_stream_sum(plz4hip_ctx* ctx, plz4hip_xxh32_stream* h, uint32_t* out);
int  plz4hip_ctx_set_content_hash(plz4hip_ctx* ctx, plz4hip_xxh32_stream* h);

/* ---------------------------------------------------------------------------------------------------------
 * C. Device-resident pipeline (bench.py, GPU-to-GPU producers).  Every pointer below is a DEVICE pointer
 *    on ctx's device; work is enqueued on `stream` (a hipStream_t; NULL is HIP's default/null stream, as for any
 *    hip*Async call) and the call returns without synchronising.  Layout:
 *      src     plaintext, block i = src[i*bsz, min((i+1)*bsz, srcBytes))          nBlocks = ceil(srcBytes/bsz)
 *      stage   nBlocks records at stride plz4hip_dev_stage_stride(bsz)             (scratch)
 *      recLen  int32[nBlocks]   record i length (4 + payload + 4 if checksums)
 *      recOff  int64[nBlocks+1] exclusive prefix sum of recLen (recOff[nBlocks] = body bytes)
 *      body    the records, compacted back to back: exactly the frame's block section
 *    plz4hip_dev_encode_records  : the encode kernel; fills stage + recLen.
 *    plz4hip_dev_compact_records : exclusive scan of recLen -> recOff, then moves every record to body+recOff[i];
 *                                  records that would end past bodyCap are skipped (recOff[nBlocks] > bodyCap tells).
 *    plz4hip_dev_scatter_records : generic record mover, dst+dstOff[k] <- src+srcOff[k] (len[k] bytes); used by the
 *                                  multi-GPU frame assembly, where the owner rank interleaves the bodies it
 *                                  received from the other ranks (block i lives on rank i mod G).
 *    plz4hip_dev_decode_records reads records at body+recOff[i] and writes block i at dst + i*dstStride
 *    (capacity dstCap, bsz+8 in the reference); result/status as in B.
 * ------------------------------------------------------------------------------------------------------- */
int64_t plz4hip_dev_stage_stride(int bsz);

int plz4hip_dev_encode_records(plz4hip_ctx* ctx, const void* src, int64_t srcBytes, int bsz, int level,
                               int blockChecksum, void* stage, int32_t* recLen, void* stream);

int plz4hip_dev_compact_records(plz4hip_ctx* ctx, const void* stage, int64_t stageStride, const int32_t* recLen,
                                int nBlocks, int64_t* recOff, void* body, int64_t bodyCap, void* stream);

int plz4hip_dev_scatter_records(plz4hip_ctx* ctx, const void* src, const int64_t* srcOff, const int32_t* len,
                                const int64_t* dstOff, int n, int maxLen, void* dst, int64_t dstCap, void* stream);

int plz4hip_dev_decode_records(plz4hip_ctx* ctx, const void* body, const int64_t* recOff, int nBlocks,
                               int bsz, int blockChecksum, void* dst, int64_t dstStride, int dstCap,
                               int32_t* result, int32_t* status, void* stream);

/* plz4hip_dev_duplex_records: plz4hip_dev_encode_records (level 1) of one batch AND plz4hip_dev_decode_records of another --
 * a writer's next batch and a reader's -- enqueued as one call, with results identical to the two calls made one after the
 * other; the operands of the two sides must not overlap.  The reference runs its compress and decompress workers side by side
 * on the host's cores (async/writer.go:232-282, async/reader.go:192-221); here the decoder's waves share every CU with the
 * level-1 parser's (k_l1_duplex): the parser is a serial chain per block that leaves about half of the vector issue slots
 * idle, the decoder is bound by those slots, so the decode costs the pair little more than the parse alone.  Either side may be
 * empty (srcBytes == 0 / nDecBlocks == 0). */
int plz4hip_dev_duplex_records(plz4hip_ctx* ctx, const void* src, int64_t srcBytes, int bsz, int blockChecksum, void* stage, int32_t* recLen,
                               const void* body, const int64_t* recOff, int nDecBlocks, int decBsz, int decBlockChecksum,
                               void* dst, int64_t dstStride, int dstCap, int32_t* result, int32_t* status, void* stream);

/* plz4hip_dev_encode_body: plz4hip_dev_encode_records + plz4hip_dev_compact_records in one call, without the staging area and
 * without the second pass over the records -- blk.CompressToBlk (blk/blk.go:69-109) for every block and the writer's in-order
 * emission (async/writer.go:284-381) as ONE contiguous frame body: record i lands at body + recOff[i], recOff[nBlocks] = the body's
 * length, recLen[i] = record i's bytes (a negative PLZ4HIP_E_* for a block the engine failed on; it takes no room).  The emit stage
 * knows every record's length before it writes a byte, so the records' places come from a scan between its passes.  Levels 1
 * and 2, blocks up to 4 MiB (the staged call; PLZ4HIP_E_NOMEM if its workspace cannot be had).  recOff[nBlocks] > bodyCap: the
 * records beyond the room were not written.  plz4hip_dev_duplex_body: the same with the record decode of another batch in the
 * same launch, as plz4hip_dev_duplex_records. */
int plz4hip_dev_encode_body(plz4hip_ctx* ctx, const void* src, int64_t srcBytes, int bsz, int level, int blockChecksum,
                            void* body, int64_t bodyCap, int64_t* recOff, int32_t* recLen, void* stream);
int plz4hip_dev_duplex_body(plz4hip_ctx* ctx, const void* src, int64_t srcBytes, int bsz, int blockChecksum,
                            void* body, int64_t bodyCap, int64_t* recOff, int32_t* recLen,
                            const void* decBody, const int64_t* decRecOff, int nDecBlocks, int decBsz, int decBlockChecksum,
                            void* dst, int64_t dstStride, int dstCap, int32_t* result, int32_t* status, void* stream);

/* Raw LZ4 blocks on the device (no record framing): block i = src + i*srcStride (srcLen[i] bytes) ->
 * dst + i*dstStride (capacity dstCap[i]); result[i] as in A, levels 1..12.  srcLen/dstCap/result are device arrays.
 * maxLen is a host value the per-block workspaces are sized from, at EVERY level: it must be >= every srcLen[i]; a block whose
 * device-side length is outside [0, maxLen] is not touched and gets result[i] = PLZ4HIP_E_ARG (its neighbours are unaffected).
 * maxLen <= 0 means "unknown to the host": allowed at level 1 only, where it selects the one-kernel encoder that needs no
 * workspace (levels 2..12 return PLZ4HIP_E_ARG for the call).  A level-1 caller with a loose or stale positive maxLen therefore
 * gets per-block errors, not slower output: pass 0 rather than a guess.
 * The staged level-1 call and the HC levels (here and in plz4hip_dev_encode_records) work in per-ctx workspaces and read a
 * per-ctx sanitised copy of srcLen: jobs enqueued on different streams of one ctx are ordered behind each other on the device
 * (an event wait, no host block); they never overlap.  The exception is the staged level-1 call on device-resident records
 * (plz4hip_dev_encode_records / _encode_body / _duplex_records / _duplex_body): a ctx keeps two record workspaces, so calls that
 * alternate over two streams overlap -- behind a call that fills the chip (>= 8 blocks per CU) the parse launch of the later call
 * waits on the device until the earlier one has handed out its last block, then moves in as that one's workgroups leave, and the
 * earlier call's emit kernels run beside it (bench.py --pipelines: + 10 % over one stream).
 * The library never issues work on the NULL stream (its own small copies and memsets run on a stream of the ctx): a process that
 * does not use the NULL stream itself keeps all four hardware queues for the streams it passes in and the staging slots. */
int plz4hip_dev_compress(plz4hip_ctx* ctx, int nBlocks, const void* src, int64_t srcStride, const int32_t* srcLen,
                         void* dst, int64_t dstStride, const int32_t* dstCap, int level, int maxLen, int32_t* result, void* stream);
int plz4hip_dev_decompress(plz4hip_ctx* ctx, int nBlocks, const void* src, int64_t srcStride, const int32_t* srcLen,
                           void* dst, int64_t dstStride, const int32_t* dstCap, int32_t* result, void* stream);

/* Number of waves the encode / decode kernels keep resident on ctx's device (for sizing batches). */
int plz4hip_dev_resident_waves(plz4hip_ctx* ctx, int decode);

/* ---------------------------------------------------------------------------------------------------------
 * D. Several GPUs of one node behind one handle (BASELINE.json north_star: independent blocks shard round-robin).
 *    The reference spreads the blocks of a stream over NParallel worker goroutines and emits their records in order
 *    (async/writer.go:232-282 compressLoop, :284-381 writeLoop, :439-467 kickoffAsync); here the workers are the GPUs:
 *    block i of a call belongs to device i mod G, G = the number of entries passed to plz4hip_mgpu_create (one plz4hip_ctx
 *    each; the same device may be listed more than once).  Results come back in block order; value semantics as in A / B.
 *    Host buffers (what the cgo shim binds): one host thread per device drives that device's ctx.
 *    Device-resident shards (GPU producers, bench): shard k is ONE buffer on device k holding blocks k, k+G, ... back to back.
 *      dev_encode_frame: every device encodes + compacts its shard; record sizes go to the host, one prefix sum gives the
 *        frame offsets (recOff, host, nBlocks+1 entries), and the frame body is assembled on device `owner` (an index into
 *        the handle's devices): the other devices' records arrive peer to peer in pieces of <= 256 MiB through two scratch
 *        buffers and are moved to their place while the next piece travels -- the owner holds the body + two pieces.
 *      dev_decode_frame: the reverse: records are dealt from `owner` to the devices, each decodes its shard into shardDst[k]
 *        (block j of shard k at shardDst[k] + j*dstStride).
 *    Linked decode does not shard (one serial chain per frame): use plz4hip_decode_records_chains per device.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct plz4hip_mgpu plz4hip_mgpu;
int          plz4hip_mgpu_create(const int* devices, int nDevices, plz4hip_mgpu** out);
void         plz4hip_mgpu_destroy(plz4hip_mgpu* m);
int          plz4hip_mgpu_count(const plz4hip_mgpu* m);
plz4hip_ctx* plz4hip_mgpu_ctx(plz4hip_mgpu* m, int k);                /* the k-th device's ctx (dictionaries, trims, ...) */
const char*  plz4hip_mgpu_last_error(const plz4hip_mgpu* m);

int plz4hip_mgpu_compress_batch(plz4hip_mgpu* m, int nBlocks, const void* const* src, const int32_t* srcLen,
                                void* const* dst, const int32_t* dstCap, int level, int32_t* result);
int plz4hip_mgpu_decompress_batch(plz4hip_mgpu* m, int nBlocks, const void* const* src, const int32_t* srcLen,
                                  void* const* dst, const int32_t* dstCap, int32_t* result);
int plz4hip_mgpu_encode_records(plz4hip_mgpu* m, int nBlocks, const void* const* src, const int32_t* srcLen,
                                int bsz, int level, int blockChecksum, void* const* rec, int32_t* recLen);
int plz4hip_mgpu_decode_records(plz4hip_mgpu* m, int nBlocks, const void* const* rec, const int32_t* recLen,
                                int bsz, int blockChecksum, void* const* dst, int32_t* result, int32_t* status);

int plz4hip_mgpu_dev_encode_frame(plz4hip_mgpu* m, int nBlocks, const void* const* shardSrc, const int64_t* shardBytes,
                                  int bsz, int level, int blockChecksum, int owner, void* body, int64_t bodyCap,
                                  int64_t* recOff, int64_t* bodyBytes);
int plz4hip_mgpu_dev_decode_frame(plz4hip_mgpu* m, int nBlocks, int owner, const void* body, const int64_t* recOff,
                                  int bsz, int blockChecksum, void* const* shardDst, int64_t dstStride, int dstCap,
                                  int32_t* result, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* PLZ4HIP_H */
