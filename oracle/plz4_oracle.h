/*
 * plz4_oracle.h -- CPU restatement of plz4's per-block hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped path (plz4_amd/libplz4hip.so) never links,
 * includes or calls anything in this directory.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py) against
 *   (a) the reference's own KATs ("hello" frame, `theWorks` frame, header KATs,
 *       content-hash KATs) and
 *   (b) outputs of the real reference (liblz4 v1.10.0 as vendored by plz4, compiled unmodified into
 *       oracle/_ref/liblz4ref.so by oracle/Makefile) over seeded corpora, plus golden fixtures
 *       generated from it and committed under tests/golden/.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef PLZ4_ORACLE_H
#define PLZ4_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- xxHash32, seed 0 (internal/pkg/xxh32/xxh32zero.go:238-280, :58-86, :204-235) ---- */
uint32_t orc_xxh32(const uint8_t* p, size_t n);

typedef struct {
    uint32_t acc[4];
    uint64_t total;
    uint8_t  buf[16];
    uint32_t fill;
} orc_xxh32_state;
void     orc_xxh32_reset (orc_xxh32_state* s);
void     orc_xxh32_update(orc_xxh32_state* s, const uint8_t* p, size_t n);
uint32_t orc_xxh32_digest(const orc_xxh32_state* s);

/* ---- LZ4 block, level 1 (internal/pkg/clz4/lz4.c) ---- */
int orc_compress_bound(int n);                                             /* lz4.h:215 */

/* LZ4_compress_fast(src,dst,n,cap,accel=1): lz4.c:1453 -> :1382 -> :930.  0 = failure. */
int orc_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap);

/* LZ4_decompress_safe: lz4.c:2451 -> :2022.  <0 = -(error position)-1. */
int orc_decompress_safe(const uint8_t* src, int n, uint8_t* dst, int cap);

/* LZ4_decompress_safe_usingDict with a dictionary that is NOT address-adjacent to dst
 * (lz4.c:2719-2732 -> forceExtDict :2523-2531). dictLen==0 falls back to orc_decompress_safe. */
int orc_decompress_safe_dict(const uint8_t* src, int n, uint8_t* dst, int cap,
                             const uint8_t* dict, int dictLen);

/* Streaming encoder state == LZ4_stream_t_internal (lz4.h:715-723). */
typedef struct orc_stream {
    uint32_t                 table[4096];
    const uint8_t*           dictionary;
    const struct orc_stream* dictCtx;
    uint32_t                 currentOffset;
    uint32_t                 tableType;      /* 0 cleared, 2 byU32, 3 byU16 (lz4.c:717) */
    uint32_t                 dictSize;
} orc_stream;

void orc_stream_init        (orc_stream* s);                               /* LZ4_initStream  lz4.c:1552 */
void orc_stream_reset_fast  (orc_stream* s);                               /* LZ4_resetStream_fast :1570 */
int  orc_stream_load_dict   (orc_stream* s, const uint8_t* d, int n, int slow); /* LZ4_loadDict[Slow] :1587-1656 */
void orc_stream_attach      (orc_stream* s, const orc_stream* dictStream); /* LZ4_attach_dictionary :1658 */
int  orc_stream_compress    (orc_stream* s, const uint8_t* src, int n, uint8_t* dst, int cap); /* LZ4_compress_fast_continue :1707 */

/* ---- plz4 glue: one block record (internal/pkg/blk/blk.go:69-109) ----
 * Produces [LE32 size|stored-flag][payload][LE32 xxh32(payload)?] for ONE source block at level 1,
 * independent, no dictionary: encoder capacity == bsz (NOT the bound), stored raw iff the encoder
 * returns 0.  `rec` must hold bsz+8 bytes.  Returns record length. */
int orc_block_record(const uint8_t* src, int n, int bsz, int blockChecksum, uint8_t* rec);

/* Frame header (internal/pkg/header/write.go:23-73).  bsIdx 4..7.  contentSize/dictId are
 * optional (has_* flags).  `out` must hold 19 bytes.  Returns header length (7..19). */
int orc_frame_header(uint8_t* out, int bsIdx, int linked, int blockChecksum, int contentChecksum,
                     int hasContentSize, uint64_t contentSize, int hasDictId, uint32_t dictId);

/* Whole frame as the *sync* writer emits it for one Write(src) + Close()
 * (internal/pkg/sync/writer.go:52-122,133-167,265-290): header, one record per bsz bytes (last one short),
 * end mark, optional content hash.  Level 1, independent blocks.  Returns bytes written or -1 if outCap is too small. */
int64_t orc_frame_encode(const uint8_t* src, int64_t n, int bsIdx, int blockChecksum, int contentChecksum,
                         uint8_t* out, int64_t outCap);

/* Frame decode (internal/pkg/header/read.go:26-119, blk/frame.go:54-127, sync/reader.go:49-87,
 * blk/blk.go:50-61: decoder capacity is bsz+8).  Independent blocks, no dictionary.
 * Returns plaintext length, or a negative orc_err code. */
enum {
    ORC_ERR_HEADER_READ = -1, ORC_ERR_MAGIC = -2, ORC_ERR_VERSION = -3, ORC_ERR_RESERVED_BIT = -4,
    ORC_ERR_BLOCK_DESCRIPTOR = -5, ORC_ERR_HEADER_HASH = -6, ORC_ERR_BLOCK_SIZE_READ = -7,
    ORC_ERR_BLOCK_SIZE_OVERFLOW = -8, ORC_ERR_BLOCK_READ = -9, ORC_ERR_BLOCK_HASH = -10,
    ORC_ERR_DECOMPRESS = -11, ORC_ERR_CONTENT_HASH_READ = -12, ORC_ERR_CONTENT_HASH = -13,
    ORC_ERR_DST_TOO_SMALL = -14, ORC_ERR_UNSUPPORTED = -15
};
int64_t orc_frame_decode(const uint8_t* frame, int64_t n, uint8_t* out, int64_t outCap);

#ifdef __cplusplus
}
#endif
#endif
