//go:build cgo && plz4_hip

// Package plz4hip is the cgo binding of the MI355X block engine (include/plz4hip.h, libplz4hip.so).
// It takes the place of internal/pkg/clz4 (clz4.go:27-94) for the level-1, independent-block path.
//
// NOT COMPILED IN THE BUILD IMAGE (no Go toolchain there): this file is the binding a plz4 maintainer adds; every C entry
// point it calls is exercised through the same ABI by tests/test_gpu_parity.py and tests/test_gpu_host.py.
package plz4hip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../plz4_amd -lplz4hip -Wl,-rpath,${SRCDIR}/../../plz4_amd
#include <stdlib.h>
#include "plz4hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"os"
	"runtime"
	"strconv"
	"strings"
	"unsafe"
)

// The HIP runtime deals a process's streams onto four hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and streams that share
// a queue run behind each other (INTEGRATION.md section 3).  A ctx brings up to six streams of its own; the variable is read when
// the runtime initialises, i.e. at the first engine call, so it is set here unless the operator has set it.
func init() {
	if os.Getenv("GPU_MAX_HW_QUEUES") == "" {
		os.Setenv("GPU_MAX_HW_QUEUES", "16")
	}
}

var (
	ErrLz4Compress   = errors.New("lz4 fail compress; insufficient destination buffer") // == clz4.ErrLz4Compress
	ErrLz4Decompress = errors.New("lz4 fail decompress")                                // == clz4.ErrLz4Decompress
	ErrEngine        = errors.New("plz4hip engine failure")                             // never mapped to a stored block
)

// Ctx is one plz4hip_ctx (one device, internal staging, serialised by the library).  borrowed: the ctx belongs to a Multi
// (plz4hip_mgpu_ctx hands out the handle's own pointers): Close leaves it alone, Multi.Close destroys it.
type Ctx struct {
	p        *C.plz4hip_ctx
	borrowed bool
}

func NewCtx(device int) (*Ctx, error) {
	var p *C.plz4hip_ctx
	if rc := C.plz4hip_ctx_create(C.int(device), &p); rc != C.PLZ4HIP_OK {
		return nil, fmt.Errorf("%w: plz4hip_ctx_create rc=%d", ErrEngine, int(rc))
	}
	c := &Ctx{p: p}
	runtime.SetFinalizer(c, func(c *Ctx) { c.Close() })
	return c, nil
}

func (c *Ctx) Close() {
	if c.p != nil && !c.borrowed {
		C.plz4hip_ctx_destroy(c.p)
	}
	c.p = nil
}

// Trim gives back what the ctx keeps between calls: pinned host + device staging, the level-1 and HC workspaces
// (plz4hip_ctx_trim).  The HC levels hold tens of GiB after a large call.
func (c *Ctx) Trim() error {
	if rc := C.plz4hip_ctx_trim(c.p); rc != C.PLZ4HIP_OK {
		return c.engineErr(rc)
	}
	return nil
}

func CompressBound(sz int) int { return int(C.plz4hip_compress_bound(C.int(sz))) } // == clz4.CompressBound

func (c *Ctx) engineErr(rc C.int) error {
	return fmt.Errorf("%w: rc=%d: %s", ErrEngine, int(rc), C.GoString(C.plz4hip_last_error(c.p)))
}

// batch marshals [][]byte into the C pointer/length arrays.  The pointer arrays live in C memory (cgo forbids Go
// pointers to Go pointers); the Go slices are pinned for the duration of the call (Go 1.21 runtime.Pinner).
type batch struct {
	ptrs unsafe.Pointer
	lens []C.int32_t
	pin  runtime.Pinner
}

func newBatch(bufs [][]byte, useCap bool) *batch {
	n := len(bufs)
	b := &batch{ptrs: C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))), lens: make([]C.int32_t, n)}
	pp := unsafe.Slice((*unsafe.Pointer)(b.ptrs), n)
	for i, s := range bufs {
		if useCap {
			s = s[:cap(s)]
		}
		b.lens[i] = C.int32_t(len(s))
		if len(s) > 0 {
			b.pin.Pin(&s[0])
			pp[i] = unsafe.Pointer(&s[0])
		} else {
			pp[i] = nil
		}
	}
	return b
}
// i32p: &v[0], or nil for an empty batch (the C ABI accepts nBlocks == 0; indexing an empty slice would panic).
func i32p(v []C.int32_t) *C.int32_t {
	if len(v) == 0 {
		return nil
	}
	return &v[0]
}

func (b *batch) free() { b.pin.Unpin(); C.free(b.ptrs) }

// CompressBatch == clz4.CompressFast(src[i], dst[i], 1) for every i (clz4.go:31-45).
// n[i] > 0 bytes written, n[i] == 0 => ErrLz4Compress for that block (the caller stores it raw, blk.go:78-92).
func (c *Ctx) CompressBatch(src, dst [][]byte, level int) ([]int, error) {
	s, d := newBatch(src, false), newBatch(dst, false)
	defer s.free()
	defer d.free()
	res := make([]C.int32_t, len(src))
	rc := C.plz4hip_compress_batch(c.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), (*unsafe.Pointer)(d.ptrs), i32p(d.lens), C.int(level), i32p(res))
	if rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	out := make([]int, len(res))
	for i, r := range res {
		out[i] = int(r)
	}
	return out, nil
}

// DecompressBatch == clz4.DecompressSafe(src[i], dst[i]) for every i (clz4.go:47-60): n[i] < 0 is liblz4's code.
func (c *Ctx) DecompressBatch(src, dst [][]byte) ([]int, error) {
	s, d := newBatch(src, false), newBatch(dst, false)
	defer s.free()
	defer d.free()
	res := make([]C.int32_t, len(src))
	rc := C.plz4hip_decompress_batch(c.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), (*unsafe.Pointer)(d.ptrs), i32p(d.lens), i32p(res))
	if rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	out := make([]int, len(res))
	for i, r := range res {
		out[i] = int(r)
	}
	return out, nil
}

// EncodeRecords == blk.CompressToBlk for every block (blk.go:69-109): rec[i] must have cap bsz+8; returns record lengths.
func (c *Ctx) EncodeRecords(src, rec [][]byte, bsz, level int, blockChecksum bool) ([]int, error) {
	s, r := newBatch(src, false), newBatch(rec, true)
	defer s.free()
	defer r.free()
	res := make([]C.int32_t, len(src))
	bc := C.int(0)
	if blockChecksum {
		bc = 1
	}
	rc := C.plz4hip_encode_records(c.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), C.int(bsz), C.int(level), bc, (*unsafe.Pointer)(r.ptrs), i32p(res))
	if rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	out := make([]int, len(res))
	for i, v := range res {
		out[i] = int(v)
	}
	return out, nil
}

// DecodeRecords == FrameReader checks + BlkT.Decompress for every record (blk/frame.go:54-127, blk.go:50-61).
// status[i]: 0 ok, 1 ErrBlockHash, 2 ErrBlockSizeOverflow, 3 ErrDecompress (n[i] then holds liblz4's code).
func (c *Ctx) DecodeRecords(rec, dst [][]byte, bsz int, blockChecksum bool) (n []int, status []int, err error) {
	r, d := newBatch(rec, false), newBatch(dst, true)
	defer r.free()
	defer d.free()
	res := make([]C.int32_t, len(rec))
	st := make([]C.int32_t, len(rec))
	bc := C.int(0)
	if blockChecksum {
		bc = 1
	}
	rc := C.plz4hip_decode_records(c.p, C.int(len(rec)), (*unsafe.Pointer)(r.ptrs), i32p(r.lens), C.int(bsz), bc, (*unsafe.Pointer)(d.ptrs), i32p(res), i32p(st))
	if rc != C.PLZ4HIP_OK {
		return nil, nil, c.engineErr(rc)
	}
	n, status = make([]int, len(res)), make([]int, len(res))
	for i := range res {
		n[i], status[i] = int(res[i]), int(st[i])
	}
	return
}

// ---- dictionaries and linked blocks (include/plz4hip.h section B'; the counterparts of clz4.DictCtx, StreamIndieCtx,
// StreamLinkedCtx and DecompressSafeWithDict, clz4.go:96-248, and of their HC twins DictCtxHC, StreamCtxHC, StreamLinkedCtxHC,
// clz4.go:122-147, :181-209, :250-283).  `level` 1..12 as everywhere else.

// Dict is a dictionary context on the device: the last 64 KiB of the user dictionary, its LZ4_loadDictSlow table (level 1) and
// its LZ4_loadDictHC tables (level 2: lz4mid tables; levels 3..12: hash chain).
type Dict struct {
	c *Ctx
	p *C.plz4hip_dict
}

func (c *Ctx) NewDict(dict []byte) (*Dict, error) {
	var p *C.plz4hip_dict
	var dp unsafe.Pointer
	if len(dict) > 0 {
		dp = unsafe.Pointer(&dict[0])
	}
	if rc := C.plz4hip_dict_create(c.p, dp, C.int(len(dict)), &p); rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	return &Dict{c: c, p: p}, nil
}

func (d *Dict) Close() {
	if d != nil && d.p != nil {
		C.plz4hip_dict_destroy(d.c.p, d.p)
		d.p = nil
	}
}

func (d *Dict) ptr() *C.plz4hip_dict {
	if d == nil {
		return nil
	}
	return d.p
}

// CompressBatchDict == StreamIndieCtx.Compress(src[i], dst[i]) (level 1, clz4.go:160-179) or StreamCtxHC.Compress (levels 2..12,
// clz4.go:191-209) for every i: independent blocks, one shared dictionary.
func (c *Ctx) CompressBatchDict(src, dst [][]byte, level int, d *Dict) ([]int, error) {
	s, o := newBatch(src, false), newBatch(dst, true)
	defer s.free()
	defer o.free()
	res := make([]C.int32_t, len(src))
	rc := C.plz4hip_compress_batch_dict(c.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), (*unsafe.Pointer)(o.ptrs), i32p(o.lens), C.int(level), d.ptr(), i32p(res))
	if rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	out := make([]int, len(res))
	for i, v := range res {
		out[i] = int(v)
	}
	return out, nil
}

// DecompressBatchDict == clz4.DecompressSafeWithDict(src[i], dst[i], dict) for every i.
func (c *Ctx) DecompressBatchDict(src, dst [][]byte, d *Dict) ([]int, error) {
	s, o := newBatch(src, false), newBatch(dst, true)
	defer s.free()
	defer o.free()
	res := make([]C.int32_t, len(src))
	rc := C.plz4hip_decompress_batch_dict(c.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), (*unsafe.Pointer)(o.ptrs), i32p(o.lens), d.ptr(), i32p(res))
	if rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	out := make([]int, len(res))
	for i, v := range res {
		out[i] = int(v)
	}
	return out, nil
}

// EncodeRecordsEx == blk.CompressToBlk through a StreamIndieCtx / StreamCtxHC (dict) or a StreamLinkedCtx / StreamLinkedCtxHC
// (linked), chosen by level as compress.NewCompressorFactory does (compress/compress.go:32-80): block i of a linked
// call is primed with the last 64 KiB of block i-1; prevTail is that window for block 0 when the call continues a frame
// (nil: block 0 starts the frame, with the dictionary if there is one -- async/writer.go:412-437).
func (c *Ctx) EncodeRecordsEx(src, rec [][]byte, bsz, level int, blockChecksum, linked bool, d *Dict, prevTail []byte) ([]int, error) {
	s, r := newBatch(src, false), newBatch(rec, true)
	defer s.free()
	defer r.free()
	res := make([]C.int32_t, len(src))
	var tp unsafe.Pointer
	tl := C.int(-1)
	if prevTail != nil {
		tl = C.int(len(prevTail))
		if len(prevTail) > 0 {
			tp = unsafe.Pointer(&prevTail[0])
		}
	}
	rc := C.plz4hip_encode_records_ex(c.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), C.int(bsz), C.int(level), b2i(blockChecksum), b2i(linked),
		d.ptr(), tp, tl, (*unsafe.Pointer)(r.ptrs), i32p(res))
	if rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	out := make([]int, len(res))
	for i, v := range res {
		out[i] = int(v)
	}
	return out, nil
}

// DecodeRecordsEx: the decode side.  For a linked frame the records of one call form a chain; window (64 KiB, in/out)
// and *windowLen carry the sliding dictionary between calls exactly as compress.DictT does (compress/dict.go:43-86),
// including its rule that a stored block does not move the window.
func (c *Ctx) DecodeRecordsEx(rec, dst [][]byte, bsz int, blockChecksum, linked bool, d *Dict, window []byte, windowLen *int) (n []int, status []int, err error) {
	r, o := newBatch(rec, false), newBatch(dst, true)
	defer r.free()
	defer o.free()
	res := make([]C.int32_t, len(rec))
	st := make([]C.int32_t, len(rec))
	var wp unsafe.Pointer
	wl := C.int(0)
	if window != nil {
		wp = unsafe.Pointer(&window[0])
		wl = C.int(*windowLen)
	}
	rc := C.plz4hip_decode_records_ex(c.p, C.int(len(rec)), (*unsafe.Pointer)(r.ptrs), i32p(r.lens), C.int(bsz), b2i(blockChecksum), b2i(linked),
		d.ptr(), wp, &wl, (*unsafe.Pointer)(o.ptrs), i32p(res), i32p(st))
	if rc != C.PLZ4HIP_OK {
		return nil, nil, c.engineErr(rc)
	}
	if windowLen != nil {
		*windowLen = int(wl)
	}
	n, status = make([]int, len(res)), make([]int, len(res))
	for i := range res {
		n[i], status[i] = int(res[i]), int(st[i])
	}
	return
}

// DecodeRecordsChains decodes several linked frames in one call: chain k = rec[chainFirst[k]:chainFirst[k+1]], one wavefront
// each; windows holds the 64 KiB compress.DictT state of every chain back to back (len(chainFirst)-1 of them), windowLen
// their lengths, both updated.  (No counterpart in the reference, which decodes a linked frame on one goroutine, rdr.go:339-341;
// a server with many such frames hands them over together instead of one after the other.)
func (c *Ctx) DecodeRecordsChains(chainFirst []int32, rec, dst [][]byte, bsz int, blockChecksum bool, windows []byte, windowLen []int32) (n []int, status []int, err error) {
	r, o := newBatch(rec, false), newBatch(dst, true)
	defer r.free()
	defer o.free()
	res := make([]C.int32_t, len(rec)+1)
	st := make([]C.int32_t, len(rec)+1)
	nChains := len(chainFirst) - 1
	rc := C.plz4hip_decode_records_chains(c.p, C.int(nChains), (*C.int32_t)(unsafe.Pointer(&chainFirst[0])), (*unsafe.Pointer)(r.ptrs), i32p(r.lens),
		C.int(bsz), b2i(blockChecksum), unsafe.Pointer(&windows[0]), (*C.int32_t)(unsafe.Pointer(&windowLen[0])),
		(*unsafe.Pointer)(o.ptrs), i32p(res), i32p(st))
	if rc != C.PLZ4HIP_OK {
		return nil, nil, c.engineErr(rc)
	}
	n, status = make([]int, len(rec)), make([]int, len(rec))
	for i := range rec {
		n[i], status[i] = int(res[i]), int(st[i])
	}
	return
}

func b2i(b bool) C.int {
	if b {
		return 1
	}
	return 0
}


// ---- several GPUs behind one handle (plz4hip_mgpu, section D of plz4hip.h): block i of a batch runs on device i mod G.
// Multi offers the batch calls of Ctx; what needs a single device's state (dictionaries, linked decode chains, the content
// checksum stream) goes through Ctx(k).
type Multi struct {
	p    *C.plz4hip_mgpu
	ctxs []*Ctx
}

// DevicesFromEnv: PLZ4_HIP_DEVICES="0,2,5", or every device of the node.
func DevicesFromEnv() []int {
	var out []int
	if v := os.Getenv("PLZ4_HIP_DEVICES"); v != "" {
		for _, f := range strings.Split(v, ",") {
			if d, err := strconv.Atoi(strings.TrimSpace(f)); err == nil {
				out = append(out, d)
			}
		}
	}
	if len(out) == 0 {
		n := int(C.plz4hip_device_count())
		for d := 0; d < n; d++ {
			out = append(out, d)
		}
	}
	return out
}

func NewMulti(devices []int) (*Multi, error) {
	if len(devices) == 0 {
		return nil, fmt.Errorf("%w: no device", ErrEngine)
	}
	dv := make([]C.int, len(devices))
	for i, d := range devices {
		dv[i] = C.int(d)
	}
	m := &Multi{}
	if rc := C.plz4hip_mgpu_create(&dv[0], C.int(len(dv)), &m.p); rc != C.PLZ4HIP_OK {
		return nil, fmt.Errorf("%w: plz4hip_mgpu_create = %d", ErrEngine, int(rc))
	}
	for k := range devices {
		m.ctxs = append(m.ctxs, &Ctx{p: C.plz4hip_mgpu_ctx(m.p, C.int(k)), borrowed: true})
	}
	return m, nil
}

func (m *Multi) Close()         { C.plz4hip_mgpu_destroy(m.p); m.p = nil }

// Trim: Ctx.Trim on every device.
func (m *Multi) Trim() error {
	for _, c := range m.ctxs {
		if err := c.Trim(); err != nil {
			return err
		}
	}
	return nil
}
func (m *Multi) Ctx(k int) *Ctx { return m.ctxs[k%len(m.ctxs)] }
func (m *Multi) err(rc C.int) error {
	return fmt.Errorf("%w: %d: %s", ErrEngine, int(rc), C.GoString(C.plz4hip_mgpu_last_error(m.p)))
}

func (m *Multi) CompressBatch(src, dst [][]byte, level int) ([]int, error) {
	s, d := newBatch(src, false), newBatch(dst, false)
	defer s.free()
	defer d.free()
	res := make([]C.int32_t, len(src))
	if rc := C.plz4hip_mgpu_compress_batch(m.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), (*unsafe.Pointer)(d.ptrs), i32p(d.lens), C.int(level), i32p(res)); rc != C.PLZ4HIP_OK {
		return nil, m.err(rc)
	}
	return toInts(res), nil
}

func (m *Multi) DecompressBatch(src, dst [][]byte) ([]int, error) {
	s, d := newBatch(src, false), newBatch(dst, false)
	defer s.free()
	defer d.free()
	res := make([]C.int32_t, len(src))
	if rc := C.plz4hip_mgpu_decompress_batch(m.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), (*unsafe.Pointer)(d.ptrs), i32p(d.lens), i32p(res)); rc != C.PLZ4HIP_OK {
		return nil, m.err(rc)
	}
	return toInts(res), nil
}

func (m *Multi) EncodeRecords(src, rec [][]byte, bsz, level int, blockChecksum bool) ([]int, error) {
	s, r := newBatch(src, false), newBatch(rec, true)
	defer s.free()
	defer r.free()
	res := make([]C.int32_t, len(src))
	if rc := C.plz4hip_mgpu_encode_records(m.p, C.int(len(src)), (*unsafe.Pointer)(s.ptrs), i32p(s.lens), C.int(bsz), C.int(level), b2i(blockChecksum), (*unsafe.Pointer)(r.ptrs), i32p(res)); rc != C.PLZ4HIP_OK {
		return nil, m.err(rc)
	}
	return toInts(res), nil
}

func (m *Multi) DecodeRecords(rec, dst [][]byte, bsz int, blockChecksum bool) ([]int, []int, error) {
	r, d := newBatch(rec, false), newBatch(dst, true)
	defer r.free()
	defer d.free()
	res := make([]C.int32_t, len(rec))
	st := make([]C.int32_t, len(rec))
	if rc := C.plz4hip_mgpu_decode_records(m.p, C.int(len(rec)), (*unsafe.Pointer)(r.ptrs), i32p(r.lens), C.int(bsz), b2i(blockChecksum), (*unsafe.Pointer)(d.ptrs), i32p(res), i32p(st)); rc != C.PLZ4HIP_OK {
		return nil, nil, m.err(rc)
	}
	return toInts(res), toInts(st), nil
}

func toInts(v []C.int32_t) []int {
	out := make([]int, len(v))
	for i, x := range v {
		out[i] = int(x)
	}
	return out
}


// ---- streaming content checksum on the device (section B'' of plz4hip.h): xxh32.XXHZero kept by the engine.  While a stream is
// attached to a ctx, EncodeRecords[Ex] / DecodeRecords[Ex] on that ctx also write the plaintext of their blocks into it, in order
// (async/hash.go:99-111 without the plaintext passing through a Go hasher).
type HashStream struct {
	c *Ctx
	p *C.plz4hip_xxh32_stream
}

func (c *Ctx) NewHashStream() (*HashStream, error) {
	h := &HashStream{c: c}
	if rc := C.plz4hip_xxh32_stream_create(c.p, &h.p); rc != C.PLZ4HIP_OK {
		return nil, c.engineErr(rc)
	}
	return h, nil
}
func (h *HashStream) Close()        { C.plz4hip_xxh32_stream_destroy(h.c.p, h.p); h.p = nil }
func (h *HashStream) Reset() error  { return h.c.chk(C.plz4hip_xxh32_stream_reset(h.c.p, h.p)) }
func (h *HashStream) Attach() error { return h.c.chk(C.plz4hip_ctx_set_content_hash(h.c.p, h.p)) }
func (h *HashStream) Detach() error { return h.c.chk(C.plz4hip_ctx_set_content_hash(h.c.p, nil)) }
func (h *HashStream) Write(b []byte) (int, error) { // io.Writer over host bytes (what is not an encode / decode call's plaintext)
	if len(b) == 0 {
		return 0, nil
	}
	var pin runtime.Pinner
	pin.Pin(&b[0])
	defer pin.Unpin()
	if rc := C.plz4hip_xxh32_stream_update(h.c.p, h.p, unsafe.Pointer(&b[0]), C.int64_t(len(b))); rc != C.PLZ4HIP_OK {
		return 0, h.c.engineErr(rc)
	}
	return len(b), nil
}
func (h *HashStream) Sum32() (uint32, error) {
	var out C.uint32_t
	if rc := C.plz4hip_xxh32_stream_sum(h.c.p, h.p, &out); rc != C.PLZ4HIP_OK {
		return 0, h.c.engineErr(rc)
	}
	return uint32(out), nil
}
func (c *Ctx) chk(rc C.int) error {
	if rc != C.PLZ4HIP_OK {
		return c.engineErr(rc)
	}
	return nil
}
