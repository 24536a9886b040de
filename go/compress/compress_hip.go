//go:build cgo && plz4_hip

// Drop-in for internal/pkg/compress (third implementation next to compress.go `cgo` and nocgo_compress.go `!cgo`):
// same Compressor / Decompressor interfaces (compress.go:7-13, decompress.go:14-16), so plz4_block.go and every caller
// that works block-at-a-time keeps compiling unchanged, plus the batch interfaces the writer/reader loops use to hand the
// engine many blocks at once (where compressLoop / _decompressLoop sit today: async/writer.go:232-282, async/reader.go:192-221).
//
// NOT COMPILED IN THE BUILD IMAGE (no Go toolchain): see INTEGRATION.md.
package compress

import (
	"errors"
	"fmt"
	"sync"

	"github.com/prequel-dev/plz4/internal/pkg/plz4hip"
	"github.com/prequel-dev/plz4/internal/pkg/zerr"
)

type Compressor interface {
	Compress(src, dst, dict []byte) (int, error)
}
type Decompressor interface {
	Decompress(src, dst []byte) (int, error)
}

// BatchCompressor / BatchDecompressor: what blk.CompressToBlk / BlkT.Decompress become when N blocks are in flight.
type BatchCompressor interface {
	EncodeRecords(src, rec [][]byte, bsz int, blockChecksum bool) ([]int, error)
}
type BatchDecompressor interface {
	DecodeRecords(rec, dst [][]byte, bsz int, blockChecksum bool) (n []int, status []int, err error)
}

// BatchDecompressorRaw: raw LZ4 blocks, n[i] < 0 is liblz4's error code (what async/batch_hip.go hands the reader's blocks to:
// the frame reader has taken the size word and the checksum off already).
type BatchDecompressorRaw interface {
	DecompressBatch(src, dst [][]byte) (n []int, err error)
}

type LevelT int

// One engine per process over EVERY GPU of the node (plz4hip_mgpu, section D of plz4hip.h): the blocks of a batch are dealt round
// robin over the devices (the device that takes a call's block 0 rotates from call to call, so one-block calls from many
// goroutines spread over the GPUs), each device behind its own ctx and its own lock -- the handle itself takes none for these calls.
// PLZ4_HIP_DEVICES=0,2,5 restricts the set (default: all of them).
var (
	once sync.Once
	ctx  *plz4hip.Multi
	cerr error
)

func engine() (*plz4hip.Multi, error) {
	once.Do(func() { ctx, cerr = plz4hip.NewMulti(plz4hip.DevicesFromEnv()) })
	return ctx, cerr
}

type hipCompressor struct{ level LevelT }

func (c hipCompressor) Compress(src, dst, _ []byte) (int, error) {
	e, err := engine()
	if err != nil {
		return 0, err // NOT ErrCompress: an engine failure must not become a stored block
	}
	n, err := e.CompressBatch([][]byte{src}, [][]byte{dst}, int(c.level))
	if err != nil {
		return 0, err
	}
	if n[0] == 0 {
		return 0, errors.Join(zerr.ErrCompress, plz4hip.ErrLz4Compress) // indie.go:66-74
	}
	return n[0], nil
}

func (c hipCompressor) EncodeRecords(src, rec [][]byte, bsz int, blockChecksum bool) ([]int, error) {
	e, err := engine()
	if err != nil {
		return nil, err
	}
	return e.EncodeRecords(src, rec, bsz, int(c.level), blockChecksum)
}

type hipDecompressor struct{}

func (hipDecompressor) Decompress(src, dst []byte) (int, error) {
	e, err := engine()
	if err != nil {
		return 0, err
	}
	n, err := e.DecompressBatch([][]byte{src}, [][]byte{dst})
	if err != nil {
		return 0, err
	}
	if n[0] < 0 {
		return 0, errors.Join(zerr.ErrCorrupted, zerr.ErrDecompress, fmt.Errorf("%w: code %d", plz4hip.ErrLz4Decompress, n[0])) // decompress.go:32-38
	}
	return n[0], nil
}

func (hipDecompressor) DecompressBatch(src, dst [][]byte) ([]int, error) {
	e, err := engine()
	if err != nil {
		return nil, err
	}
	return e.DecompressBatch(src, dst)
}

func (hipDecompressor) DecodeRecords(rec, dst [][]byte, bsz int, blockChecksum bool) ([]int, []int, error) {
	e, err := engine()
	if err != nil {
		return nil, nil, err
	}
	return e.DecodeRecords(rec, dst, bsz, blockChecksum)
}

// DictT is the reference's dictionary holder (compress/dict.go:5-24); here it also owns the device-side context.
type DictT struct {
	Data []byte        // the last 64 KiB of the user dictionary
	dev  *plz4hip.Dict // plz4hip_dict_create(Data), built on first use
}

// device: the dictionary's device-side context.  A plz4hip_dict belongs to the ctx that created it, so dictionary / linked calls
// all go through device 0's ctx (e.Ctx(0) below and at every call site): they are stateful per frame anyway and do not deal.
func (d *DictT) device(e *plz4hip.Multi) (*plz4hip.Dict, error) {
	if d == nil {
		return nil, nil
	}
	if d.dev == nil {
		dd, err := e.Ctx(0).NewDict(d.Data)
		if err != nil {
			return nil, err
		}
		d.dev = dd
	}
	return d.dev, nil
}

// hipDictCompressor: a dictionary and/or linked blocks at any level (indieCompressorDict / indieCompressorDictHC /
// linkedCompressor / linkedCompressorHC of the reference, compress/indie.go:12-60, compress/linked.go).  Block-at-a-time Compress keeps the previous block's tail itself, the batch
// form hands the whole window of blocks to the engine (block i is primed with block i-1 on the device).
type hipDictCompressor struct {
	level  LevelT
	linked bool
	dict   *DictT
	tail   []byte // linked: last <= 64 KiB of the previous block of this frame; nil before the first block
}

func (c *hipDictCompressor) EncodeRecords(src, rec [][]byte, bsz int, blockChecksum bool) ([]int, error) {
	e, err := engine()
	if err != nil {
		return nil, err
	}
	dd, err := c.dict.device(e)
	if err != nil {
		return nil, err
	}
	n, err := e.Ctx(0).EncodeRecordsEx(src, rec, bsz, int(c.level), blockChecksum, c.linked, dd, c.tail)
	if err == nil && c.linked && c.level > 1 {
		// linkedCompressorHC.Compress returns liblz4's "does not fit" without joining zerr.ErrCompress (compress/linked.go:47-49),
		// so blk.CompressToBlk does not store the block raw there (blk/blk.go:75-86): the write fails.  Kept.
		for i := range rec {
			if n[i] >= 4 && rec[i][3]&0x80 != 0 {
				return nil, plz4hip.ErrLz4Compress
			}
		}
	}
	if err == nil && c.linked && len(src) > 0 {
		last := src[len(src)-1]
		if len(last) > 65536 {
			last = last[len(last)-65536:]
		}
		c.tail = append(c.tail[:0], last...)
	}
	return n, err
}

func (c *hipDictCompressor) Compress(src, dst, _ []byte) (int, error) {
	e, err := engine()
	if err != nil {
		return 0, err
	}
	dd, err := c.dict.device(e)
	if err != nil {
		return 0, err
	}
	if c.linked {
		return 0, errors.New("plz4_hip: linked blocks go through EncodeRecords (the frame writer), not through Compress")
	}
	n, err := e.Ctx(0).CompressBatchDict([][]byte{src}, [][]byte{dst}, int(c.level), dd)
	if err != nil {
		return 0, err
	}
	if n[0] == 0 {
		return 0, errors.Join(zerr.ErrCompress, plz4hip.ErrLz4Compress)
	}
	return n[0], nil
}

type hipDictDecompressor struct {
	linked bool
	dict   *DictT
	window []byte // linked: compress.DictT's sliding 64 KiB
	wlen   int
}

func (c *hipDictDecompressor) Decompress(src, dst []byte) (int, error) {
	e, err := engine()
	if err != nil {
		return 0, err
	}
	dd, err := c.dict.device(e)
	if err != nil {
		return 0, err
	}
	if c.linked {
		return 0, errors.New("plz4_hip: linked blocks go through DecodeRecords (the frame reader), not through Decompress")
	}
	n, err := e.Ctx(0).DecompressBatchDict([][]byte{src}, [][]byte{dst}, dd)
	if err != nil {
		return 0, err
	}
	if n[0] < 0 {
		return 0, errors.Join(zerr.ErrCorrupted, zerr.ErrDecompress, fmt.Errorf("%w: code %d", plz4hip.ErrLz4Decompress, n[0]))
	}
	return n[0], nil
}

func (c *hipDictDecompressor) DecodeRecords(rec, dst [][]byte, bsz int, blockChecksum bool) ([]int, []int, error) {
	e, err := engine()
	if err != nil {
		return nil, nil, err
	}
	dd, err := c.dict.device(e)
	if err != nil {
		return nil, nil, err
	}
	if !c.linked {
		return e.Ctx(0).DecodeRecordsEx(rec, dst, bsz, blockChecksum, false, dd, nil, nil)
	}
	if c.window == nil { // the window starts as the dictionary's last 64 KiB (compress/dict.go:43-56)
		c.window = make([]byte, 65536)
		if c.dict != nil {
			c.wlen = copy(c.window, c.dict.Data)
		}
	}
	return e.Ctx(0).DecodeRecordsEx(rec, dst, bsz, blockChecksum, true, nil, c.window, &c.wlen)
}

type CompressorFactory struct {
	indie bool
	level LevelT
	dict  *DictT
}

func NewCompressorFactory(level LevelT, independent bool, dict *DictT) CompressorFactory {
	return CompressorFactory{level: level, indie: independent, dict: dict}
}

func (f CompressorFactory) NewCompressor() Compressor {
	// Levels 1..12, with or without a dictionary, independent or linked blocks.
	if f.level < 1 || f.level > 12 {
		panic("plz4_hip: level out of range")
	}
	if f.indie && f.dict == nil {
		return hipCompressor{level: f.level}
	}
	return &hipDictCompressor{level: f.level, linked: !f.indie, dict: f.dict}
}

func NewDecompressor(independent bool, dict *DictT) Decompressor {
	if independent && dict == nil {
		return hipDecompressor{}
	}
	return &hipDictDecompressor{linked: !independent, dict: dict}
}

func CompressBound(sz int) int { return plz4hip.CompressBound(sz) }
