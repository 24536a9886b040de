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

type LevelT int

var (
	once sync.Once
	ctx  *plz4hip.Ctx
	cerr error
)

func engine() (*plz4hip.Ctx, error) {
	once.Do(func() { ctx, cerr = plz4hip.NewCtx(0) })
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

func (hipDecompressor) DecodeRecords(rec, dst [][]byte, bsz int, blockChecksum bool) ([]int, []int, error) {
	e, err := engine()
	if err != nil {
		return nil, nil, err
	}
	return e.DecodeRecords(rec, dst, bsz, blockChecksum)
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
	// Levels 1..12 are served for independent blocks.  The engine also has level-1 dictionaries and linked blocks
	// (plz4hip_dict_create, plz4hip_encode_records_ex / decode_records_ex); this thin shim does not bind them yet.
	if f.level < 1 || f.level > 12 || !f.indie || f.dict != nil {
		panic("plz4_hip: independent blocks without a dictionary only in this shim; build without the plz4_hip tag for the rest")
	}
	return hipCompressor{level: f.level}
}

func NewDecompressor(independent bool, dict *DictT) Decompressor {
	if !independent || dict != nil {
		panic("plz4_hip: linked blocks / dictionaries are not bound by this shim yet")
	}
	return hipDecompressor{}
}

func CompressBound(sz int) int { return plz4hip.CompressBound(sz) }
