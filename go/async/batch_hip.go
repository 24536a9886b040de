//go:build cgo && plz4_hip

// Batching stages for the HIP engine: what replaces asyncWriterT.compressLoop (internal/pkg/async/writer.go:232-282) and
// asyncRdrT._decompressLoop (internal/pkg/async/reader.go:192-221) in a plz4 built with -tags plz4_hip.  The originals keep their
// build tag `!plz4_hip`; nothing else in the package changes: kickoffAsync (writer.go:439-467) starts ONE of these goroutines instead
// of NParallel workers, the semaphore (semChan, capacity opts.CalcPending) still bounds the blocks in flight, writeLoop / the
// reader's _nextBlock still put results back in block order by idx.
//
// Why a batch: one cgo call per 4 MiB block cannot feed a GPU (a block is one wavefront: ~100 ms alone, but thousands run side by
// side).  So the stage drains inChan into a slice until (a) the writer admits no more blocks (the semaphore is full), (b) the
// input is closed, or (c) nothing arrived for `gatherWait` -- without (c) a Flush() with a partial batch in hand would wait for
// ever -- and hands the whole slice to the engine in one call.  Below `minBatch` blocks the batch goes block by block through the
// CPU compressor instead (liblz4 through cgo, as today): through host memory the engine only draws level with the host at
// several hundred blocks per call (bench.py --crossover, profiles/r03_crossover.json), so a writer that only ever has a few
// blocks in flight must not pay for it.  PLZ4_HIP_MIN_BATCH overrides.
//
// NOT COMPILED IN THE BUILD IMAGE (no Go toolchain): the C++ mirror (plz4_amd/csrc/host/plz4_host.cpp, WriterImpl::submit /
// ReaderImpl::fill) is the executable form of the same logic and is what the tests drive.  See INTEGRATION.md §2.
package async

import (
	"os"
	"strconv"
	"time"

	"github.com/prequel-dev/plz4/internal/pkg/blk"
	"github.com/prequel-dev/plz4/internal/pkg/compress"
	"github.com/prequel-dev/plz4/internal/pkg/descriptor"
	"github.com/prequel-dev/plz4/internal/pkg/zerr"

	"encoding/binary"
)

const gatherWait = 200 * time.Microsecond

// A batch below this many blocks goes block by block through the CPU compressor: the level-1 ENCODER is one wavefront per block
// (89 ms per 4 MiB block whatever the batch), so through host memory the engine only draws level with the host's cores at about a
// thousand blocks per call (DESIGN.md 6.0).  The batch can never exceed the writer's window -- cap(semChan) = opts.CalcPending(),
// NParallel by default (opts/opts.go:62-95) -- so a deployment that wants the GPU on the write side sets WithPendingSize (or
// NParallel) to a few thousand blocks; with default options this build compresses on the host, by design.  The READ side has a
// few-block path since round 4 (one 4 MiB block in 1.4 ms, 16 in 9.8 ms through host memory; DESIGN.md 3.9) and its own, lower
// threshold.
func hipMinBatch() int {
	if v, err := strconv.Atoi(os.Getenv("PLZ4_HIP_MIN_BATCH")); err == nil && v >= 1 {
		return v
	}
	return 256
}
func hipMinReadBatch() int {
	if v, err := strconv.Atoi(os.Getenv("PLZ4_HIP_MIN_READ_BATCH")); err == nil && v >= 1 {
		return v
	}
	return 32 // (below that the host's cores decode a handful of blocks faster than the copies to and from the device take)
}

// gather: the first block is waited for; more are taken while they keep coming, up to max.  closed: inChan was closed.
// full() says that the producer cannot admit another block (its semaphore is at capacity): then nothing more will arrive once
// inChan is drained, so the blocks already queued are taken first and the wait is skipped -- checking full() before looking
// at the channel would strand every admitted block but the first in batches of one.
func gatherIn(in <-chan inBlkT, max int, full func() bool) (batch []inBlkT, closed bool) {
	first, ok := <-in
	if !ok {
		return nil, true
	}
	batch = append(batch, first)
	t := time.NewTimer(gatherWait)
	defer t.Stop()
	for len(batch) < max {
		select { // what is queued already
		case b, ok := <-in:
			if !ok {
				return batch, true
			}
			batch = append(batch, b)
			continue
		default:
		}
		if full() {
			return batch, false
		}
		if !t.Stop() {
			select {
			case <-t.C:
			default:
			}
		}
		t.Reset(gatherWait)
		select {
		case b, ok := <-in:
			if !ok {
				return batch, true
			}
			batch = append(batch, b)
		case <-t.C:
			return batch, false
		}
	}
	return batch, false
}

// compressLoop, HIP build.  Same contract as the reference's: every block taken from inChan produces exactly one outBlkT on
// outChan (or, in the error state, is dropped with its semaphore slot released), source blocks go back to the pool / the hasher.
func (w *asyncWriterT) compressLoop() {
	defer w.wg.Done()

	var (
		bsz      = w.bsz
		cmp      = w.cmpF.NewCompressor() // block at a time: the CPU path below minBatch, and every stateful (linked) compressor
		bc, _    = cmp.(compress.BatchCompressor)
		blkCheck = w.opts.BlockChecksum
		minBatch = hipMinBatch()
		maxBatch = cap(w.semChan)
	)

	freeSrcBlk := func(srcBlk inBlkT) {
		blk.ReturnBlk(srcBlk.dict)
		if w.hasher != nil {
			w.hasher.Free(srcBlk.blk, srcBlk.idx) // coordinate block free with hasher (writer.go:244-249)
		} else {
			blk.ReturnBlk(srcBlk.blk)
		}
	}
	one := func(srcBlk inBlkT) {
		srcSz := srcBlk.blk.Len()
		dstBlk, err := srcBlk.blk.Compress(cmp, bsz, blkCheck, srcBlk.Dict())
		freeSrcBlk(srcBlk)
		w.outChan <- outBlkT{err: err, idx: srcBlk.idx, blk: dstBlk, srcSz: srcSz}
	}

	for {
		batch, closed := gatherIn(w.inChan, maxBatch, func() bool { return len(w.semChan) == cap(w.semChan) })

		// error state: don't bother compressing, drop and release (writer.go:260-264)
		if w.errState() {
			for _, b := range batch {
				freeSrcBlk(b)
				<-w.semChan
			}
			batch = nil
		}

		switch {
		case len(batch) == 0:
		case bc == nil || len(batch) < minBatch:
			for _, b := range batch {
				one(b)
			}
		default:
			// blk.CompressToBlk (blk/blk.go:69-109) for the whole batch: the engine writes [size word][payload][xxh32] into
			// the pooled destination blocks; a block liblz4 would not fit comes back stored (the engine applies "stored iff
			// the encoder returned 0" itself); an engine failure is NOT ErrCompress and aborts the stream like any other error.
			src := make([][]byte, len(batch))
			dst := make([]*blk.BlkT, len(batch))
			rec := make([][]byte, len(batch))
			for i, b := range batch {
				src[i] = b.blk.Data()
				dst[i] = blk.BorrowBlk(bsz)
				rec[i] = dst[i].Data()[:bsz+8]
			}
			n, err := bc.EncodeRecords(src, rec, bsz, blkCheck)
			for i, b := range batch {
				out := outBlkT{idx: b.idx, srcSz: b.blk.Len()}
				if err != nil {
					blk.ReturnBlk(dst[i])
					out.err = err
				} else {
					dst[i].Trim(n[i])
					out.blk = dst[i]
				}
				freeSrcBlk(b)
				w.outChan <- out
			}
		}

		if closed {
			return // compressLoop only exits on close of w.inChan
		}
	}
}

// _decompressLoop, HIP build: the frame reader has parsed the size word and verified the block checksum on the host
// (blk/frame.go:54-127), so what arrives is a raw LZ4 block or a stored block; a batch goes through DecompressBatch.
func (r *asyncRdrT) _decompressLoop() {
	var (
		bd, _    = r.dc.(compress.BatchDecompressorRaw)
		minBatch = hipMinReadBatch()
	)
	one := func(srcBlk inBlkT) bool {
		dstBlk, err := srcBlk.blk.Decompress(r.dc)
		blk.ReturnBlk(srcBlk.blk)
		select {
		case r.outChan <- outBlkT{err: err, idx: srcBlk.idx, blk: dstBlk, srcSz: srcBlk.srcSz}:
			return true
		case <-r.finChan:
			if dstBlk != nil {
				blk.ReturnBlk(dstBlk)
			}
			return false
		}
	}
	for {
		batch, closed := gatherIn(r.inChan, cap(r.outChan)+r.opts.NParallel, func() bool { return false })
		if bd == nil || len(batch) < minBatch {
			for _, b := range batch {
				if !one(b) {
					return
				}
			}
		} else if len(batch) > 0 {
			src := make([][]byte, len(batch))
			dst := make([]*blk.BlkT, len(batch))
			out := make([][]byte, len(batch))
			for i, b := range batch {
				src[i] = b.blk.Data()
				dst[i] = blk.BorrowBlk(b.blk.Cap() - 8) // == BlkT.Decompress (blk.go:51-53; szOverhead = 8, pool.go:15): decoded into the whole pooled buffer, bsz + 8
				out[i] = dst[i].Data()
			}
			n, err := bd.DecompressBatch(src, out)
			for i, b := range batch {
				o := outBlkT{idx: b.idx, srcSz: b.srcSz}
				switch {
				case err != nil:
					o.err = err
				case n[i] < 0:
					o.err = zerr.WrapCorrupted(zerr.ErrDecompress) // decompress.go:32-38
				default:
					dst[i].Trim(n[i])
					o.blk = dst[i]
				}
				if o.blk == nil {
					blk.ReturnBlk(dst[i])
				}
				blk.ReturnBlk(b.blk)
				select {
				case r.outChan <- o:
				case <-r.finChan:
					if o.blk != nil {
						blk.ReturnBlk(o.blk)
					}
					return
				}
			}
		}
		if closed {
			return
		}
	}
}

// (kept for reference: the stored-block size word the engine writes is descriptor.DataBlockSize with the high bit set,
// exactly what CompressToBlk writes -- blk/blk.go:94-96)
var _ = func() { var s descriptor.DataBlockSize; s.SetUncompressed(); _ = binary.LittleEndian }
