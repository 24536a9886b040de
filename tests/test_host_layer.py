"""The C++ host layer (NewWriter / NewReader / CompressBlock / DecompressBlock mirror, plz4_amd/csrc/host) driven the way
the reference's own tests drive the Go API (internal/test/wr_test.go, rd_test.go, block_test.go, plz4_test.go), with an
oracle-backed engine so it runs without a GPU.  The same cases run against the HIP engine in test_gpu_host.py."""
import hashlib

import numpy as np
import pytest

import corpus
from plz4_amd import host, synth
from test_oracle_kats import HELLO_FRAME, THE_WORKS, ONE_FRAME, ONE_FRAME_NOHASH, ONE_FRAME_SHA, HEADER_KATS, MAGIC


@pytest.fixture(scope="module")
def eng():
    from hostengine import oracle_engine
    e = oracle_engine()
    yield e
    e.close()


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ------------------------------------------------------------------------------------------------ KATs
def test_example_new_writer(eng):
    """plz4_test.go:41-75."""
    w = host.Writer(eng, parallel=1, content_checksum=False)
    assert w.write(b"hello") == (5, 0)
    assert not w.flush()
    assert not w.close()
    assert w.output() == HELLO_FRAME
    assert int(w.close()) == host.ErrClosed                # double close is harmless (async/writer.go:136-138)


def test_example_new_reader(eng):
    """plz4_test.go:9-39."""
    r = host.Reader(eng, HELLO_FRAME, parallel=0)
    n, out, err = r.write_to()
    assert (n, out, int(err)) == (5, b"hello", 0)
    assert not r.close() and int(r.close()) == host.ErrClosed


def test_the_works_written_byte_exact(eng):
    """rd_test.go:527-538: content size 9, dict id 11, block + content checksums; two flushed blocks."""
    w = host.Writer(eng, parallel=0, content_size=9, dictionary_id=11, block_checksum=True, content_checksum=True)
    assert w.write(b"testy")[1] == 0 and not w.flush()
    assert w.write(b"code")[1] == 0
    assert not w.close()
    assert w.output() == THE_WORKS


def test_header_kats(eng):
    for name, (kw, body) in HEADER_KATS.items():
        k = dict(block_size=kw["bs_idx"], block_linked=kw.get("linked", False), block_checksum=kw.get("block_checksum", False),
                 content_checksum=kw.get("content_checksum", False), content_size=kw.get("content_size"),
                 dictionary_id=kw.get("dict_id"))
        assert host.write_header(**k) == MAGIC + bytes(body), name


# ------------------------------------------------------------------------------------------------ writer matrix (wr_test.go:50-195)
WRITE_CASES = {
    "defaults": {},
    "no_content_checksum": dict(content_checksum=False),
    "block_checksum": dict(block_checksum=True),
    "both_checksums_64k": dict(block_checksum=True, block_size=host.BlockIdx64KB),
    "256k": dict(block_size=host.BlockIdx256KB),
    "1m": dict(block_size=host.BlockIdx1MB),
    "content_size": dict(content_size=None),          # filled in per payload
}


@pytest.mark.parametrize("parallel", [0, 1, 4])
@pytest.mark.parametrize("case", sorted(WRITE_CASES))
def test_writer_matrix(eng, orc, parallel, case):
    kw = dict(WRITE_CASES[case])
    payload = synth.make("M", (1 << 20) * 3 + 12345, 256 << 10).tobytes()
    if case == "content_size":
        kw["content_size"] = len(payload)
    bs_idx = kw.get("block_size", host.BlockIdx4MB)
    # a) ReadFrom (wr_test.go:107)  b) Write in 24 KiB pieces (:138)  c) random 1-3 MiB pieces (:172)
    frames = []
    w = host.Writer(eng, parallel=parallel, **kw); n, e = w.read_from(payload, chunk=70001); assert (n, int(e)) == (len(payload), 0)
    assert not w.close(); frames.append(w.output())
    w = host.Writer(eng, parallel=parallel, **kw)
    for o in range(0, len(payload), 24 << 10):
        assert w.write(payload[o:o + (24 << 10)])[1] == 0
    assert not w.close(); frames.append(w.output())
    rng = np.random.default_rng(1); o = 0
    w = host.Writer(eng, parallel=parallel, **kw)
    while o < len(payload):
        k = int(rng.integers(1 << 20, 3 << 20)); assert w.write(payload[o:o + k])[1] == 0; o += k
    assert not w.close(); frames.append(w.output())
    assert frames[0] == frames[1] == frames[2]                       # block boundaries depend only on the byte stream
    # oracle: same frame bytes as the reference's sync writer produces, and SHA-256 of the round trip (wr_test.go:128)
    if "content_size" not in kw:
        want = orc.frame_encode(np.frombuffer(payload, dtype=np.uint8), bs_idx, kw.get("block_checksum", False),
                                kw.get("content_checksum", True))
        assert frames[0] == want.tobytes()
    for par in (0, 2):
        r = host.Reader(eng, frames[0], parallel=par)
        n, out, err = r.write_to()
        assert int(err) == 0 and n == len(payload) and sha(out) == sha(payload)


def test_uncompressable_blocks_are_stored(eng, orc):
    payload = synth.random_bytes(3 * (64 << 10) + 100).tobytes()
    w = host.Writer(eng, parallel=1, block_size=host.BlockIdx64KB, block_checksum=True)
    assert w.write(payload)[1] == 0 and not w.close()
    f = w.output()
    assert f[7 + 3] & 0x80                                   # first block: stored flag in the size word
    n, out, err = host.Reader(eng, f).write_to()
    assert int(err) == 0 and out == payload


def test_empty_input_sync_vs_async(eng):
    """SURVEY §8a trap 9: the async writer emits nothing for empty input, the sync writer header + end mark (+ hash)."""
    w = host.Writer(eng, parallel=1); assert not w.close(); assert w.output() == b""
    w = host.Writer(eng, parallel=0); assert not w.close()
    assert w.output() == host.write_header(content_checksum=True) + bytes(4) + (0x02cc5d05).to_bytes(4, "little")
    w = host.Writer(eng, parallel=0, content_checksum=False); assert not w.close()
    assert w.output() == host.write_header(content_checksum=False) + bytes(4)


def test_flush_makes_short_blocks(eng, orc):
    """wr_test.go:238-346: 1-byte writes with Flush in between produce one block per flush; round trip intact."""
    w = host.Writer(eng, parallel=1, block_size=host.BlockIdx64KB)
    data = bytes(range(40))
    for b in data:
        assert w.write(bytes([b]))[1] == 0 and not w.flush()
    assert not w.close()
    f = w.output()
    n, out, err = host.Reader(eng, f).write_to()
    assert int(err) == 0 and out == data
    assert f.count(b"\x02\x00\x00\x00") >= 40                # forty 2-byte blocks (token + literal)


def test_progress_and_read_offset(eng):
    """wr_test.go:202-232: progress pairs are block boundaries usable with WithReadOffset."""
    payload = synth.text(5 * (64 << 10) + 777).tobytes()
    w = host.Writer(eng, parallel=2, block_size=host.BlockIdx64KB, content_checksum=False)
    assert w.write(payload)[1] == 0 and not w.close()
    f = w.output(); pairs = w.progress()
    assert [p[0] for p in pairs] == [0, 65536, 131072, 196608, 262144, 327680, len(payload)]
    assert pairs[0][1] == 7 and pairs[-1][1] == len(f) - 4
    for src_off, dst_off in pairs[:-1]:
        n, out, err = host.Reader(eng, f, read_offset=dst_off).write_to()
        assert int(err) == 0 and out == payload[src_off:]
    n, out, err = host.Reader(eng, f, read_offset=3).write_to()
    assert int(err) == host.ErrReadOffset


def test_writer_sink_failures(eng):
    """wr_test.go:853-960: fail the N-th Write of the io.Writer; the error is reported once, Close() then returns nil."""
    payload = synth.text(4 * (64 << 10)).tobytes()
    for par in (0, 1):
        for nfail in range(0, 6):
            w = host.Writer(eng, parallel=par, block_size=host.BlockIdx64KB, fail_after_writes=nfail)
            n, e1 = w.write(payload)
            e2 = w.close()
            if e1:                                                   # reported by Write: Close() then returns nil
                assert int(e1) in (host.ErrIO, host.ErrHeaderWrite) and not e2, (par, nfail)
            else:
                assert int(e2) in (host.ErrIO, host.ErrHeaderWrite), (par, nfail)
        w = host.Writer(eng, parallel=par, block_size=host.BlockIdx64KB, fail_after_writes=6)
        assert w.write(payload)[1] == 0 and not w.close()            # header + 4 blocks + trailer = 6 writes


def test_write_after_close(eng):
    w = host.Writer(eng, parallel=1)
    assert w.write(b"abc")[1] == 0 and not w.close()
    assert int(w.write(b"x")[1]) == host.ErrClosed


def test_unsupported_modes_fail_loudly(eng):
    """An engine that answers PLZ4HIP_E_UNSUPPORTED (the oracle-backed test engine has no HC levels; the HIP engine builds every
    level, tests/test_hc_dict.py): the writer reports ErrUnsupported, it never falls back to other bytes."""
    for kw in (dict(level=9, block_linked=True), dict(level=12, block_linked=True), dict(level=2, dictionary=b"abcd" * 64)):
        w = host.Writer(eng, parallel=1, **kw)
        w.write(b"some payload that needs compressing")
        assert int(w.close()) == host.ErrUnsupported


def test_linked_frames_above_level_1_have_no_stored_fallback(eng, monkeypatch):
    """compress/linked.go:47-49: linkedCompressorHC.Compress returns liblz4's error without joining zerr.ErrCompress, so
    blk.CompressToBlk (blk/blk.go:75-86) does not store the block raw: the write fails.  Independent blocks at the same level,
    and linked blocks at level 1, fall back to a stored block.  (The test engine answers level 9 with its level-1 encoder:
    only the host layer's handling is under test here; the HIP engine's bytes are in tests/test_hc_dict.py.)"""
    monkeypatch.setenv("ORACLE_ENGINE_ANY_LEVEL", "1")
    noise = synth.random_bytes(3 * (64 << 10), seed=8).tobytes()
    for par in (0, 2):
        w = host.Writer(eng, parallel=par, level=9, block_size=host.BlockIdx64KB, block_linked=True)
        w.write(noise)
        assert int(w.close()) == host.ErrCompress
        for kw in (dict(level=9, block_linked=False), dict(level=1, block_linked=True)):
            w = host.Writer(eng, parallel=par, block_size=host.BlockIdx64KB, **kw)
            w.write(noise)
            assert not w.close()
            n, out, err = host.Reader(eng, w.output()).write_to()
            assert not err and out == noise
    small = host.Writer(eng, parallel=2, level=9, block_size=host.BlockIdx64KB, block_linked=True)     # the _writeSync shortcut
    small.write(noise[:65535])
    assert int(small.close()) == host.ErrCompress


# ------------------------------------------------------------------------------------------------ config 5 (wr_test.go: dict, linked, linked_with_dict)
DICT = synth.text(70000, seed=321).tobytes()


@pytest.mark.parametrize("case", ["dict", "linked", "linked_with_dict"])
@pytest.mark.parametrize("bs", [host.BlockIdx64KB, host.BlockIdx256KB])
def test_writer_dict_and_linked_roundtrip(eng, case, bs):
    kw = dict(block_size=bs, block_checksum=True)
    if "dict" in case:
        kw["dictionary"] = DICT
    if "linked" in case:
        kw["block_linked"] = True
    payload = (DICT[5000:45000] + synth.make("M", 900000, 64 << 10, seed=8).tobytes())
    frames = []
    for par, batch in ((1, 0), (4, 2), (0, 3)):              # different batch splits must not change a byte
        w = host.Writer(eng, parallel=par, gpu_batch=batch, **kw)
        for o in range(0, len(payload), 100000):
            assert w.write(payload[o:o + 100000])[1] == 0
        assert not w.close()
        frames.append(w.output())
    assert frames[0] == frames[1] == frames[2]
    flg = frames[0][4]
    assert bool(flg & 0x20) == ("linked" not in case)
    rkw = {"dictionary": DICT} if "dict" in case else {}
    for batch in (0, 1, 3):
        n, out, err = host.Reader(eng, frames[0], gpu_batch=batch, **rkw).write_to()
        assert not err and out == payload
    if "dict" in case:                                       # the dictionary is really used: decoding without it fails or differs
        n, out, err = host.Reader(eng, frames[0]).write_to()
        assert err or out != payload
    if "linked" in case:
        n, out, err = host.Reader(eng, frames[0], read_offset=7 + 4, **rkw).write_to()
        assert int(err) in (host.ErrReadOffsetLinked, host.ErrReadOffset)


def test_dictionary_makes_small_payloads_smaller(eng):
    """README.md:29 use case: many small payloads sharing a dictionary."""
    msg = DICT[10000:10400]                                   # inside the last 64 KiB, the part liblz4 keeps
    plain, e1 = host.compress_block(eng, msg)
    withd, e2 = host.compress_block(eng, msg, dictionary=DICT)
    assert not e1 and not e2 and len(withd) < len(plain) // 3
    back, e = host.decompress_block(eng, withd, dictionary=DICT)
    assert not e and back == msg
    back, e = host.decompress_block(eng, withd, dst_cap=len(msg), dictionary=DICT)
    assert not e and back == msg
    bad, e = host.decompress_block(eng, withd, dst_cap=len(msg))
    assert e or bad != msg


# ------------------------------------------------------------------------------------------------ reader (rd_test.go)
SHORT_READ = [  # name, clipOff, clipCnt, expect, err      rd_test.go:548-611
    ("clip_content_hash", 54, 4, 9, host.ErrContentHashRead),
    ("clip_read_trailer", 50, 4, 9, host.ErrBlockSizeRead),
    ("clip_read_block2_crc", 46, 4, 5, host.ErrBlockRead),
    ("clip_read_block2", 42, 5, 5, host.ErrBlockRead),
    ("clip_read_block2_size", 37, 4, 5, host.ErrBlockSizeRead),
    ("clip_read_block1_crc", 33, 4, 0, host.ErrBlockRead),
    ("clip_read_block1", 29, 6, 0, host.ErrBlockRead),
    ("clip_read_block1_size", 23, 4, 0, host.ErrBlockSizeRead),
    ("clip_read_header_crc", 19, 18, 0, host.ErrHeaderRead),
]


@pytest.mark.parametrize("name,clip_off,clip_cnt,expect,want", SHORT_READ)
def test_short_read(eng, name, clip_off, clip_cnt, expect, want):
    decoded = b"testycode"
    for i in range(1, clip_cnt + 1):
        data = THE_WORKS[:clip_off - i]
        n, out, err = host.Reader(eng, data).write_to()
        assert int(err) == want and not err.corrupted and n == expect and out == decoded[:expect], (name, i)
        r = host.Reader(eng, data)
        got, err = r.read(len(THE_WORKS) + 16)
        if not err and got:
            got2, err = r.read(len(THE_WORKS) + 4)                   # deferred error (rdr.go:71-80)
            assert got2 == b""
        assert int(err) == want and got == decoded[:expect], (name, i)
        assert not r.close()


def test_short_read_empty_stream(eng):
    n, out, err = host.Reader(eng, b"").write_to()
    assert (n, out, int(err)) == (0, b"", 0)


def test_bit_flips_in_header(eng):
    """rd_test.go:26-128."""
    def wt(b): return host.Reader(eng, bytes(b)).write_to()[2]
    b = bytearray(HELLO_FRAME); b[0] ^= 1; e = wt(b); assert int(e) == host.ErrMagic and e.corrupted
    b = bytearray(HELLO_FRAME); b[4] = 0x20; assert int(wt(b)) == host.ErrVersion
    b = bytearray(HELLO_FRAME); b[4] |= 2; e = wt(b); assert int(e) == host.ErrReserveBitSet and e.corrupted
    b = bytearray(HELLO_FRAME); b[5] = 0x30; e = wt(b); assert int(e) == host.ErrBlockDescriptor and e.corrupted
    b = bytearray(HELLO_FRAME); b[6] ^= 1; e = wt(b); assert int(e) == host.ErrHeaderHash and e.corrupted


def test_content_crc(eng):
    """rd_test.go:710-810."""
    for f in (ONE_FRAME, ONE_FRAME_NOHASH):
        n, out, err = host.Reader(eng, f).write_to()
        assert int(err) == 0 and sha(out) == ONE_FRAME_SHA
    bad = bytearray(ONE_FRAME); bad[-1] = (bad[-1] + 1) & 0xFF
    n, out, err = host.Reader(eng, bytes(bad)).write_to()
    assert int(err) == host.ErrContentHash and err.corrupted
    n, out, err = host.Reader(eng, bytes(bad), content_checksum=False).write_to()     # validation disabled
    assert int(err) == 0 and sha(out) == ONE_FRAME_SHA
    n, out, err = host.Reader(eng, ONE_FRAME[:-1]).write_to()
    assert int(err) == host.ErrContentHashRead


def test_block_crc_and_size_overflow(eng):
    """rd_test.go:896-954."""
    payload = synth.text(3 * (64 << 10)).tobytes()
    w = host.Writer(eng, parallel=1, block_size=host.BlockIdx64KB, block_checksum=True)
    assert w.write(payload)[1] == 0 and not w.close()
    f = bytearray(w.output())
    bad = bytearray(f); bad[len(bad) - 12] = (bad[len(bad) - 12] + 1) & 0xFF            # last block's CRC
    n, out, err = host.Reader(eng, bytes(bad)).write_to()
    assert int(err) == host.ErrBlockHash and err.corrupted and out == payload[:2 * (64 << 10)]
    bad = bytearray(f); bad[len(bad) - 8 + 3] |= 0x7F                                   # end mark -> huge block size
    n, out, err = host.Reader(eng, bytes(bad)).write_to()
    assert int(err) == host.ErrBlockSizeOverflow and err.corrupted


def test_content_size_validate(eng):
    """rd_test.go:131-200."""
    zero = MAGIC + bytes([0x68, 0x40, 0, 0, 0, 0, 0, 0, 0, 0, 0x05]) + bytes(4)
    one = MAGIC + bytes([0x68, 0x40, 1, 0, 0, 0, 0, 0, 0, 0, 0x2C, 1, 0, 0, 0x80, 0]) + bytes(4)
    one_with_zero = MAGIC + bytes([0x68, 0x40, 0, 0, 0, 0, 0, 0, 0, 0, 0x05, 1, 0, 0, 0x80, 0]) + bytes(4)
    assert int(host.Reader(eng, zero).write_to()[2]) == 0
    assert int(host.Reader(eng, one).write_to()[2]) == 0
    e = host.Reader(eng, one_with_zero).write_to()[2]
    assert int(e) == host.ErrContentSize and e.corrupted
    assert int(host.Reader(eng, one_with_zero, content_size_check=False).write_to()[2]) == 0


def test_concatenated_and_skippable_frames(eng):
    """wr_test.go:728-848 (reader side): frames back to back, a skippable frame in front."""
    a = host.Writer(eng, parallel=0); a.write(b"first "); a.close()
    b = host.Writer(eng, parallel=1); b.write(b"second"); b.close()
    skip = bytes([0x5A, 0x2A, 0x4D, 0x18]) + (5).to_bytes(4, "little") + b"junk!"
    n, out, err = host.Reader(eng, skip + a.output() + b.output()).write_to()
    assert int(err) == 0 and out == b"first second"
    r = host.Reader(eng, a.output() + b.output())
    got = b""
    while True:
        chunk, err = r.read(4)
        got += chunk
        if err or not chunk:
            break
    assert got == b"first second"


def test_read_small_chunks_matches_write_to(eng):
    payload = synth.make("M", 300000, 64 << 10).tobytes()
    w = host.Writer(eng, parallel=1, block_size=host.BlockIdx64KB, block_checksum=True)
    w.write(payload); w.close()
    r = host.Reader(eng, w.output(), parallel=3)
    got = bytearray()
    while True:
        chunk, err = r.read(7777)
        got += chunk
        if int(err) == host.ErrEOF or (not chunk and not err):
            break
        assert not err
    assert bytes(got) == payload


def test_corrupt_block_payload_is_lz4_corrupted(eng):
    payload = synth.text(100000).tobytes()
    w = host.Writer(eng, parallel=1, block_size=host.BlockIdx64KB, content_checksum=False)
    w.write(payload); w.close()
    f = bytearray(w.output()); f[7 + 4] = 0xFF; f[7 + 5] = 0xFF                        # break the first token / length chain
    n, out, err = host.Reader(eng, bytes(f)).write_to()
    assert err and err.corrupted and int(err) in (host.ErrDecompress, host.ErrContentHash)


# ------------------------------------------------------------------------------------------------ block API (block_test.go:14-353)
def test_block_api(eng, orc):
    assert host.compress_block_bound(1000) == 1000 + 1000 // 255 + 16
    for name, src in corpus.small_cases()[:60] + [("T", synth.text(200000))]:
        c, e = host.compress_block(eng, src)
        assert not e
        want_n, want = orc.compress_fast(src, orc.bound(src.size))
        assert c == want.tobytes()
        if src.size:
            d, e = host.decompress_block(eng, c)
            assert not e and d == src.tobytes()
            d, e = host.decompress_block(eng, c, dst_cap=src.size)
            assert not e and d == src.tobytes()
    src = synth.text(50000)
    c, e = host.compress_block(eng, src, dst_cap=100)                # WithBlockDst too small -> ErrCompress
    assert int(e) == host.ErrCompress
    good, _ = host.compress_block(eng, src)
    d, e = host.decompress_block(eng, good, dst_cap=100)
    assert int(e) == host.ErrDecompress and e.corrupted
    d, e = host.decompress_block(eng, good[:-3])                      # truncated block
    assert int(e) == host.ErrDecompress and e.corrupted
