"""Pin the oracle against every known-answer vector the reference's own tests hold for the path
(SURVEY.md §8c).  Runs on CPU."""
import hashlib

import numpy as np

HELLO_FRAME = bytes.fromhex("04224d18607073060000005068656c6c6f00000000")     # plz4_test.go:12,74

THE_WORKS = bytes([                                                             # internal/test/rd_test.go:527-538
    0x04, 0x22, 0x4d, 0x18,
    0x7d, 0x70, 0x09, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x0b, 0x00, 0x00, 0x00, 0x0c,
    0x06, 0x00, 0x00, 0x00, 0x50, 0x74, 0x65, 0x73, 0x74, 0x79, 0xcf, 0x22, 0x82, 0x16,
    0x05, 0x00, 0x00, 0x00, 0x40, 0x63, 0x6f, 0x64, 0x65, 0xc5, 0x63, 0x71, 0xe5,
    0x00, 0x00, 0x00, 0x00, 0x4a, 0x73, 0x1c, 0xae])

ONE_FRAME = bytes([0x04, 0x22, 0x4d, 0x18, 0x64, 0x40, 0xa7, 0x06, 0x00, 0x00, 0x80, 0x74, 0x65, 0x73, 0x74, 0x79,
                   0x0a, 0x00, 0x00, 0x00, 0x00, 0x5d, 0xc7, 0x3f, 0x2a])     # rd_test.go:714
ONE_FRAME_NOHASH = bytes([0x04, 0x22, 0x4d, 0x18, 0x60, 0x40, 0x82, 0x06, 0x00, 0x00, 0x80, 0x74, 0x65, 0x73, 0x74,
                          0x79, 0x0a, 0x00, 0x00, 0x00, 0x00])                 # rd_test.go:715
ONE_FRAME_SHA = "4e64edc52754ee847f3f043382f70d8cc4f83e38113d3555bdee20442d0d5f50"   # rd_test.go:717

MAGIC = bytes([0x04, 0x22, 0x4d, 0x18])
HEADER_KATS = {                                                                  # header/write_test.go:25-83
    "bsz_4M": (dict(bs_idx=7), [0x60, 0x70, 0x73]),
    "bsz_1M": (dict(bs_idx=6), [0x60, 0x60, 0x51]),
    "bsz_256KB": (dict(bs_idx=5), [0x60, 0x50, 0xfb]),
    "bsz_64KB": (dict(bs_idx=4), [0x60, 0x40, 0x82]),
    "linked": (dict(bs_idx=7, linked=True), [0x40, 0x70, 0xDF]),
    "block_checksum": (dict(bs_idx=7, block_checksum=True), [0x70, 0x70, 0x72]),
    "content_checksum": (dict(bs_idx=7, content_checksum=True), [0x64, 0x70, 0xb9]),
    "content_size": (dict(bs_idx=7, content_size=11), [0x68, 0x70, 0x0B, 0, 0, 0, 0, 0, 0, 0, 0x38]),
    "dict_id": (dict(bs_idx=7, dict_id=6789), [0x61, 0x70, 0x85, 0x1A, 0x00, 0x00, 0xaf]),
    "dict_id+content_size": (dict(bs_idx=7, dict_id=6789, content_size=11),
                             [0x69, 0x70, 0x0B, 0, 0, 0, 0, 0, 0, 0, 0x85, 0x1A, 0x00, 0x00, 0xe2]),
    # rd_test.go:136-139
    "csz_zero_64k": (dict(bs_idx=4, content_size=0), [0x68, 0x40, 0, 0, 0, 0, 0, 0, 0, 0, 0x05]),
    "csz_one_64k": (dict(bs_idx=4, content_size=1), [0x68, 0x40, 1, 0, 0, 0, 0, 0, 0, 0, 0x2C]),
}


def test_xxh32_kats(orc):
    assert orc.xxh32(b"") == 0x02cc5d05
    assert orc.xxh32(b"Ptesty") == 0x168222cf           # block 1 of theWorks
    assert orc.xxh32(b"@code") == 0xe57163c5            # block 2 of theWorks
    assert orc.xxh32(b"testycode") == 0xae1c734a        # content checksum of theWorks
    assert orc.xxh32(b"testy\n") == 0x2a3fc75d          # rd_test.go:714 trailer


def test_xxh32_matches_libxxhash(orc):
    import xxhash
    rng = np.random.default_rng(1)
    for n in list(range(0, 70)) + [255, 256, 257, 4096, 100003]:
        b = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert orc.xxh32(b) == xxhash.xxh32(b, seed=0).intdigest(), n


def test_header_kats(orc):
    for name, (kw, body) in HEADER_KATS.items():
        assert orc.frame_header(**kw) == MAGIC + bytes(body), name


def test_hello_frame(orc):
    """'hello' is kept *compressed although larger than the source* (6 bytes): stored-raw happens only
    when the encoder returns 0 (blk/blk.go:78-92)."""
    src = np.frombuffer(b"hello", dtype=np.uint8)
    got = orc.frame_encode(src, 7, block_checksum=False, content_checksum=False)
    assert got.tobytes() == HELLO_FRAME
    n, out = orc.frame_decode(np.frombuffer(HELLO_FRAME, dtype=np.uint8), 64)
    assert n == 5 and out.tobytes() == b"hello"


def test_the_works_blocks(orc):
    """Block payloads, block checksums and the content checksum of the 54-byte `theWorks` frame."""
    rec1 = orc.block_record(np.frombuffer(b"testy", dtype=np.uint8), 4 << 20, True)
    rec2 = orc.block_record(np.frombuffer(b"code", dtype=np.uint8), 4 << 20, True)
    assert rec1.tobytes() == THE_WORKS[19:33]
    assert rec2.tobytes() == THE_WORKS[33:46]
    hdr = orc.frame_header(7, block_checksum=True, content_checksum=True, content_size=9, dict_id=11)
    assert hdr == THE_WORKS[:19]


def test_stored_block_frames(orc):
    for frame in (ONE_FRAME, ONE_FRAME_NOHASH):
        n, out = orc.frame_decode(np.frombuffer(frame, dtype=np.uint8), 64)
        assert n == 6
        assert hashlib.sha256(out.tobytes()).hexdigest() == ONE_FRAME_SHA
    bad = bytearray(ONE_FRAME); bad[-1] = (bad[-1] + 1) & 0xFF
    n, _ = orc.frame_decode(np.frombuffer(bytes(bad), dtype=np.uint8), 64)
    assert n == -13      # ORC_ERR_CONTENT_HASH


def test_frame_header_errors(orc):
    f = np.frombuffer(HELLO_FRAME, dtype=np.uint8)
    def dec(b): return orc.frame_decode(np.frombuffer(bytes(b), dtype=np.uint8), 64)[0]
    b = bytearray(HELLO_FRAME); b[0] ^= 1; assert dec(b) == -2           # magic
    b = bytearray(HELLO_FRAME); b[4] = 0x20; assert dec(b) == -3         # version
    b = bytearray(HELLO_FRAME); b[4] |= 2; assert dec(b) == -4           # reserved bit
    b = bytearray(HELLO_FRAME); b[5] = 0x30; assert dec(b) == -5         # BD idx < 4
    b = bytearray(HELLO_FRAME); b[6] ^= 1; assert dec(b) == -6           # header checksum
    assert dec(HELLO_FRAME[:9]) == -7                                     # block size read
    assert dec(HELLO_FRAME[:13]) == -9                                    # block read
    assert f.size == 21
