"""Pin the oracle (our CPU restatement) against the REAL reference: liblz4 v1.10.0 compiled unmodified from
/root/reference/internal/pkg/clz4 into oracle/_ref/liblz4ref.so.  Byte-exact encode, and exact return
codes + bytes for decode, over seeded corpora that cover the reference's edge cases."""
import numpy as np
import pytest

import corpus
from plz4_amd import synth


def _same_encode(orc, ref, src, cap):
    a, da = orc.compress_fast(src, cap)
    b, db = ref.compress_fast(src, cap)
    assert a == b, (src.size, cap, a, b)
    assert np.array_equal(da, db)
    return b, db


def test_encode_small(orc, ref):
    for name, src in corpus.small_cases():
        n = src.size
        for cap in (orc.bound(n), n, max(n - 1, 0), n + 8, 4 << 20):
            _same_encode(orc, ref, src, cap)


def test_encode_64k_boundary(orc, ref):
    for name, src in corpus.block_cases_64k():
        n = src.size
        for cap in (orc.bound(n), n, 4 << 20):
            _same_encode(orc, ref, src, cap)


@pytest.mark.parametrize("kind", ["T", "R", "Z", "M"])
def test_encode_4m_blocks(orc, ref, kind):
    bsz = 4 << 20
    data = synth.make(kind, 2 * bsz + 12345, bsz)
    for off in range(0, data.size, bsz):
        blk = data[off:off + bsz]
        _same_encode(orc, ref, blk, bsz)               # frame path: cap == bsz (blk/blk.go:73)
    _same_encode(orc, ref, data[:bsz], orc.bound(bsz))  # block API: cap == bound (plz4_block.go:105)


def test_encode_structured_sweep(orc, ref):
    for seed in range(40):
        n = int(np.random.default_rng(seed).integers(13, 300000))
        src = corpus.structured(n, seed)
        _same_encode(orc, ref, src, n)
        _same_encode(orc, ref, src, orc.bound(n))


def test_limited_output_threshold(orc, ref):
    """The stored-block decision is liblz4's conservative early-out, not `size > cap` (SURVEY §8a trap 1):
    sweep the capacity through the whole interesting range around the compressed size."""
    for seed in (1, 2, 3):
        src = corpus.structured(5000, seed)
        full, _ = ref.compress_fast(src, orc.bound(src.size))
        for cap in range(max(full - 40, 0), full + 40):
            _same_encode(orc, ref, src, cap)


def _same_decode(orc, ref, comp, cap):
    a, da = orc.decompress_safe(comp, cap)
    b, db = ref.decompress_safe(comp, cap)
    assert a == b, (comp.size, cap, a, b)
    if b >= 0:
        assert np.array_equal(da, db)
    return b


def test_decode_valid(orc, ref):
    for name, src in corpus.small_cases() + corpus.block_cases_64k():
        n = src.size
        c, comp = ref.compress_fast(src, orc.bound(n))
        for cap in (n, n + 8, n + 64, max(n - 1, 0), max(n - 13, 0)):
            _same_decode(orc, ref, comp, cap)


def test_decode_corrupt(orc, ref):
    """Return codes must agree exactly (they carry the failing input position, lz4.c:2443)."""
    rng = np.random.default_rng(5)
    nbad = 0
    for seed in range(30):
        n = int(rng.integers(20, 20000))
        src = corpus.structured(n, seed + 100)
        c, comp = ref.compress_fast(src, orc.bound(n))
        comp = comp.copy()
        for trial in range(40):
            bad = comp.copy()
            k = int(rng.integers(0, 4))
            if k == 0:
                bad = bad[:int(rng.integers(0, bad.size))]
            elif k == 1:
                i = int(rng.integers(0, bad.size)); bad[i] ^= 1 << int(rng.integers(0, 8))
            elif k == 2:
                i = int(rng.integers(0, bad.size)); bad[i] = 0xFF
            else:
                i = int(rng.integers(0, bad.size)); bad[i:i + 2] = 0
            for cap in (n, n + 8):
                if bad.size == 0:
                    continue
                r = _same_decode(orc, ref, np.ascontiguousarray(bad), cap)
                nbad += r < 0
    assert nbad > 500


def test_decode_offset_zero_and_special(orc, ref):
    # token 0x1F: 1 literal 'A', match len 15+ext; offset 0 -> liblz4 zero-fills (lz4.c:499-507)
    blk = np.array([0x14, 0x41, 0x00, 0x00] + [0x50, 1, 2, 3, 4, 5], dtype=np.uint8)
    for cap in (14, 15, 100):
        _same_decode(orc, ref, blk, cap)
    # empty block / empty dst special cases (lz4.c:2064-2069)
    _same_decode(orc, ref, np.array([0], dtype=np.uint8), 0)
    _same_decode(orc, ref, np.array([0], dtype=np.uint8), 10)
    _same_decode(orc, ref, np.array([0, 0], dtype=np.uint8), 0)
    _same_decode(orc, ref, np.array([0x10, 0x41], dtype=np.uint8), 0)


def test_stream_linked_and_dict(orc, ref):
    """Config 5 arithmetic: LZ4_loadDict[Slow] / attach_dictionary / compress_fast_continue and
    LZ4_decompress_safe_usingDict, emulating clz4.go:96-120,160-179,211-248."""
    import ctypes as C
    from orclib import OrcStream, _ptr
    dct = synth.text(70000, seed=99)
    data = synth.text(6 * 65536 + 777, seed=7)

    # reference dict ctx (LZ4_loadDictSlow) + oracle twin
    rbuf, rdict = ref.new_stream()
    ref.L.LZ4_resetStream_fast(rdict)
    ref.L.LZ4_loadDictSlow(rdict, _ptr(dct), dct.size)
    odict = OrcStream(); orc.L.orc_stream_init(C.byref(odict))
    orc.L.orc_stream_reset_fast(C.byref(odict))
    orc.L.orc_stream_load_dict(C.byref(odict), _ptr(dct), dct.size, 1)

    # --- independent blocks with dictionary (StreamIndieCtx): sizes on both sides of the 4 KiB switch
    rsb, rs = ref.new_stream(); os_ = OrcStream(); orc.L.orc_stream_init(C.byref(os_))
    for n in (0, 5, 100, 4096, 4097, 65536, 200000):
        src = np.ascontiguousarray(data[:n])
        cap = max(n, 16)
        d1 = np.empty(cap, dtype=np.uint8); d2 = np.empty(cap, dtype=np.uint8)
        ref.L.LZ4_resetStream_fast(rs); ref.L.LZ4_attach_dictionary(rs, rdict)
        a = ref.L.LZ4_compress_fast_continue(rs, _ptr(src), _ptr(d1), n, cap, 1)
        orc.L.orc_stream_reset_fast(C.byref(os_)); orc.L.orc_stream_attach(C.byref(os_), C.byref(odict))
        b = orc.L.orc_stream_compress(C.byref(os_), _ptr(src), n, _ptr(d2), cap)
        assert a == b, n
        assert np.array_equal(d1[:a], d2[:b])
        if a > 0:
            dd = np.ascontiguousarray(dct[-65536:])
            ra, outa = ref.decompress_safe_dict(np.ascontiguousarray(d1[:a]), n + 8, dd)
            rb, outb = orc.decompress_safe_dict(np.ascontiguousarray(d1[:a]), n + 8, dd)
            assert ra == rb == n and np.array_equal(outa, outb) and np.array_equal(outa, src)

    # --- linked blocks: block 0 uses the attached dict ctx, later blocks LZ4_loadDict(prev tail)
    for bsz in (65536, 100000):
        rsb, rs = ref.new_stream(); ref.L.LZ4_resetStream_fast(rs); ref.L.LZ4_attach_dictionary(rs, rdict)
        os_ = OrcStream(); orc.L.orc_stream_init(C.byref(os_)); orc.L.orc_stream_reset_fast(C.byref(os_))
        orc.L.orc_stream_attach(C.byref(os_), C.byref(odict))
        prev = None
        for off in range(0, data.size, bsz):
            src = np.ascontiguousarray(data[off:off + bsz]); n = src.size
            d1 = np.empty(n + 16, dtype=np.uint8); d2 = np.empty(n + 16, dtype=np.uint8)
            if prev is not None:
                tail = np.ascontiguousarray(prev[-65536:]).copy()
                ref.L.LZ4_loadDict(rs, _ptr(tail), tail.size)
                orc.L.orc_stream_load_dict(C.byref(os_), _ptr(tail), tail.size, 0)
            a = ref.L.LZ4_compress_fast_continue(rs, _ptr(src), _ptr(d1), n, n, 1)
            b = orc.L.orc_stream_compress(C.byref(os_), _ptr(src), n, _ptr(d2), n)
            assert a == b and np.array_equal(d1[:a], d2[:b]), (bsz, off)
            prev = src
