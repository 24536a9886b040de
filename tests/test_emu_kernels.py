"""The device code (plz4_amd/csrc/lz4_device.inl), compiled for CPU by the lane-emulation harness, against the
oracle.  Same source hipcc builds for gfx950; this covers the kernel LOGIC without a GPU (the GPU parity tests
in test_gpu_parity.py run the real thing).  Both resolutions of same-slot LDS store conflicts are exercised."""
import numpy as np
import pytest

import corpus
from plz4_amd import synth


@pytest.fixture(scope="module")
def emu():
    from emulib import Emu
    return Emu()


@pytest.fixture(params=[False, True], ids=["lanes_asc", "lanes_desc"])
def emu_o(emu, request):
    emu.set_descending(request.param)
    yield emu
    emu.set_descending(False)


def _enc(orc, emu, src, cap):
    a, da = orc.compress_fast(src, cap)
    b, db = emu.compress_fast(src, cap)
    assert a == b, (src.size, cap, a, b)
    assert np.array_equal(da, db), (src.size, cap)


def test_emu_encode_small(orc, emu_o):
    for name, src in corpus.small_cases():
        n = src.size
        for cap in (orc.bound(n), n, max(n - 1, 0), n + 8):
            _enc(orc, emu_o, src, cap)


def test_emu_encode_64k_boundary(orc, emu_o):
    for name, src in corpus.block_cases_64k():
        for cap in (orc.bound(src.size), src.size):
            _enc(orc, emu_o, src, cap)


def test_emu_encode_structured(orc, emu_o):
    for seed in range(25):
        n = int(np.random.default_rng(seed).integers(13, 200000))
        src = corpus.structured(n, seed)
        _enc(orc, emu_o, src, n)
        _enc(orc, emu_o, src, orc.bound(n))


def test_emu_encode_twins(orc, emu_o):
    for name, src in corpus.twin_cases():
        _enc(orc, emu_o, src, src.size)
        _enc(orc, emu_o, src, orc.bound(src.size))


def test_emu_encode_limited_threshold(orc, emu):
    src = corpus.structured(5000, 2)
    full, _ = orc.compress_fast(src, orc.bound(src.size))
    for cap in range(max(full - 30, 0), full + 30):
        _enc(orc, emu, src, cap)


@pytest.mark.parametrize("kind", ["T", "R", "Z", "M"])
def test_emu_encode_4m(orc, emu, kind):
    bsz = 4 << 20
    data = synth.make(kind, bsz + 70001, bsz)
    _enc(orc, emu, data[:bsz], bsz)
    _enc(orc, emu, data[bsz:], bsz)


def test_emu_encode_over_4m(orc, emu):
    """Blocks above 4 MiB (raw block API only) run the table without tags."""
    src = synth.text((5 << 20) + 123, seed=9)
    _enc(orc, emu, src, orc.bound(src.size))


def _dec(orc, emu, comp, cap):
    a, da = orc.decompress_safe(comp, cap)
    b, db = emu.decompress_safe(comp, cap)
    assert a == b, (comp.size, cap, a, b)
    if a >= 0:
        assert np.array_equal(da, db)
    return a


def test_emu_decode_valid(orc, emu):
    for name, src in corpus.small_cases() + corpus.block_cases_64k():
        n = src.size
        c, comp = orc.compress_fast(src, orc.bound(n))
        comp = np.ascontiguousarray(comp)
        for cap in (n, n + 8, n + 64, max(n - 1, 0), max(n - 13, 0)):
            _dec(orc, emu, comp, cap)


def test_emu_decode_corrupt(orc, emu):
    rng = np.random.default_rng(11)
    nbad = 0
    for seed in range(20):
        n = int(rng.integers(20, 20000))
        src = corpus.structured(n, seed + 500)
        c, comp = orc.compress_fast(src, orc.bound(n))
        for trial in range(40):
            bad = comp.copy()
            k = int(rng.integers(0, 4))
            if k == 0:
                bad = bad[:int(rng.integers(1, bad.size))]
            elif k == 1:
                i = int(rng.integers(0, bad.size)); bad[i] ^= 1 << int(rng.integers(0, 8))
            elif k == 2:
                i = int(rng.integers(0, bad.size)); bad[i] = 0xFF
            else:
                i = int(rng.integers(0, bad.size)); bad[i:i + 2] = 0
            for cap in (n, n + 8):
                nbad += _dec(orc, emu, np.ascontiguousarray(bad), cap) < 0
    assert nbad > 300


def test_emu_decode_special(orc, emu):
    blk = np.array([0x14, 0x41, 0x00, 0x00] + [0x50, 1, 2, 3, 4, 5], dtype=np.uint8)   # offset 0 -> zero fill
    for cap in (14, 15, 100):
        _dec(orc, emu, blk, cap)
    _dec(orc, emu, np.array([0], dtype=np.uint8), 0)
    _dec(orc, emu, np.array([0], dtype=np.uint8), 10)
    _dec(orc, emu, np.array([0, 0], dtype=np.uint8), 0)


def test_emu_decode_4m(orc, emu):
    bsz = 4 << 20
    for kind in ("T", "Z", "M"):
        src = synth.make(kind, bsz, bsz)
        c, comp = orc.compress_fast(src, orc.bound(bsz))
        n, out = emu.decompress_safe(np.ascontiguousarray(comp), bsz + 8)     # plz4 decodes into bsz+8 (blk/blk.go:51-53)
        assert n == bsz and np.array_equal(out, src)


def test_emu_xxh32(orc, emu):
    rng = np.random.default_rng(3)
    for n in list(range(0, 70)) + [127, 128, 129, 255, 1000, 4096, 100003, 1 << 20]:
        a = rng.integers(0, 256, size=n, dtype=np.uint8)
        assert emu.xxh32(a) == orc.xxh32(a), n


# ------------------------------------------------------------------------------------------------ config 5: dictionary / linked modes
def _dict_table(dctx):
    return np.ctypeslib.as_array(dctx.table).astype(np.uint32).copy()


def test_emu_encode_indie_with_dictionary(orc, emu):
    """StreamIndieCtx (clz4.go:160-179): > 4 KiB copies the dictionary context's table (mode 2), <= 4 KiB looks it up (mode 3)."""
    dct_full = synth.text(70000, seed=99)
    data = synth.text(300000, seed=7)
    for dct_user in (dct_full, dct_full[:30000], dct_full[:100], dct_full[:5]):
        dctx = orc.dict_ctx(dct_user)
        dct = np.ascontiguousarray(dct_user[-65536:])
        tab = _dict_table(dctx)
        for n in (0, 5, 12, 13, 100, 4095, 4096, 4097, 65536, 200000):
            src = np.ascontiguousarray(data[:n])
            for cap in (max(n, 1), n + n // 255 + 16):
                a, da = orc.compress_indie_dict(src, cap, dctx)
                mode = 4 if dct.size < 8 else (2 if n > 4096 else 3)
                b, db = emu.compress_dict(src, cap, dct if mode != 4 else None, mode, tab)
                assert a == b and np.array_equal(da, db), (dct.size, n, cap, a, b)


def test_emu_encode_linked(orc, emu):
    """StreamLinkedCtx (clz4.go:224-248): block 0 fresh (mode 0), later blocks LZ4_loadDict(previous tail) (mode 1, or 4 if < 8 bytes)."""
    data = synth.make("M", 5 * 65536 + 777, 65536, seed=3)
    for bsz in (65536, 100000):
        prev = None
        for off in range(0, data.size, bsz):
            src = np.ascontiguousarray(data[off:off + bsz]); n = src.size
            tail = None if prev is None else prev[-65536:].copy()      # a separate buffer, like the pooled dict block (async/writer.go:412-437)
            a, da = orc.compress_linked(src, n, tail)
            mode = 0 if tail is None else (1 if tail.size >= 8 else 4)
            b, db = emu.compress_dict(src, n, tail if mode == 1 else None, mode)
            assert a == b and np.array_equal(da, db), (bsz, off, a, b)
            prev = src
    # tiny previous block (flush of a few bytes): LZ4_loadDict drops it (lz4.c:1613-1615)
    src = np.ascontiguousarray(data[:5000]); tail = data[5000:5005].copy()
    a, da = orc.compress_linked(src, 5000, tail)
    b, db = emu.compress_dict(src, 5000, None, 4)
    assert a == b and np.array_equal(da, db)


def test_emu_both_dictionary_encoders_agree(orc, emu):
    """The external-segment mode of the full encoder (grid batches) and the one-sequence-per-batch dictionary encoder are two
    restatements of the same reference path: same bytes, both lane orders, linked blocks and dictionary contexts."""
    user = synth.text(40000, seed=5)
    dctx = orc.dict_ctx(user); tab = _dict_table(dctx); dct = np.ascontiguousarray(user[-65536:])
    data = corpus.structured(3 * 70000, 11)
    blocks = [np.ascontiguousarray(data[o:o + 70000]) for o in range(0, data.size, 70000)]
    try:
        for desc in (False, True):
            emu.set_descending(desc)
            prev = None
            for b in blocks:
                cases = [(2, dct, tab)] if prev is None else [(1, np.ascontiguousarray(prev[-65536:]), None), (1, np.ascontiguousarray(prev[-9:]), None)]
                for mode, seg, t in cases:
                    emu.set_old_dict(False); r1, d1 = emu.compress_dict(b, b.size, seg, mode, t)
                    emu.set_old_dict(True);  r2, d2 = emu.compress_dict(b, b.size, seg, mode, t)
                    assert r1 == r2 and np.array_equal(d1, d2), (desc, mode, seg.size)
                prev = b
    finally:
        emu.set_old_dict(False); emu.set_descending(False)


def test_emu_decode_with_dictionary(orc, emu):
    dct_user = synth.text(70000, seed=99)
    dct = np.ascontiguousarray(dct_user[-65536:])
    dctx = orc.dict_ctx(dct_user)
    data = synth.text(200000, seed=7)
    rng = np.random.default_rng(4)
    for n in (100, 4096, 5000, 65536, 200000):
        src = np.ascontiguousarray(data[:n])
        c, comp = orc.compress_indie_dict(src, n + n // 255 + 16, dctx)
        comp = np.ascontiguousarray(comp)
        for d in (dct, np.ascontiguousarray(dct[-1000:])):          # full window and a short one (checkOffset on)
            for cap in (n, n + 8, n - 1):
                a, da = orc.decompress_safe_dict(comp, cap, d)
                b, db = emu.decompress_dict(comp, cap, d)
                assert a == b, (n, d.size, cap, a, b)
                if a >= 0:
                    assert np.array_equal(da, db)
        for t in range(20):
            bad = comp.copy(); bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
            a, da = orc.decompress_safe_dict(bad, n + 8, dct)
            b, db = emu.decompress_dict(bad, n + 8, dct)
            assert a == b
            if a >= 0:
                assert np.array_equal(da, db)
