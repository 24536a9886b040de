"""The few-block decoder (plz4_amd/csrc/lz4_dx_device.inl: token chain and copy chain cut by pointer jumping) on the lane-emulated
build: whatever it returns must be LZ4_decompress_safe's result (lz4.c:2022-2445) -- and anything it does not want to decide
(-999999) is left to the one-wave decoder, whose parity is tested elsewhere.  GPU: the same through the C ABI, where the hand-over
is inside the call."""
import numpy as np
import pytest

import corpus
from plz4_amd import synth

LEFT = -999999


@pytest.fixture(scope="module")
def emu():
    from emulib import Emu
    return Emu()


def _check(orc, emu, comp, cap, must_take=False):
    a, da = orc.decompress_safe(comp, cap)
    r, out, rounds = emu.dx_decode(comp, cap)
    assert r != -888888, "units disagree"
    if r == LEFT:
        assert not must_take, (comp.size, cap, a)
        return 0
    assert r == a and a >= 0 and np.array_equal(out, da), (comp.size, cap, r, a)      # it only ever answers for blocks that decode
    return 1


def test_emu_dx_valid_blocks(orc, emu):
    taken = total = 0
    for name, src in corpus.small_cases()[::7] + corpus.block_cases_64k() + corpus.twin_cases()[:6]:
        n = src.size
        c, comp = orc.compress_fast(src, orc.bound(n))
        comp = np.ascontiguousarray(comp[:c])
        for cap in (n, n + 8, max(n - 1, 0)):
            taken += _check(orc, emu, comp, cap); total += 1
    assert taken > total // 2


def test_emu_dx_full_size_blocks_take_the_path(orc, emu):
    bsz = 4 << 20
    for kind in ("T", "M", "Z"):
        src = synth.make(kind, bsz, 1 << 16)
        c, comp = orc.compress_fast(src, orc.bound(bsz))
        comp = np.ascontiguousarray(comp[:c])
        r, out, rounds = emu.dx_decode(comp, bsz + 8)
        assert r == bsz and np.array_equal(out, src), kind
        assert 1 <= rounds <= 23, (kind, rounds)
    # a run of one byte: the deepest copy chain there is (4 M pointers, each to the byte before it): log2 rounds
    z = np.zeros(bsz, np.uint8)
    c, comp = orc.compress_fast(z, orc.bound(bsz))
    r, out, rounds = emu.dx_decode(np.ascontiguousarray(comp[:c]), bsz + 8)
    assert r == bsz and not out.any() and 10 <= rounds <= 23


def test_emu_dx_corrupt_blocks(orc, emu):
    rng = np.random.default_rng(11)
    answered = 0
    for seed in range(20):
        n = int(rng.integers(20, 60000))
        src = corpus.structured(n, seed + 500) if seed % 2 else synth.text(n, seed=seed + 1)
        c, comp = orc.compress_fast(src, orc.bound(n))
        comp = comp[:c]
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
                answered += _check(orc, emu, np.ascontiguousarray(bad), cap)
    assert answered > 50            # (a flipped literal byte still decodes: the path answers, with the same wrong bytes as the reference)


def test_emu_dx_corruption_deep_inside_a_large_block(orc, emu):
    bsz = 1 << 20
    rng = np.random.default_rng(23)
    for kind in ("T", "M"):
        src = synth.make(kind, bsz, 1 << 16)
        c, comp = orc.compress_fast(src, orc.bound(bsz))
        comp = np.ascontiguousarray(comp[:c])
        for at in (c // 7, c // 3, c // 2, c - 70000, c - 300, c - 20):
            for k in range(4):
                bad = comp.copy()
                i = min(at + int(rng.integers(0, 64)), c - 2)
                if k == 0: bad[i] ^= 1 << int(rng.integers(0, 8))
                elif k == 1: bad[i] = 0xFF
                elif k == 2: bad[i:i + 2] = 0
                else: bad = bad[:i]
                _check(orc, emu, np.ascontiguousarray(bad), bsz + 8)


@pytest.fixture(scope="module")
def eng():
    from plz4_amd._native import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
def test_gpu_dx_few_full_size_blocks(orc, eng):
    """decompress_batch of 1, 3 and 16 blocks of 4 MiB takes the few-block path (k_dx_tables .. k_dx_gather); T, M, Z and R blocks
    (R: one literal run to the block's end -- handed to the one-wave decoder inside the call) come back as they went in."""
    bsz = 4 << 20
    blocks = []
    for i, kind in enumerate(("T", "M", "Z", "R") * 4):
        src = synth.make(kind, bsz, 1 << 16, seed=100 + i)
        c, comp = orc.compress_fast(src, orc.bound(bsz))
        blocks.append((src, np.ascontiguousarray(comp[:c])))
    for nb in (1, 3, 16):
        res, outs = eng.decompress_batch([c for _, c in blocks[:nb]], [bsz + 8] * nb)
        for (src, _), r, o in zip(blocks, res, outs):
            assert int(r) == bsz and np.array_equal(o, src)
    # tight and short capacities: the reference's verdict (an error code for the short ones), whichever path gives it
    for cap in (bsz, bsz - 1, bsz // 2):
        res, outs = eng.decompress_batch([c for _, c in blocks[:4]], [cap] * 4)
        for (src, comp), r, o in zip(blocks, res, outs):
            a, da = orc.decompress_safe(comp, cap)
            assert int(r) == a and (a < 0 or np.array_equal(o, da)), (cap, int(r), a)


@pytest.mark.gpu
def test_gpu_dx_corrupt_blocks(orc, eng):
    """Corrupt blocks large enough for the few-block path, a handful per call: the call's results are LZ4_decompress_safe's, codes
    included (the path answers only for blocks that decode; the rest is decoded again by the one-wave decoder)."""
    rng = np.random.default_rng(5)
    comps, caps = [], []
    for seed in range(6):
        n = int(rng.integers(60000, 400000))
        src = corpus.structured(n, seed + 900) if seed % 2 else synth.text(n, seed=seed + 3)
        c, comp = orc.compress_fast(src, orc.bound(n))
        comp = comp[:c]
        for trial in range(24):
            bad = comp.copy()
            k = int(rng.integers(0, 5))
            if k == 0: bad = bad[:int(rng.integers(bad.size // 2, bad.size))]
            elif k == 1: i = int(rng.integers(0, bad.size)); bad[i] ^= 1 << int(rng.integers(0, 8))
            elif k == 2: i = int(rng.integers(0, bad.size)); bad[i] = 0xFF
            elif k == 3: i = int(rng.integers(0, bad.size)); bad[i:i + 2] = 0
            comps.append(np.ascontiguousarray(bad)); caps.append(n + 8 if trial % 3 else n)
    nbad = 0
    for lo in range(0, len(comps), 12):
        res, outs = eng.decompress_batch(comps[lo:lo + 12], caps[lo:lo + 12])
        for cmp_, cap, r, o in zip(comps[lo:lo + 12], caps[lo:lo + 12], res, outs):
            a, da = orc.decompress_safe(cmp_, cap)
            assert int(r) == a, (cmp_.size, cap, int(r), a)
            if a >= 0: assert np.array_equal(o, da)
            nbad += a < 0
    assert nbad > 30


@pytest.mark.gpu
def test_gpu_dx_few_records(orc, eng):
    """decode_records of a handful of 4 MiB records (what a reader with few blocks in flight hands over): compressed records take
    the few-block path with their block checksums verified beside it; a stored record, a record whose checksum does not match, one
    whose payload is damaged behind a matching checksum and one with an oversized size word get the one-wave kernel's verdict."""
    bsz = 4 << 20
    srcs = [synth.make(k, bsz, 1 << 16, seed=200 + i) for i, k in enumerate(("T", "M", "R", "T", "Z", "T"))]
    recs = [np.ascontiguousarray(orc.block_record(s, bsz, True)) for s in srcs]
    res, st, outs = eng.decode_records(recs, bsz, True)
    for s, r, k, o in zip(srcs, res, st, outs):
        assert int(k) == 0 and int(r) == bsz and np.array_equal(o, s)
    bad = [r.copy() for r in recs]
    bad[0][1000] ^= 0x10                                        # checksum mismatch
    p = bad[3][4:-4].copy(); p[5000] ^= 0x01                    # damaged payload, checksum recomputed: LZ4's own verdict
    bad[3] = np.concatenate([bad[3][:4], p, np.frombuffer(np.uint32(orc.xxh32(np.ascontiguousarray(p))).tobytes(), np.uint8)])
    bad[5] = bad[5].copy(); bad[5][:4] = np.frombuffer(np.uint32(bsz + 1).tobytes(), np.uint8)      # size word beyond the block size
    res, st, outs = eng.decode_records([np.ascontiguousarray(b) for b in bad], bsz, True)
    assert int(st[0]) != 0 and int(st[5]) != 0 and int(st[0]) != int(st[5])
    a, da = orc.decompress_safe(np.ascontiguousarray(p), bsz + 8)
    assert (int(st[3]) == 0 and int(res[3]) == a and np.array_equal(outs[3], da)) if a >= 0 else int(st[3]) != 0
    for i in (1, 2, 4):
        assert int(st[i]) == 0 and np.array_equal(outs[i], srcs[i])
