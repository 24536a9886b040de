"""GPU parity proper: the HIP path, called through the C ABI (include/plz4hip.h), against the oracle on the same
seeded inputs.  Bit-exact for every byte and every return code.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest

import corpus
from plz4_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from plz4_amd._native import Engine
    e = Engine(0)
    yield e
    e.close()


def _check_encode(orc, eng, srcs, caps):
    res, outs = eng.compress_batch(srcs, caps)
    for i, (s, c) in enumerate(zip(srcs, caps)):
        a, da = orc.compress_fast(s, c)
        assert int(res[i]) == a, (i, s.size, c, int(res[i]), a)
        assert np.array_equal(outs[i], da), (i, s.size, c)


def test_gpu_encode_small(orc, eng):
    srcs, caps = [], []
    for name, src in corpus.small_cases():
        n = src.size
        for cap in (orc.bound(n), n, max(n - 1, 0), n + 8):
            srcs.append(src); caps.append(cap)
    _check_encode(orc, eng, srcs, caps)


def test_gpu_encode_64k_boundary(orc, eng):
    srcs, caps = [], []
    for name, src in corpus.block_cases_64k():
        for cap in (orc.bound(src.size), src.size):
            srcs.append(src); caps.append(cap)
    _check_encode(orc, eng, srcs, caps)


def test_gpu_encode_structured(orc, eng):
    srcs, caps = [], []
    for seed in range(40):
        n = int(np.random.default_rng(seed).integers(13, 300000))
        src = corpus.structured(n, seed)
        srcs += [src, src]; caps += [n, orc.bound(n)]
    _check_encode(orc, eng, srcs, caps)


def test_gpu_encode_twins(orc, eng):
    srcs, caps = [], []
    for name, src in corpus.twin_cases():
        srcs += [src, src]; caps += [src.size, orc.bound(src.size)]
    _check_encode(orc, eng, srcs, caps)


def test_gpu_encode_limited_threshold(orc, eng):
    src = corpus.structured(5000, 2)
    full, _ = orc.compress_fast(src, orc.bound(src.size))
    caps = list(range(max(full - 40, 0), full + 40))
    _check_encode(orc, eng, [src] * len(caps), caps)


@pytest.mark.parametrize("kind", ["T", "R", "Z", "M"])
def test_gpu_encode_4m(orc, eng, kind):
    bsz = 4 << 20
    data = synth.make(kind, 4 * bsz + 70001, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    _check_encode(orc, eng, srcs, [bsz] * len(srcs))           # frame path: cap == bsz (blk/blk.go:73)
    _check_encode(orc, eng, srcs[:1], [orc.bound(bsz)])         # block API: cap == bound (plz4_block.go:105)


def test_gpu_encode_4m_against_the_reference_itself(ref, orc, eng):
    """The level-1 GPU tests above compare with the oracle restatement (itself pinned to the reference on the CPU); here one 4 MiB
    T and one M block go against LZ4_compress_fast of oracle/_ref directly, at both capacities plz4 uses."""
    bsz = 4 << 20
    srcs = [synth.make("T", bsz, bsz), synth.make("M", 2 * bsz, bsz)[bsz:]]
    for cap in (bsz, orc.bound(bsz)):
        res, outs = eng.compress_batch(srcs, [cap] * len(srcs))
        for s, r, o in zip(srcs, res, outs):
            n, want = ref.compress_fast(s, cap)
            assert int(r) == n and np.array_equal(o, want[:n]), (cap, int(r), n)


def test_gpu_encode_over_4m(orc, eng):
    """Blocks above 4 MiB (raw block API only) run the table without tags."""
    src = synth.text((5 << 20) + 123, seed=9)
    _check_encode(orc, eng, [src], [orc.bound(src.size)])


def _check_decode(orc, eng, comps, caps):
    res, outs = eng.decompress_batch(comps, caps)
    nbad = 0
    for i, (cmp_, cap) in enumerate(zip(comps, caps)):
        a, da = orc.decompress_safe(cmp_, cap)
        assert int(res[i]) == a, (i, cmp_.size, cap, int(res[i]), a)
        if a >= 0:
            assert np.array_equal(outs[i], da), i
        nbad += a < 0
    return nbad


def test_gpu_decode_valid(orc, eng):
    comps, caps = [], []
    for name, src in corpus.small_cases() + corpus.block_cases_64k():
        n = src.size
        c, comp = orc.compress_fast(src, orc.bound(n))
        comp = np.ascontiguousarray(comp)
        for cap in (n, n + 8, n + 64, max(n - 1, 0), max(n - 13, 0)):
            comps.append(comp); caps.append(cap)
    _check_decode(orc, eng, comps, caps)


def test_gpu_decode_corrupt(orc, eng):
    rng = np.random.default_rng(11)
    comps, caps = [], []
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
                comps.append(np.ascontiguousarray(bad)); caps.append(cap)
    assert _check_decode(orc, eng, comps, caps) > 300


def test_gpu_decode_special(orc, eng):
    blk = np.array([0x14, 0x41, 0x00, 0x00] + [0x50, 1, 2, 3, 4, 5], dtype=np.uint8)    # offset 0 -> zero fill
    one = np.array([0], dtype=np.uint8)
    _check_decode(orc, eng, [blk, blk, blk, one, one, np.array([0, 0], dtype=np.uint8)], [14, 15, 100, 0, 10, 0])


def test_gpu_decode_4m(orc, eng):
    bsz = 4 << 20
    comps, srcs = [], []
    for kind in ("T", "Z", "M", "R"):
        src = synth.make(kind, bsz, bsz)
        c, comp = orc.compress_fast(src, orc.bound(bsz))
        comps.append(np.ascontiguousarray(comp)); srcs.append(src)
    res, outs = eng.decompress_batch(comps, [bsz + 8] * len(comps))        # plz4 decodes into bsz+8 (blk/blk.go:51-53)
    for r, o, s in zip(res, outs, srcs):
        assert int(r) == bsz and np.array_equal(o, s)


def test_gpu_decode_corrupt_4m(orc, eng):
    """Corruption deep inside a 4 MiB block: the LDS-staged vector decoder must stop where LZ4_decompress_safe stops, with the
    same code (or accept what it accepts, with the same bytes)."""
    bsz = 4 << 20
    rng = np.random.default_rng(23)
    comps, caps = [], []
    for kind in ("T", "M"):
        src = synth.make(kind, bsz, bsz)
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
                comps.append(np.ascontiguousarray(bad)); caps.append(bsz + 8)
    nbad = _check_decode(orc, eng, comps, caps)
    assert nbad > 10


def test_gpu_xxh32(orc, eng):
    rng = np.random.default_rng(3)
    bufs = [rng.integers(0, 256, size=n, dtype=np.uint8)
            for n in list(range(0, 70)) + [127, 128, 129, 255, 1000, 4096, 100003, 1 << 20, (4 << 20) + 3]]
    got = eng.xxh32_batch(bufs)
    for b, g in zip(bufs, got):
        assert int(g) == orc.xxh32(b), b.size


@pytest.mark.parametrize("bsz", [64 << 10, 256 << 10, 1 << 20, 4 << 20])      # descriptor/index.go:5-38, idx 4..7
@pytest.mark.parametrize("checksum", [False, True])
def test_gpu_records(orc, eng, bsz, checksum):
    """blk.CompressToBlk / FrameReader+BlkT.Decompress on the device == oracle, incl. the stored-raw rule."""
    data = synth.make("M", 5 * bsz + 4321, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)] + [np.frombuffer(b"hello", dtype=np.uint8),
                                                                   np.zeros(0, dtype=np.uint8)]
    recs = eng.encode_records(srcs, bsz, checksum)
    for s, r in zip(srcs, recs):
        want = orc.block_record(s, bsz, checksum)
        assert np.array_equal(r, want), s.size
    res, st, outs = eng.decode_records([np.ascontiguousarray(r) for r in recs], bsz, checksum)
    for s, r, k, o in zip(srcs, res, st, outs):
        assert int(k) == 0 and int(r) == s.size and np.array_equal(o, s)
    if checksum:
        bad = recs[0].copy(); bad[10] ^= 1
        res, st, _ = eng.decode_records([np.ascontiguousarray(bad)], bsz, True)
        assert int(st[0]) == 1                                   # PLZ4HIP_BLK_HASH_MISMATCH
    over = recs[0].copy(); over[0:4] = np.frombuffer(np.uint32(bsz + 1).tobytes(), dtype=np.uint8)
    res, st, _ = eng.decode_records([np.ascontiguousarray(over)], bsz, False)
    assert int(st[0]) == 2                                       # PLZ4HIP_BLK_SIZE_OVERFLOW


def test_gpu_records_that_inflate_past_the_block_size(ref, orc, eng):
    """The reference decodes a block into the whole pooled buffer, bsz + 8 bytes (blk/pool.go:23-26, blk/blk.go:51-53): a block that
    inflates to bsz + 1 .. bsz + 8 bytes is ACCEPTED, bsz + 9 is liblz4's output-overflow error.  Same here, same codes."""
    bsz = 64 << 10
    base = synth.text(bsz + 64, seed=3)
    recs, wants = [], []
    for extra in (0, 1, 7, 8, 9, 16):
        plain = np.ascontiguousarray(base[:bsz + extra])
        n, comp = ref.compress_fast(plain, orc.bound(plain.size))
        assert 0 < n <= bsz                                              # (a size word above bsz is a different error)
        rec = np.concatenate([np.frombuffer(np.uint32(n).tobytes(), dtype=np.uint8), comp[:n]])
        recs.append(np.ascontiguousarray(rec))
        wants.append((plain, ref.decompress_safe(np.ascontiguousarray(comp[:n]), bsz + 8)[0]))
    res, st, outs = eng.decode_records(recs, bsz, False)
    for (plain, code), r, k, o, extra in zip(wants, res, st, outs, (0, 1, 7, 8, 9, 16)):
        assert int(r) == code, (extra, int(r), code)
        if extra <= 8:
            assert int(k) == 0 and code == plain.size and np.array_equal(o, plain), extra
        else:
            assert code < 0 and int(k) != 0, (extra, code, int(k))


def test_gpu_dev_pipeline(orc, eng):
    """Device-resident encode -> scan -> compaction gives exactly the oracle frame's block section, and the
    device decoder restores the plaintext (config 2 / config 3 plumbing at a size the oracle handles quickly)."""
    import torch
    bsz = 4 << 20
    data = synth.make("M", 6 * bsz + 12345, bsz)
    nblk = (data.size + bsz - 1) // bsz
    frame = orc.frame_encode(data, 7, block_checksum=True, content_checksum=False)
    body_want = frame[7:-4]
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    stride = eng.stage_stride(bsz)
    d_stage = torch.empty(nblk * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(nblk, dtype=torch.int32, device=dev)
    d_off = torch.zeros(nblk + 1, dtype=torch.int64, device=dev)
    d_body = torch.empty(data.size + 8 * nblk, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    eng.dev_encode_records(d_src.data_ptr(), data.size, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), s)
    eng.dev_compact_records(d_stage.data_ptr(), stride, d_len.data_ptr(), nblk, d_off.data_ptr(),
                            d_body.data_ptr(), d_body.numel(), s)
    torch.cuda.synchronize()
    total = int(d_off[-1].item())
    assert total == body_want.size
    assert np.array_equal(d_body[:total].cpu().numpy(), body_want)
    d_out = torch.zeros(nblk * bsz, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(nblk, dtype=torch.int32, device=dev)
    d_st = torch.zeros(nblk, dtype=torch.int32, device=dev)
    eng.dev_decode_records(d_body.data_ptr(), d_off.data_ptr(), nblk, bsz, True, d_out.data_ptr(), bsz, bsz,
                           d_res.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    assert int(d_res.sum().item()) == data.size
    assert np.array_equal(d_out[:data.size].cpu().numpy(), data)


def test_gpu_host_calls_in_many_chunks(orc, monkeypatch):
    """The host-buffer entry points cut a large call into chunks that overlap on the GPU; a tiny chunk size forces many
    chunks (and the slot ring to wrap) on a small input: results must not depend on it."""
    from plz4_amd._native import Engine
    monkeypatch.setenv("PLZ4HIP_HOST_CHUNK_MB", "1")
    e = Engine(0)
    bsz = 64 << 10
    data = synth.make("T", 45 * bsz + 777, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)] + [np.zeros(0, dtype=np.uint8), np.frombuffer(b"hello", dtype=np.uint8)]
    srcs.insert(7, synth.make("R", bsz, bsz))                         # a stored-raw block in the middle
    recs = e.encode_records(srcs, bsz, True)
    for s, r in zip(srcs, recs):
        assert np.array_equal(r, orc.block_record(s, bsz, True)), s.size
    res, st, outs = e.decode_records([np.ascontiguousarray(r) for r in recs], bsz, True)
    for s, r, k, o in zip(srcs, res, st, outs):
        assert int(k) == 0 and int(r) == s.size and np.array_equal(o, s)
    caps = [orc.bound(s.size) if i % 3 else max(s.size // 2, 1) for i, s in enumerate(srcs)]
    res, outs = e.compress_batch(srcs, caps)
    for s, cap, r, o in zip(srcs, caps, res, outs):
        want_n, want = orc.compress_fast(s, cap)
        assert int(r) == want_n and (want_n == 0 or np.array_equal(o, want[:want_n])), (s.size, cap)
    comp = [o for r, o in zip(res, outs) if int(r) > 0]
    plain = [s for r, s in zip(res, srcs) if int(r) > 0]
    res2, outs2 = e.decompress_batch([np.ascontiguousarray(x) for x in comp], [p.size for p in plain])
    for p, r, o in zip(plain, res2, outs2):
        assert int(r) == p.size and np.array_equal(o, p)
    e.close()


@pytest.mark.parametrize("budget_mib,bsz,bsz_b", [(0, 4 << 20, 4 << 20), (40, 4 << 20, 4 << 20), (0, 256 << 10, 64 << 10)])
def test_gpu_duplex_call(orc, monkeypatch, budget_mib, bsz, bsz_b):
    """plz4hip_dev_duplex_records == plz4hip_dev_encode_records of one batch + plz4hip_dev_decode_records of another (k_l1_duplex:
    one parser wave and one decoder wave per workgroup by default): the records are the oracle's, the decoded blocks the other batch's
    plaintext with every status OK -- a damaged record reports as it does in the plain call -- also when the encode side runs in
    groups (the decode rides in the first one) and when either side is empty."""
    import torch
    from plz4_amd._native import Engine
    if budget_mib:
        monkeypatch.setenv("PLZ4HIP_L1_BUDGET_MIB", str(budget_mib))
    e = Engine(0)
    dataA = synth.make("M", 6 * bsz + 4321, bsz)
    dataB = synth.make("T", 9 * bsz_b + 77, bsz_b)
    nA = (dataA.size + bsz - 1) // bsz; nB = (dataB.size + bsz_b - 1) // bsz_b
    wantA = np.concatenate([orc.block_record(dataA[o:o + bsz], bsz, True) for o in range(0, dataA.size, bsz)])
    recsB = [orc.block_record(dataB[o:o + bsz_b], bsz_b, True) for o in range(0, dataB.size, bsz_b)]
    recsB[4] = recsB[4].copy(); recsB[4][100] ^= 0x40                       # block 4: its checksum no longer matches
    offB = np.zeros(nB + 1, dtype=np.int64); offB[1:] = np.cumsum([r.size for r in recsB])
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(dataA).to(dev)
    stride = e.stage_stride(bsz)
    d_stage = torch.zeros(nA * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(nA, dtype=torch.int32, device=dev)
    d_off = torch.zeros(nA + 1, dtype=torch.int64, device=dev)
    d_bodyA = torch.empty(dataA.size + 8 * nA, dtype=torch.uint8, device=dev)
    d_bodyB = torch.from_numpy(np.concatenate(recsB)).to(dev)
    d_offB = torch.from_numpy(offB).to(dev)
    d_out = torch.zeros(nB * bsz_b, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(nB, dtype=torch.int32, device=dev)
    d_st = torch.full((nB,), -9, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def check_encode():
        e.dev_compact_records(d_stage.data_ptr(), stride, d_len.data_ptr(), nA, d_off.data_ptr(), d_bodyA.data_ptr(), d_bodyA.numel(), s)
        torch.cuda.synchronize()
        total = int(d_off[-1].item())
        assert total == wantA.size and np.array_equal(d_bodyA[:total].cpu().numpy(), wantA)

    def check_decode():
        st = d_st.cpu().numpy(); res = d_res.cpu().numpy(); out = d_out.cpu().numpy()
        for i in range(nB):
            blk = dataB[i * bsz_b:(i + 1) * bsz_b]
            if i == 4:
                assert st[i] != 0                                            # (PLZ4HIP_BLK_HASH_MISMATCH)
            else:
                assert st[i] == 0 and res[i] == blk.size and np.array_equal(out[i * bsz_b:i * bsz_b + blk.size], blk), i

    e.dev_duplex_records(d_src.data_ptr(), dataA.size, bsz, True, d_stage.data_ptr(), d_len.data_ptr(),
                         d_bodyB.data_ptr(), d_offB.data_ptr(), nB, bsz_b, True, d_out.data_ptr(), bsz_b, bsz_b, d_res.data_ptr(), d_st.data_ptr(), s)
    check_encode(); check_decode()
    want_st = d_st.clone()
    e.dev_decode_records(d_bodyB.data_ptr(), d_offB.data_ptr(), nB, bsz_b, True, d_out.data_ptr(), bsz_b, bsz_b, d_res.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(want_st, d_st)                                        # the same status codes as the plain call
    # either side empty
    d_stage.zero_(); d_len.zero_(); d_out.zero_(); d_st.fill_(-9)
    e.dev_duplex_records(d_src.data_ptr(), dataA.size, bsz, True, d_stage.data_ptr(), d_len.data_ptr(),
                         d_bodyB.data_ptr(), d_offB.data_ptr(), 0, bsz_b, True, d_out.data_ptr(), bsz_b, bsz_b, d_res.data_ptr(), d_st.data_ptr(), s)
    check_encode()
    assert int((d_st != -9).sum().item()) == 0
    e.dev_duplex_records(d_src.data_ptr(), 0, bsz, True, d_stage.data_ptr(), d_len.data_ptr(),
                         d_bodyB.data_ptr(), d_offB.data_ptr(), nB, bsz_b, True, d_out.data_ptr(), bsz_b, bsz_b, d_res.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    check_decode()
    e.close()


def test_gpu_duplex_call_on_a_chip_full_of_blocks(orc):
    """k_l1_duplex at the size the bench runs it in kind: 320 x 4 MiB blocks each way -- more than one block per CU, ten-wave
    occupancy on part of the chip, persistent waves taking several blocks each -- with every record compared with the oracle's
    (blk.CompressToBlk, blk/blk.go:69-109) and every decoded block with its plaintext.  The small duplex test runs 6-9 blocks."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from plz4_amd._native import Engine
    e = Engine(0)
    bsz = 4 << 20
    nb = 320
    pool = synth.make("T", 16 * bsz, bsz)
    pool[5 * bsz:6 * bsz] = synth.make("R", bsz, bsz)                       # one stored block per 16
    pool[9 * bsz:9 * bsz + (bsz >> 1)] = 0                                  # one with a 2 MiB run (long matches, 0xFF length bytes)
    dev = torch.device("cuda:0")
    d_pool = torch.from_numpy(pool).to(dev)
    d_src = torch.empty(nb * bsz, dtype=torch.uint8, device=dev)
    for r in range(nb // 16):
        d_src[r * 16 * bsz:(r + 1) * 16 * bsz] = torch.roll(d_pool, -((r * 1000003) % pool.size)) if r else d_pool
    src = d_src.cpu().numpy()
    with ThreadPoolExecutor(8) as ex:                                         # (the oracle releases the GIL inside ctypes calls)
        want = list(ex.map(lambda i: orc.block_record(src[i * bsz:(i + 1) * bsz], bsz, True), range(nb)))
    stride = e.stage_stride(bsz)
    d_stage = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(nb, dtype=torch.int32, device=dev)
    off = np.zeros(nb + 1, dtype=np.int64); off[1:] = np.cumsum([w.size for w in want])
    d_body = torch.from_numpy(np.concatenate(want)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_out = torch.zeros(nb * bsz, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_st = torch.full((nb,), -9, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    e.dev_duplex_records(d_src.data_ptr(), nb * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(),
                         d_body.data_ptr(), d_off.data_ptr(), nb, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    lens = d_len.cpu().numpy(); stage = d_stage.cpu().numpy()
    for i, w in enumerate(want):
        assert int(lens[i]) == w.size and np.array_equal(stage[i * stride:i * stride + w.size], w), i
    assert int(d_st.abs().sum().item()) == 0 and int(d_res.to(torch.int64).sum().item()) == nb * bsz
    assert torch.equal(d_out, d_src)
    e.close()


@pytest.mark.parametrize("budget_mib,level", [(0, 1), (40, 1), (0, 2)])
def test_gpu_records_straight_into_the_frame_body(orc, ref, monkeypatch, budget_mib, level):
    """plz4hip_dev_encode_body / plz4hip_dev_duplex_body: the records land back to back -- the frame's block section, byte for byte
    what blk.CompressToBlk + the writer's in-order emission produce (blk/blk.go:69-109) -- with recOff / recLen as
    plz4hip_dev_compact_records makes them; also when the call runs in groups (the scan continues across them), with and without
    block checksums, into a body that is too small (what fits is written, recOff says the rest), and at level 2."""
    import torch
    from plz4_amd._native import Engine
    if budget_mib:
        monkeypatch.setenv("PLZ4HIP_L1_BUDGET_MIB", str(budget_mib))
    e = Engine(0)
    bsz = 4 << 20
    data = synth.make("M", 6 * bsz + 4321, bsz)
    nb = (data.size + bsz - 1) // bsz
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    for cs in (True, False):
        if level == 1:
            want = [orc.block_record(data[o:o + bsz], bsz, cs) for o in range(0, data.size, bsz)]
        else:
            want = []
            for o in range(0, data.size, bsz):
                blk = data[o:o + bsz]
                r, c = ref.compress_hc(blk, bsz, 2)
                payload, word = (blk, 0x80000000 | blk.size) if r == 0 else (c[:r], r)
                rec = np.uint32(word).tobytes() + payload.tobytes() + (np.uint32(orc.xxh32(np.ascontiguousarray(payload))).tobytes() if cs else b"")
                want.append(np.frombuffer(rec, np.uint8))
        wantBody = np.concatenate(want)
        d_body = torch.zeros(wantBody.size + 100, dtype=torch.uint8, device=dev)
        d_off = torch.full((nb + 1,), -1, dtype=torch.int64, device=dev)
        d_len = torch.zeros(nb, dtype=torch.int32, device=dev)
        e.dev_encode_body(d_src.data_ptr(), data.size, bsz, cs, d_body.data_ptr(), d_body.numel(), d_off.data_ptr(), d_len.data_ptr(), s, level=level)
        torch.cuda.synchronize()
        off = d_off.cpu().numpy(); ln = d_len.cpu().numpy()
        assert [int(x) for x in ln] == [w.size for w in want] and int(off[0]) == 0 and int(off[-1]) == wantBody.size
        assert np.array_equal(np.diff(off), ln) and np.array_equal(d_body[:wantBody.size].cpu().numpy(), wantBody)
        # too small a body: the records that fit are there, recOff tells the whole length
        small = int(off[3]) + 10
        d_body2 = torch.zeros(small, dtype=torch.uint8, device=dev)
        e.dev_encode_body(d_src.data_ptr(), data.size, bsz, cs, d_body2.data_ptr(), small, d_off.data_ptr(), d_len.data_ptr(), s, level=level)
        torch.cuda.synchronize()
        assert int(d_off[-1].item()) == wantBody.size and np.array_equal(d_body2[:int(off[3])].cpu().numpy(), wantBody[:int(off[3])])
    if level == 1:
        # the duplex form: this batch's body beside the decode of another body
        dataB = synth.make("T", 5 * bsz + 77, bsz)
        recsB = [orc.block_record(dataB[o:o + bsz], bsz, True) for o in range(0, dataB.size, bsz)]
        nB = len(recsB)
        offB = np.zeros(nB + 1, dtype=np.int64); offB[1:] = np.cumsum([r.size for r in recsB])
        d_bodyB = torch.from_numpy(np.concatenate(recsB)).to(dev); d_offB = torch.from_numpy(offB).to(dev)
        d_out = torch.zeros(nB * bsz, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(nB, dtype=torch.int32, device=dev); d_st = torch.full((nB,), -9, dtype=torch.int32, device=dev)
        want = [orc.block_record(data[o:o + bsz], bsz, True) for o in range(0, data.size, bsz)]
        wantBody = np.concatenate(want)
        d_body = torch.zeros(wantBody.size, dtype=torch.uint8, device=dev)
        e.dev_duplex_body(d_src.data_ptr(), data.size, bsz, True, d_body.data_ptr(), d_body.numel(), d_off.data_ptr(), d_len.data_ptr(),
                          d_bodyB.data_ptr(), d_offB.data_ptr(), nB, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), s)
        torch.cuda.synchronize()
        assert int(d_off[-1].item()) == wantBody.size and np.array_equal(d_body.cpu().numpy(), wantBody)
        assert int(d_st.abs().sum().item()) == 0 and np.array_equal(d_out[:dataB.size].cpu().numpy(), dataB)
    e.close()


def test_gpu_level1_in_groups(orc, monkeypatch):
    """The staged level-1 call keeps 9 bytes of workspace per possible sequence of the blocks of one group; a call that does not
    fit the memory set aside runs in groups of equal size.  A 40 MiB budget makes 4 MiB blocks go four to a group: records and raw
    blocks of a 7-block call (one of them incompressible, the last one short) must not depend on it -- device-resident and host calls."""
    import torch
    from plz4_amd._native import Engine
    monkeypatch.setenv("PLZ4HIP_L1_BUDGET_MIB", "40")
    e = Engine(0)
    bsz = 4 << 20
    data = synth.make("M", 6 * bsz + 54321, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    recs = e.encode_records(srcs, bsz, True)
    for s_, r in zip(srcs, recs):
        assert np.array_equal(r, orc.block_record(s_, bsz, True)), s_.size
    res, outs = e.compress_batch(srcs, [orc.bound(s_.size) for s_ in srcs])
    for s_, r, o in zip(srcs, res, outs):
        n, want = orc.compress_fast(s_, orc.bound(s_.size))
        assert int(r) == n and np.array_equal(o, want[:n])
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    stride = e.stage_stride(bsz)
    d_stage = torch.zeros(len(srcs) * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(len(srcs), dtype=torch.int32, device=dev)
    e.dev_encode_records(d_src.data_ptr(), data.size, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lens = d_len.cpu().numpy(); stage = d_stage.cpu().numpy()
    for i, r in enumerate(recs):
        assert int(lens[i]) == r.size and np.array_equal(stage[i * stride:i * stride + r.size], r), i
    e.close()


def test_gpu_dev_compress_levels_and_trim(ref, orc):
    """plz4hip_dev_compress (raw blocks on the device) at level 1, a hash-chain level and level 12; plz4hip_ctx_trim gives the
    staging and HC workspaces back and the ctx keeps working."""
    import ctypes as C
    import torch
    from plz4_amd._native import Engine
    e = Engine(0)
    dev = torch.device("cuda:0")
    n = 300000
    srcs = [synth.text(n, seed=3), synth.make("M", n, 65536), np.zeros(n, np.uint8), synth.text(4097, seed=4)]
    stride = 1 << 19
    cap = orc.bound(n)
    d_src = torch.zeros(len(srcs) * stride, dtype=torch.uint8, device=dev)
    for i, s in enumerate(srcs):
        d_src[i * stride:i * stride + s.size] = torch.from_numpy(s).to(dev)
    d_len = torch.tensor([s.size for s in srcs], dtype=torch.int32, device=dev)
    d_cap = torch.full((len(srcs),), cap, dtype=torch.int32, device=dev)
    for lvl in (1, 5, 12):
        d_dst = torch.zeros(len(srcs) * stride, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(len(srcs), dtype=torch.int32, device=dev)
        e._chk(e.L.plz4hip_dev_compress(e.h, len(srcs), d_src.data_ptr(), stride, d_len.data_ptr(), d_dst.data_ptr(), stride,
                                        d_cap.data_ptr(), lvl, n, d_res.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        res = d_res.cpu().numpy(); out = d_dst.cpu().numpy()
        for i, s in enumerate(srcs):
            want_n, want = (orc.compress_fast(s, cap) if lvl == 1 else ref.compress_hc(s, cap, lvl))
            assert int(res[i]) == want_n and np.array_equal(out[i * stride:i * stride + want_n], want[:want_n]), (lvl, i)
        if lvl == 5:
            e.trim()                                                             # HC workspace gone; the next call allocates again
    # an understated maxLen: the lengths live on the device and the workspaces are sized from maxLen, so a block longer than that
    # must not be touched -- its result is PLZ4HIP_E_ARG, its neighbours are unaffected (every level's kernels)
    for lvl in (1, 5, 12):
        d_dst = torch.zeros(len(srcs) * stride, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(len(srcs), dtype=torch.int32, device=dev)
        e._chk(e.L.plz4hip_dev_compress(e.h, len(srcs), d_src.data_ptr(), stride, d_len.data_ptr(), d_dst.data_ptr(), stride,
                                        d_cap.data_ptr(), lvl, 8192, d_res.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        res = d_res.cpu().numpy(); out = d_dst.cpu().numpy()
        assert [int(r) for r in res[:3]] == [-1, -1, -1], (lvl, res)             # PLZ4HIP_E_ARG
        want_n, want = (orc.compress_fast(srcs[3], cap) if lvl == 1 else ref.compress_hc(srcs[3], cap, lvl))
        assert int(res[3]) == want_n and np.array_equal(out[3 * stride:3 * stride + want_n], want[:want_n]), lvl
    recs = e.encode_records(srcs, 1 << 19, True)
    e.trim()
    recs2 = e.encode_records(srcs, 1 << 19, True)
    assert all(np.array_equal(a, b) for a, b in zip(recs, recs2))
    e.close()


@pytest.mark.gpu
def test_gpu_dev_compress_on_two_streams_of_one_ctx(ref, orc):
    """Two plz4hip_dev_compress jobs with different block counts and lengths enqueued back to back on two streams of one ctx
    (include/plz4hip.h documents that as supported): each job's kernels must see its own sanitised lengths -- the ctx-wide copy
    is ordered across streams like the workspaces."""
    import torch
    from plz4_amd._native import Engine
    e = Engine(0)
    dev = torch.device("cuda:0")
    stride = 1 << 19
    jobs = []
    for j, (cnt, n) in enumerate(((24, 300000), (7, 70001))):
        srcs = [synth.text(n - 13 * i, seed=20 + 7 * j + i) for i in range(cnt)]
        d_src = torch.zeros(cnt * stride, dtype=torch.uint8, device=dev)
        for i, s in enumerate(srcs):
            d_src[i * stride:i * stride + s.size] = torch.from_numpy(s).to(dev)
        jobs.append({"srcs": srcs, "src": d_src, "len": torch.tensor([s.size for s in srcs], dtype=torch.int32, device=dev),
                     "cap": torch.full((cnt,), orc.bound(n), dtype=torch.int32, device=dev), "n": n, "stream": torch.cuda.Stream(device=dev)})
    torch.cuda.synchronize()
    for lvl in (1, 4):
        for rep in range(3):
            for jb in jobs:
                jb["dst"] = torch.zeros(len(jb["srcs"]) * stride, dtype=torch.uint8, device=dev)
                jb["res"] = torch.zeros(len(jb["srcs"]), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for jb in (jobs if rep % 2 == 0 else jobs[::-1]):
                e._chk(e.L.plz4hip_dev_compress(e.h, len(jb["srcs"]), jb["src"].data_ptr(), stride, jb["len"].data_ptr(), jb["dst"].data_ptr(), stride,
                                                jb["cap"].data_ptr(), lvl, jb["n"], jb["res"].data_ptr(), jb["stream"].cuda_stream))
            torch.cuda.synchronize()
            for jb in jobs:
                res = jb["res"].cpu().numpy(); out = jb["dst"].cpu().numpy()
                for i, s in enumerate(jb["srcs"]):
                    cap = orc.bound(jb["n"])
                    want_n, want = (orc.compress_fast(s, cap) if lvl == 1 else ref.compress_hc(s, cap, lvl))
                    assert int(res[i]) == want_n and np.array_equal(out[i * stride:i * stride + want_n], want[:want_n]), (lvl, rep, i)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nstreams,starve", [(2, False), (3, False), (2, True)])
def test_gpu_duplex_body_calls_alternating_over_streams_of_one_ctx(orc, nstreams, starve):
    """The bench's pipelines in small: plz4hip_dev_duplex_body calls of DIFFERENT batches enqueued back to back on two (three)
    streams of one ctx -- the ctx has two record workspaces, the parse of a call waits on the device behind k_parse_gate for the
    parse before it, the emit kernels of one call run beside the parse of the next (a third stream shares a workspace: ordered by
    an event).  Every call's frame body must be its own batch's records (blk.CompressToBlk, blk/blk.go:69-109) and every decode
    side the plaintext of the body the same stream's call before it wrote."""
    import torch
    from plz4_amd._native import Engine
    e = Engine(0)
    bsz = 1 << 20
    dev = torch.device("cuda:0")
    rounds = 3
    pipes = []
    for p in range(nstreams):
        batches = []
        for r in range(rounds + 1):
            kind = "TMT"[(p + r) % 3]
            data = synth.make(kind, (40 + 7 * p + 3 * r) * bsz + 1234 * (r + 1), bsz)
            data = np.roll(data, 1000 * (p * 5 + r) + 17)
            recs = [orc.block_record(data[o:o + bsz], bsz, True) for o in range(0, data.size, bsz)]
            batches.append({"data": data, "d_src": torch.from_numpy(data).to(dev), "want": np.concatenate(recs), "nb": len(recs)})
        cap = max(b["want"].size for b in batches) + 64
        nbmax = max(b["nb"] for b in batches)
        pipes.append({"batches": batches, "stream": torch.cuda.Stream(device=dev),
                      "bodies": [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(rounds + 1)],
                      "offs": [torch.zeros(nbmax + 1, dtype=torch.int64, device=dev) for _ in range(rounds + 1)],
                      "lens": [torch.zeros(nbmax, dtype=torch.int32, device=dev) for _ in range(rounds + 1)],
                      "outs": [torch.zeros(nbmax * bsz, dtype=torch.uint8, device=dev) for _ in range(rounds + 1)],
                      "res": [torch.zeros(nbmax, dtype=torch.int32, device=dev) for _ in range(rounds + 1)],
                      "st": [torch.full((nbmax,), -9, dtype=torch.int32, device=dev) for _ in range(rounds + 1)]})
    torch.cuda.synchronize()
    hog = None
    if starve:
        # no room for a second record workspace (it is taken whole or not at all): the calls of the second stream share the first
        # one's and wait for it -- the results are the same
        e.dev_encode_body(pipes[0]["batches"][0]["d_src"].data_ptr(), pipes[0]["batches"][0]["data"].size, bsz, True, pipes[0]["bodies"][0].data_ptr(),
                          pipes[0]["bodies"][0].numel(), pipes[0]["offs"][0].data_ptr(), pipes[0]["lens"][0].data_ptr(), pipes[0]["stream"].cuda_stream, level=1)
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info(dev)
        hog = torch.empty(max(free - (48 << 20), 1 << 20), dtype=torch.uint8, device=dev)
    for r in range(rounds + 1):                       # every call of a round is enqueued before anything is waited for
        for pp in pipes:
            b = pp["batches"][r]; s = pp["stream"].cuda_stream
            if r == 0:
                e.dev_encode_body(b["d_src"].data_ptr(), b["data"].size, bsz, True, pp["bodies"][0].data_ptr(), pp["bodies"][0].numel(),
                                  pp["offs"][0].data_ptr(), pp["lens"][0].data_ptr(), s, level=1)
            else:
                prev = pp["batches"][r - 1]
                e.dev_duplex_body(b["d_src"].data_ptr(), b["data"].size, bsz, True, pp["bodies"][r].data_ptr(), pp["bodies"][r].numel(),
                                  pp["offs"][r].data_ptr(), pp["lens"][r].data_ptr(),
                                  pp["bodies"][r - 1].data_ptr(), pp["offs"][r - 1].data_ptr(), prev["nb"], bsz, True,
                                  pp["outs"][r].data_ptr(), bsz, bsz, pp["res"][r].data_ptr(), pp["st"][r].data_ptr(), s)
    torch.cuda.synchronize()
    for p, pp in enumerate(pipes):
        for r in range(rounds + 1):
            b = pp["batches"][r]
            assert int(pp["offs"][r][b["nb"]].item()) == b["want"].size, (p, r)
            assert np.array_equal(pp["bodies"][r][:b["want"].size].cpu().numpy(), b["want"]), (p, r)
            if r > 0:
                prev = pp["batches"][r - 1]
                assert int(pp["st"][r][:prev["nb"]].abs().sum().item()) == 0, (p, r)
                assert np.array_equal(pp["outs"][r][:prev["data"].size].cpu().numpy(), prev["data"]), (p, r)
    del hog, pipes, pp
    e.close()
    torch.cuda.synchronize(); torch.cuda.empty_cache()      # (the memory goes back to the device, not into this process's cache: the next test starts other processes)


@pytest.mark.gpu
def test_gpu_bench_modes_run_on_one_gpu():
    """bench.py's other code paths on the one GPU of the box: the N > 1 path (process group, gather stream, size all-gather,
    interleave on rank 0) with a single rank (PLZ4_BENCH_FORCE_GATHER), and the decode-only mode of configs[2]."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLZ4_BENCH_FORCE_GATHER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--blocks", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "frame_gather" in line["ms"]
    # (N > 1: a step is two calls, one half of the blocks on each of two streams; the exchanges run a step behind)
    assert line["config"]["pipelines"] == 2 and line["config"]["calls_per_step"] == 2 and sum(line["config"]["blocks_per_call"]) == 64
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--blocks", "64", "--steps", "1", "--warmup", "1", "--decode-only"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert "configs[2]" in line["config"]["workload"] and line["value"] > 0 and line["roofline"]["kernel"] == "k_decode_rec"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--blocks", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                          "--gather", "none"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["gather"] == "none" and line["ms"]["frame_gather"] < 0.5 and "gathered_frame" not in line["memory_gib"]["plan"]
    assert "memory plan, GiB, for 8 ranks" in out.stderr                      # the dry run logs what rank 0 of a full node would hold
    for dup in ("1", "0"):                                                   # the duplex step and the serial one, gather path included
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--blocks", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                              "--duplex", dup], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["value"] > 0 and ("k_l1_duplex" in line["roofline"]["kernel"]) == (dup == "1")
    # one GPU, no gather: the duplex steps alternate over two streams by default (two record workspaces inside the library), three
    # when asked, one when asked; an odd number of steps leaves the pipelines with different numbers of calls
    for pipes, want in (((), 2), (("--pipelines", "3"), 3), (("--pipelines", "1"), 1)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--blocks", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                              *pipes], env=dict(os.environ, PLZ4_BENCH_NO_PIPE_CHECK="1"), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["value"] > 0 and line["config"]["pipelines"] == want and line["serial_step"]["value"] > 0
        assert (want > 1) == ("pipeline_check" in line["config"])
    # the check that clocks the pipelined steps against one stream before the timed region: its two ways out (new streams; one
    # stream), forced
    for forced, want in (("1", 2), ("2", 1)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--blocks", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                             env=dict(os.environ, PLZ4_BENCH_TEST_PIPE_CHECK=forced), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        # (64 blocks per step: whether two streams beat one here is noise, so a forced first failure may be followed by a real one)
        rep = line["config"]["pipeline_check"]["streams_replaced"]
        assert rep >= int(forced) and line["config"]["pipelines"] == (2 if rep < 2 else 1) and (forced != "2" or line["config"]["pipelines"] == want)
