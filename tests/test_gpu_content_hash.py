"""The streaming content checksum on the device (plz4hip_xxh32_stream_*) against xxh32.XXHZero's
behaviour as the oracle restates it: a stream written in many pieces, host and device bytes, Sum32 in the middle of a stream, and
the plaintext of encode_records / decode_records calls (several staging chunks) fed on the device."""
import numpy as np
import pytest

from plz4_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from plz4_amd._native import Engine
    e = Engine(0)
    yield e
    e.close()


def test_gpu_xxh32_stream_pieces(orc, eng):
    rng = np.random.default_rng(7)
    h = eng.hash_create()
    assert eng.hash_sum(h) == orc.xxh32(np.zeros(0, np.uint8))                    # empty stream: 0x02cc5d05
    for trial in range(12):
        eng.hash_reset(h)
        pieces = [rng.integers(0, 256, int(rng.integers(0, [5, 17, 40, 1000, 70000, 3 << 20][int(rng.integers(0, 6))])), dtype=np.uint8)
                  for _ in range(int(rng.integers(1, 9)))]
        sofar = np.zeros(0, np.uint8)
        for k, p in enumerate(pieces):
            eng.hash_update(h, p)
            sofar = np.concatenate([sofar, p])
            if k % 3 == 1:
                assert eng.hash_sum(h) == orc.xxh32(sofar), (trial, k, sofar.size)    # Sum32 does not disturb the stream
        assert eng.hash_sum(h) == orc.xxh32(sofar), (trial, sofar.size)
    eng.hash_destroy(h)


def test_gpu_xxh32_stream_device_bytes(orc, eng):
    import torch
    data = synth.make("M", (9 << 20) + 13, 1 << 20)
    d = torch.from_numpy(data).to("cuda:0")
    h = eng.hash_create()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    cuts = [0, 5, 5 + (3 << 20), 5 + (3 << 20) + 17, data.size]
    for k in range(len(cuts) - 1):                                                   # alternate streams: updates still follow each other
        st = (s1, s2)[k & 1]
        eng.dev_hash_update(h, d.data_ptr() + cuts[k], cuts[k + 1] - cuts[k], st.cuda_stream)
    assert eng.hash_sum(h) == orc.xxh32(data)
    eng.hash_destroy(h)


def test_gpu_content_hash_of_record_calls(orc, monkeypatch):
    """encode_records / decode_records with a content-hash stream set on the ctx: the plaintext is hashed on the device, in block
    order, across staging chunks (PLZ4HIP_HOST_CHUNK_MB=1 forces many) and across calls."""
    from plz4_amd._native import Engine
    monkeypatch.setenv("PLZ4HIP_HOST_CHUNK_MB", "1")
    e = Engine(0)
    bsz = 256 << 10
    data = synth.make("M", 37 * bsz + 4321, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    h = e.hash_create()
    e.set_content_hash(h)
    recs = e.encode_records(srcs[:20], bsz, True) + e.encode_records(srcs[20:], bsz, True)
    assert e.hash_sum(h) == orc.xxh32(data)
    for s, r in zip(srcs, recs):
        assert np.array_equal(r, orc.block_record(s, bsz, True))
    e.hash_reset(h)
    res, st, outs = e.decode_records([np.ascontiguousarray(r) for r in recs], bsz, True)
    assert int(np.abs(st).sum()) == 0 and np.array_equal(np.concatenate(outs), data)
    assert e.hash_sum(h) == orc.xxh32(data)
    e.set_content_hash(None)
    e.hash_reset(h)
    e.encode_records(srcs[:3], bsz, True)
    assert e.hash_sum(h) == orc.xxh32(np.zeros(0, np.uint8))                         # cleared: nothing was written
    e.hash_destroy(h)
    e.close()
