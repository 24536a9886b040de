"""plz4hip_mgpu (section D of include/plz4hip.h): blocks dealt round-robin over several ctxs -- here 2 and 3 entries on the one
device of the GPU box, which exercises everything but a physically different peer: the dealing, the per-device threads, the
frame assembly on the owner through the scratch pieces (the records of the "other" entries travel by hipMemcpyPeerAsync),
the dealing back for decode.  Expected bytes: the oracle's frame for the same data."""
import numpy as np
import pytest

from plz4_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("g", [2, 3])
def test_mgpu_host_buffers(orc, g):
    from plz4_amd._native import MultiEngine
    m = MultiEngine([0] * g)
    bsz = 256 << 10
    data = synth.make("M", 23 * bsz + 777, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)] + [np.zeros(0, np.uint8), np.frombuffer(b"hello", dtype=np.uint8)]
    recs = m.encode_records(srcs, bsz, True)
    for s, r in zip(srcs, recs):
        assert np.array_equal(r, orc.block_record(s, bsz, True)), s.size
    res, st, outs = m.decode_records([np.ascontiguousarray(r) for r in recs], bsz, True)
    for s, r, k, o in zip(srcs, res, st, outs):
        assert int(k) == 0 and int(r) == s.size and np.array_equal(o, s)
    caps = [orc.bound(s.size) for s in srcs]
    res, outs = m.compress_batch(srcs, caps)
    for s, cap, r, o in zip(srcs, caps, res, outs):
        n, want = orc.compress_fast(s, cap)
        assert int(r) == n and np.array_equal(o, want[:n])
    assert m.encode_records([], bsz, True) == []
    m.close()


@pytest.mark.parametrize("g,owner,peer", [(2, 0, False), (3, 2, True), (2, 1, True)])
def test_mgpu_device_frame(orc, g, owner, peer, monkeypatch):
    """Shards on the device(s) -> the frame body on the owner == the oracle frame's block section; and back.  peer: the other
    entries' records go the way a different GPU's would (peer copy into the two scratch pieces, 3 MiB each here, then the
    record mover) although they sit on the same device."""
    import torch
    from plz4_amd._native import MultiEngine
    if peer:
        monkeypatch.setenv("PLZ4HIP_MGPU_FORCE_PEER", "1")
        monkeypatch.setenv("PLZ4HIP_MGPU_PIECE_KB", "3072")
    m = MultiEngine([0] * g)
    bsz = 1 << 20
    data = synth.make("M", 19 * bsz + 4321, bsz)
    nblk = (data.size + bsz - 1) // bsz
    frame = orc.frame_encode(data, 6, block_checksum=True, content_checksum=False)        # BD index 6 = 1 MiB
    body_want = frame[7:-4]
    dev = torch.device("cuda:0")
    blocks = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    shards = [torch.from_numpy(np.concatenate(blocks[k::g])).to(dev) for k in range(g)]
    body = torch.zeros(data.size + 8 * nblk, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    off, total = m.dev_encode_frame([t.data_ptr() for t in shards], [t.numel() for t in shards], nblk, bsz, True, owner,
                                    body.data_ptr(), body.numel())
    assert total == body_want.size and int(off[-1]) == total
    assert np.array_equal(body[:total].cpu().numpy(), body_want)
    outs = [torch.zeros(len(blocks[k::g]) * bsz, dtype=torch.uint8, device=dev) for k in range(g)]
    res, st = m.dev_decode_frame(nblk, owner, body.data_ptr(), off, bsz, True, [t.data_ptr() for t in outs], bsz, bsz)
    torch.cuda.synchronize()
    assert int(np.abs(st).sum()) == 0 and int(res.sum()) == data.size
    for k in range(g):
        want = np.concatenate(blocks[k::g])
        assert np.array_equal(outs[k][:want.size].cpu().numpy(), want)
    m.close()
