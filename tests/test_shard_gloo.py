"""N > 1 path on CPU: world_size-2 gloo run of the round-robin sharding + framed-output gather (plz4_amd/shard.py).  Each rank
encodes its blocks with the oracle (stand-in for its GPU), rank 0 must end up with exactly the oracle's single-process frame."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _torch_scatter(src, src_off, lens, dst_off, n, max_len, dst):
    for k in range(n):
        l = int(lens[k]); a = int(src_off[k]); d = int(dst_off[k])
        dst[d:d + l] = src[a:a + l]


def _worker(rank, world, port, nblk, bsz, q, piece=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orclib import Oracle
    from plz4_amd import shard, synth
    orc = Oracle()
    data = synth.make("M", nblk * bsz, bsz)
    mine = shard.local_block_ids(nblk, rank, world)
    recs = [orc.block_record(data[g * bsz:(g + 1) * bsz], bsz, True) for g in mine]
    body = torch.from_numpy(np.concatenate(recs))
    rec_len = torch.tensor([r.size for r in recs], dtype=torch.int32)
    kw = {} if piece is None else {"piece": piece}
    frame, total = shard.gather_frame_body(body, rec_len, rank, world, _torch_scatter, {}, bsz + 8, **kw)
    if rank == 0:
        want = orc.frame_encode(data, 4, True, False)[7:-4]
        q.put((total == want.size, bool(np.array_equal(frame[:total].numpy(), want))))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 6, 64 << 10, q)) for r in range(2)]
    for p in procs: p.start()
    ok = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    assert ok == (True, True)
    assert all(p.exitcode == 0 for p in procs)


def test_round_robin_gather_in_pieces_world3():
    """The bodies travel in pieces of whole records through two scratch pieces per sender (here ~2 records per piece, 5 rounds)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, 27, 64 << 10, q, 100000)) for r in range(3)]
    for p in procs: p.start()
    ok = q.get(timeout=180)
    for p in procs: p.join(timeout=60)
    assert ok == (True, True)
    assert all(p.exitcode == 0 for p in procs)


def test_round_robin_gather_uneven_pieces_world4():
    """Four ranks whose bodies differ in size by two orders of magnitude (the M mix has period four in the block index, so with
    four ranks one of them holds only stored blocks, one only runs of zeros): the senders take 1 ... 14 rounds of pieces, the
    owner's receive group shrinks from round to round."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 28, 64 << 10, q, 70000)) for r in range(4)]
    for p in procs: p.start()
    ok = q.get(timeout=240)
    for p in procs: p.join(timeout=60)
    assert ok == (True, True)
    assert all(p.exitcode == 0 for p in procs)


def test_piece_cuts():
    from plz4_amd import shard
    assert shard.piece_cuts([5, 5, 5, 5], 10) == [0, 2, 4]
    assert shard.piece_cuts([50, 5, 5], 10) == [0, 1, 3]              # a record larger than a piece travels alone
    assert shard.piece_cuts([], 10) == [0, 0]


def test_interleave_offsets():
    from plz4_amd import shard
    lens = torch.tensor([[5, 7, 9], [11, 13, 15]], dtype=torch.int32)       # rank 0: blocks 0,2,4; rank 1: blocks 1,3,5
    off, total = shard.interleave_offsets(lens)
    assert total == 60
    assert off.tolist() == [[0, 16, 36], [5, 23, 45]]
    assert shard.local_block_ids(7, 1, 3) == [1, 4]
