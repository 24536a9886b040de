"""Multi-GPU sharding of the per-block path: block i of the stream lives on rank i mod G; encode and decode
share nothing.  The only exchange is the framed-output gather: every rank sends its compacted records to the rank that
owns the io.Writer (rank 0), which interleaves them back into stream order.

Transport is torch.distributed ("nccl" == RCCL over xGMI on the GPU box; "gloo" in the CPU tests).  The byte mover on the
owner rank is pluggable: the product passes Engine.dev_scatter_records (HIP kernel), the CPU test a torch stand-in.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def local_block_ids(n_global: int, rank: int, world: int):
    """Round-robin ownership: global block g -> rank g % world (local index g // world)."""
    return list(range(rank, n_global, world))


def interleave_offsets(lens: torch.Tensor):
    """lens[r][j] = record length of local block j on rank r (global block j*G + r).  Returns (dst_off[r][j], total):
    the exclusive prefix sum of the lengths taken in GLOBAL block order."""
    world, b = lens.shape
    inter = lens.t().contiguous().view(-1).to(torch.int64)                 # [j*G + r]
    dst = torch.cumsum(inter, 0) - inter
    return dst.view(b, world).t().contiguous(), int(inter.sum().item())


PIECE_BYTES = 256 << 20


def piece_cuts(lens_r, piece: int):
    """Records [cut[t], cut[t+1]) of one rank travel together: as many whole records as fit `piece` bytes (at least one).
    Every rank computes the same cuts from the gathered lengths."""
    cuts, acc = [0], 0
    for j, n in enumerate(lens_r):
        if acc and acc + n > piece:
            cuts.append(j); acc = 0
        acc += n
    cuts.append(len(lens_r))
    return cuts


def gather_frame_body(body: torch.Tensor, rec_len: torch.Tensor, rank: int, world: int, scatter, scratch: dict,
                      max_rec: int, piece: int = PIECE_BYTES):
    """body: this rank's compacted records (uint8, device); rec_len: int32[B] record lengths (same B on every rank).
    scatter(src, src_off, lens, dst_off, n, max_len, dst) moves records on the owner's device.
    Returns (frame_body, total_bytes) on rank 0, (None, total_bytes_of_this_rank) elsewhere.

    The bodies travel in pieces of whole records (<= `piece` bytes): in round t every other rank sends its t-th piece, rank 0
    receives them into one of two scratch pieces per sender and moves the records of round t to their place while round t+1
    is on its way.  Rank 0 holds the frame body (the output) plus 2 x (world - 1) pieces, whatever the batch size."""
    dev = body.device
    b = rec_len.numel()
    all_lens = [torch.empty_like(rec_len) for _ in range(world)]
    dist.all_gather(all_lens, rec_len)                                     # tiny: G x B int32
    lens = torch.stack(all_lens)                                           # [G][B]
    lens_h = lens.cpu().tolist()
    totals = [sum(x) for x in lens_h]
    cuts = [piece_cuts(lens_h[r], piece) for r in range(world)]
    offs = []
    for r in range(world):
        o, acc = [0], 0
        for n in lens_h[r]:
            acc += n; o.append(acc)
        offs.append(o)
    rounds = max(len(c) - 1 for c in cuts[1:]) if world > 1 else 0
    if rank != 0:
        c = cuts[rank]
        for t in range(len(c) - 1):
            # the same group form as the receiver's: torch runs grouped and single send / recv on different communicators
            # (device-keyed vs pair-keyed), so both ends of a transfer have to take the same one
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, body[offs[rank][c[t]]:offs[rank][c[t + 1]]], 0)]):
                q.wait()
        return None, totals[rank]
    need = max([offs[r][cuts[r][t + 1]] - offs[r][cuts[r][t]] for r in range(1, world) for t in range(len(cuts[r]) - 1)] + [1])
    if scratch.get("cap", 0) < need or scratch.get("world") != world:
        scratch["recv"] = torch.empty(2 * max(world - 1, 1) * need, dtype=torch.uint8, device=dev)
        scratch["cap"] = need; scratch["world"] = world
    if scratch.get("frame_cap", 0) < sum(totals):
        # (sized from the gathered lengths, which are exact: no slack beyond a margin for the mover's 16-byte stores)
        scratch["frame"] = torch.empty(sum(totals) + (1 << 16), dtype=torch.uint8, device=dev)
        scratch["frame_cap"] = scratch["frame"].numel()
    cap = scratch["cap"]

    def slot(r, t):
        base = ((r - 1) * 2 + (t & 1)) * cap
        return scratch["recv"][base:base + cap]

    def post(t):
        # one group call for the round (ncclGroupStart / End under RCCL): the pieces of all senders arrive at once, each over
        # its own xGMI link, instead of one receive after the other on the communicator's stream
        ops = []
        for r in range(1, world):
            c = cuts[r]
            if t < len(c) - 1:
                nbytes = offs[r][c[t + 1]] - offs[r][c[t]]
                ops.append(dist.P2POp(dist.irecv, slot(r, t)[:nbytes], r))
        return dist.batch_isend_irecv(ops) if ops else []

    dst_off, total = interleave_offsets(lens)
    reqs = post(0) if rounds else []
    src0 = torch.cumsum(lens[0].to(torch.int64), 0) - lens[0]              # rank 0's own records, while round 0 travels
    scatter(body, src0, lens[0].contiguous(), dst_off[0], b, max_rec, scratch["frame"])
    for t in range(rounds):
        for q in reqs:
            q.wait()
        reqs = post(t + 1) if t + 1 < rounds else []                      # into the other scratch piece of every sender
        for r in range(1, world):
            c = cuts[r]
            if t < len(c) - 1:
                j0, j1 = c[t], c[t + 1]
                lr = lens[r][j0:j1].contiguous()
                src_r = torch.cumsum(lr.to(torch.int64), 0) - lr
                scatter(slot(r, t), src_r, lr, dst_off[r][j0:j1].contiguous(), j1 - j0, max_rec, scratch["frame"])
    return scratch["frame"], total
