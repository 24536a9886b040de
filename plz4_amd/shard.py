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


def gather_frame_body(body: torch.Tensor, rec_len: torch.Tensor, rank: int, world: int, scatter, scratch: dict,
                      max_rec: int):
    """body: this rank's compacted records (uint8, device); rec_len: int32[B] record lengths (same B on every rank).
    scatter(src, src_off, lens, dst_off, n, max_len, dst) moves records on the owner's device.
    Returns (frame_body, total_bytes) on rank 0, (None, total_bytes_of_this_rank) elsewhere."""
    dev = body.device
    b = rec_len.numel()
    all_lens = [torch.empty_like(rec_len) for _ in range(world)]
    dist.all_gather(all_lens, rec_len)                                     # tiny: G x B int32
    lens = torch.stack(all_lens)                                           # [G][B]
    totals = lens.sum(dim=1, dtype=torch.int64).cpu().tolist()
    if rank != 0:
        dist.send(body[:totals[rank]], dst=0)
        return None, totals[rank]
    cap = max(totals)
    if scratch.get("cap", 0) < cap or scratch.get("world") != world:
        scratch["recv"] = torch.empty(max(world - 1, 1) * cap, dtype=torch.uint8, device=dev)
        scratch["cap"] = cap; scratch["world"] = world
    if scratch.get("frame_cap", 0) < sum(totals):
        scratch["frame"] = torch.empty(int(sum(totals) * 1.02) + (1 << 16), dtype=torch.uint8, device=dev)
        scratch["frame_cap"] = scratch["frame"].numel()
    cap = scratch["cap"]
    reqs = [dist.irecv(scratch["recv"][(r - 1) * cap:(r - 1) * cap + totals[r]], src=r) for r in range(1, world)]
    dst_off, total = interleave_offsets(lens)                              # overlaps with the transfers
    src0 = torch.cumsum(lens[0].to(torch.int64), 0) - lens[0]
    scatter(body, src0, lens[0].contiguous(), dst_off[0], b, max_rec, scratch["frame"])
    for q in reqs:
        q.wait()
    for r in range(1, world):
        src_r = torch.cumsum(lens[r].to(torch.int64), 0) - lens[r] + (r - 1) * cap
        scatter(scratch["recv"], src_r, lens[r].contiguous(), dst_off[r], b, max_rec, scratch["frame"])
    return scratch["frame"], total
