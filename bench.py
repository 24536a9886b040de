#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: MiB/s of plaintext through encode + decode of 4 MiB independent
LZ4 blocks (level 1, block checksum on, content checksum off: configs[1]/[2]), with the % of the HBM roofline of
the dominant kernel and plz4's CPU path timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--kind T|R|Z|M]
    N > 1 is launched by torch.distributed.run, one rank per GPU (RCCL), see the contract in the task statement.

One step = one pass of the hot path, both directions, over one batch that is already resident in HBM:
    ONE duplex call (plz4hip_dev_duplex_records): the level-1 encode of this step's B blocks (-> staged records + xxh32)
    and, in the same launch, the decode of the frame body the previous step produced (-> B plaintext blocks, checksums
    verified) -- a writer's next batch beside a reader's; the records land back to back in the frame body (plz4hip_dev_duplex_body:
    no staging area, no compaction pass since round 4)  [N > 1: RCCL gather of the bodies to rank 0 + interleave into the final
    frame body, or --gather none].
Every step is one full encode and one full decode of B blocks; the body the last step produced is decoded and compared
after the timed region.  `--duplex 0` (and every HC level) runs the step as encode -> frame body -> decode of the same
batch, one call after the other; at N = 1 the default run times those K steps as well and reports them as `serial_step`.
Blocks are independent, so ranks share nothing on the data path (weak scaling: B blocks per GPU); block i of the
global stream lives on rank i mod N.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BSZ = 4 << 20
POOL_BLOCKS = 16            # 64 MiB of unique T text generated on the host, tiled (rotated) to B blocks on the device
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def committed_traffic(kernel: str, blocks: int, level: int = 1):
    """Bytes per STEP that crossed the L2 boundary for the kernels named (joined by +), from the committed rocprofv3 PMC passes
    (profiles/*_summary.json, scripts/gpu_profile_r0*.sh + scripts/summarize_profile.py; FETCH_SIZE doubled as MI355X_MICROARCH.md
    §HBM prescribes, plus WRITE_SIZE).  The counters sit behind L2: hits in the Infinity Cache are in them, so this is an upper
    bound of the HBM traffic.  A summary counts only if it was taken at this workload size AND this level, and only if it holds
    per-step totals (an HC call launches its kernels once per group of blocks, so a per-launch average times one is not a step) --
    or, for summaries older than that field, if every kernel asked for ran exactly once per step there (level 1).  None otherwise:
    PMC counters cannot be read from inside the timed process, and another level's figure is worse than none."""
    import glob, re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
            lv = d.get("level")
            if lv is None:                                                   # rounds 1-3: the level is in the tag
                m = re.search(r"level(\d+)", d.get("tag", ""))
                lv = int(m.group(1)) if m else 1
            if d.get("blocks_per_gpu") != blocks or lv != level:
                continue
            ks = kernel.split("+")
            ent = [d["kernels"].get(k, {}) for k in ks]
            if level == 1:
                # a level-1 step launches each of its kernels once (one group at the bench's sizes), and the profiled run mixes
                # duplex and serial steps: the per-launch average of a kernel is its step's figure
                key = lambda e: e.get("l2_miss_traffic_bytes", e.get("hbm_traffic_bytes_fetch_x2"))  # (second name: rounds 1-2)
                if all(key(e) is not None for e in ent):
                    best = {"bytes": sum(key(e) for e in ent), "source": os.path.basename(f), "per": "launch (one launch per step)"}
            elif all("l2_miss_traffic_bytes_per_step" in e for e in ent):
                best = {"bytes": sum(e["l2_miss_traffic_bytes_per_step"] for e in ent), "source": os.path.basename(f), "per": "step"}
        except Exception:
            pass
    return best


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(pool: np.ndarray, seconds: float = 12.0, level: int = 1, check=()):
    """plz4's CPU path for the same per-block work (LZ4_compress_fast cap=bsz -> stored fallback -> xxh32, then
    verify xxh32 + LZ4_decompress_safe into bsz+8), one block per task on all host cores.  Uses the compiled
    reference liblz4 (oracle/_ref) when present, else the oracle restatement.  This leg is the only place where bench.py
    touches oracle/: besides timing it, it checks the (block, record) pairs the GPU produced in the parity gate against it."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    orc = orclib.Oracle()
    kind = "port"
    enc = lambda s, d: orc.L.orc_compress_fast(s, BSZ, d, BSZ)
    dec = lambda s, n, d: orc.L.orc_decompress_safe(s, n, d, BSZ + 8)
    if os.path.exists(orclib.REF_SO):
        ref = orclib.Ref()
        kind = "reference"
        enc = (lambda s, d: ref.L.LZ4_compress_fast(s, d, BSZ, BSZ, 1)) if level == 1 else \
              (lambda s, d: ref.L.LZ4_compress_HC(s, d, BSZ, BSZ, level))
        dec = lambda s, n, d: ref.L.LZ4_decompress_safe(s, d, n, BSZ + 8)
    for i, (blk, got) in enumerate(check):                                 # the GPU's records == the reference's, byte for byte
        tmp = np.empty(BSZ + 8, dtype=np.uint8)
        c = enc(blk.ctypes.data_as(orclib.u8p), tmp.ctypes.data_as(orclib.u8p))
        payload, word = (blk, 0x80000000 | blk.size) if c == 0 else (tmp[:c], c)
        want = np.concatenate([np.frombuffer(np.uint32(word).tobytes(), dtype=np.uint8), payload,
                               np.frombuffer(np.uint32(orc.xxh32(np.ascontiguousarray(payload))).tobytes(), dtype=np.uint8)])
        assert np.array_equal(want, got), "record %d differs from the %s encoder" % (i, kind)
    cores = os.cpu_count() or 1
    npool = pool.size // BSZ
    nblk = max(npool, 2 * cores) if level == 1 else max(npool, cores)        # at least one (HC) / two tasks per hardware thread
    srcs = [np.ascontiguousarray(pool[(i % npool) * BSZ:(i % npool + 1) * BSZ]) for i in range(npool)]
    comp = [np.empty(BSZ + 8, dtype=np.uint8) for _ in range(nblk)]
    outs = [np.empty(BSZ + 8, dtype=np.uint8) for _ in range(nblk)]
    clen = [0] * nblk
    u8p = orclib.u8p

    def do_enc(i):
        s = srcs[i % npool].ctypes.data_as(u8p); d = comp[i].ctypes.data_as(u8p)
        c = enc(s, d)
        if c == 0:
            comp[i][:BSZ] = srcs[i % npool]; c = BSZ
        clen[i] = c
        return orc.L.orc_xxh32(d, c)

    def do_dec(i):
        d = comp[i].ctypes.data_as(u8p)
        orc.L.orc_xxh32(d, clen[i])
        return dec(d, clen[i], outs[i].ctypes.data_as(u8p))

    def timed(fn, items, threads):
        with ThreadPoolExecutor(threads) as ex:
            t0 = time.perf_counter(); list(ex.map(fn, items)); return time.perf_counter() - t0

    timed(do_enc, range(nblk), cores)                                        # warm-up pass
    t_enc = t_dec = 0.0; passes = 0
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < seconds:
        t_enc += timed(do_enc, range(nblk), cores)
        t_dec += timed(do_dec, range(nblk), cores)
        passes += 1
    mib = passes * nblk * BSZ / 2**20
    t1e = timed(do_enc, range(4), 1); t1d = timed(do_dec, range(4), 1)
    one = 4 * BSZ / 2**20
    return {
        "value": round(mib / (t_enc + t_dec), 1), "unit": "MiB/s", "cores": cores, "kind": kind,
        "sample": "%d enc+dec passes over %d x 4MiB T blocks (%.0f MiB in all), one block per task on %d threads; "
                  "enc %.0f MiB/s, dec %.0f MiB/s; 1 thread: enc %.0f, dec %.0f MiB/s; LZ4 = %s, xxh32 = oracle C"
                  % (passes, nblk, mib, cores, mib / t_enc, mib / t_dec, one / t1e, one / t1d,
                     "vendored liblz4 1.10.0 (oracle/_ref)" if kind == "reference" else "oracle restatement"),
    }


def crossover(args):
    """ABI B (plz4hip_encode_records / plz4hip_decode_records) from HOST memory -- what a plz4 built on this engine pays per
    call, PCIe and staging copies included -- for 1, 16, 256 and 2560 blocks of 4 MiB in one call, warm, caller buffers ready;
    beside it the reference CPU path (oracle/_ref, every host core) on the same number of blocks.  A labelled table, never
    `value` of the headline line."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from plz4_amd import synth
    from plz4_amd._native import Engine, _ptr_array, _i32, _i32p
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    level = args.level
    eng = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    pool = synth.make(args.kind, POOL_BLOCKS * BSZ, BSZ)
    ref = orclib.Ref() if os.path.exists(orclib.REF_SO) else None
    orc = orclib.Oracle()
    cores = os.cpu_count() or 1
    u8p = orclib.u8p
    points = []
    for nblk in (1, 16, 256, 2560):
        if level > 1 and nblk > 256:
            continue
        srcs = [np.ascontiguousarray(np.roll(pool[(i % POOL_BLOCKS) * BSZ:(i % POOL_BLOCKS + 1) * BSZ], -(i // POOL_BLOCKS) * 1009)) for i in range(nblk)]
        recs = [np.empty(BSZ + 8, dtype=np.uint8) for _ in range(nblk)]
        outs = [np.empty(BSZ + 8, dtype=np.uint8) for _ in range(nblk)]
        lens = _i32([s.size for s in srcs]); rl = np.zeros(nblk, dtype=np.int32)
        res = np.zeros(nblk, dtype=np.int32); st = np.zeros(nblk, dtype=np.int32)
        sp, rp, op = _ptr_array(srcs), _ptr_array(recs), _ptr_array(outs)
        reps = 3 if nblk <= 256 else 2
        te = td = 1e9
        for rep in range(reps + 1):                                          # first pass warms staging
            t0 = time.perf_counter()
            eng._chk(eng.L.plz4hip_encode_records(eng.h, nblk, sp, _i32p(lens), BSZ, level, 1, rp, _i32p(rl)))
            t1 = time.perf_counter()
            eng._chk(eng.L.plz4hip_decode_records(eng.h, nblk, rp, _i32p(rl), BSZ, 1, op, _i32p(res), _i32p(st)))
            t2 = time.perf_counter()
            if rep:
                te = min(te, t1 - t0); td = min(td, t2 - t1)
        assert int(np.abs(st).sum()) == 0 and all(np.array_equal(o[:BSZ], s) for o, s in zip(outs[:2], srcs[:2]))
        mib = nblk * BSZ / 2**20
        pt = {"blocks_in_flight": nblk, "gpu_enc_MiBps": round(mib / te, 1), "gpu_dec_MiBps": round(mib / td, 1),
              "gpu_encdec_MiBps": round(mib / (te + td), 1), "gpu_call_ms": {"encode": round(te * 1e3, 2), "decode": round(td * 1e3, 2)}}
        if ref is not None and not args.no_cpu_baseline:
            comp = [np.empty(BSZ + 8, dtype=np.uint8) for _ in range(nblk)]; clen = [0] * nblk

            def do_enc(i):
                c = ref.L.LZ4_compress_fast(srcs[i].ctypes.data_as(u8p), comp[i].ctypes.data_as(u8p), BSZ, BSZ, 1) if level == 1 else \
                    ref.L.LZ4_compress_HC(srcs[i].ctypes.data_as(u8p), comp[i].ctypes.data_as(u8p), BSZ, BSZ, level)
                clen[i] = c
                orc.L.orc_xxh32(comp[i].ctypes.data_as(u8p), c)

            def do_dec(i):
                orc.L.orc_xxh32(comp[i].ctypes.data_as(u8p), clen[i])
                ref.L.LZ4_decompress_safe(comp[i].ctypes.data_as(u8p), outs[i].ctypes.data_as(u8p), clen[i], BSZ + 8)

            th = min(cores, nblk)
            with ThreadPoolExecutor(th) as ex:
                list(ex.map(do_enc, range(nblk)))
                t0 = time.perf_counter(); list(ex.map(do_enc, range(nblk))); t1 = time.perf_counter()
                list(ex.map(do_dec, range(nblk))); t2 = time.perf_counter()
            pt.update({"cpu_enc_MiBps": round(mib / (t1 - t0), 1), "cpu_dec_MiBps": round(mib / (t2 - t1), 1),
                       "cpu_encdec_MiBps": round(mib / (t2 - t0), 1), "cpu_threads": th})
        points.append(pt)
        log("crossover:", pt)
    eng.close()
    return {"mode": "crossover (NOT the headline metric): host-buffer ABI B, PCIe included, vs blocks in flight", "level": level,
            "unit": "MiB/s of plaintext", "block_bytes": BSZ, "data": "synthetic %s" % args.kind, "points": points,
            "cpu": "reference liblz4 1.10.0 (oracle/_ref) + oracle xxh32, one block per task" if ref is not None else None}


def same_bytes(a, b, chunk: int = 1 << 30) -> bool:
    """torch.equal in pieces: the comparison of two 24 GiB buffers would otherwise allocate a third one for its mask"""
    import torch
    n = a.numel()
    if n != b.numel():
        return False
    for lo in range(0, n, chunk):
        if not torch.equal(a[lo:lo + chunk], b[lo:lo + chunk]):
            return False
    return True


def memory_plan(B: int, world: int, gather_rank0: bool, duplex: bool, ratio: float, level: int = 1, pipelines: int = 1):
    """GiB this bench holds on the busiest rank (rank 0) for B blocks per GPU: its own buffers, the library's level-1 workspace
    (9 bytes per possible sequence -- a sequence takes at least 4 input bytes -- plus the chunk tables: launch_l1 in plz4hip.hip) and,
    when the framed output is gathered, the assembled frame body and the receive pieces (plz4_amd/shard.py)."""
    S = B * BSZ
    C = ratio * S
    g = 2.0**30
    P = max(1, pipelines)                     # (every pipeline: its own plaintext out, two frame bodies, one record workspace)
    plan = {
        "src": S / g, "out": P * S / g, "stage": (B * (BSZ + 16) / g) if level > 2 else 0.0,
        "frame_body": P * min(B * (BSZ + 8), C * 1.02 + (1 << 20)) / g * (2 if duplex else 1),
        "l1_workspace": P * (B * ((BSZ // 4 + 3 + 63) // 64 * 64) * 9 + B * 1024 * 8) / g if level <= 2 else 0.0,
    }
    if gather_rank0 and world > 1:
        plan["gathered_frame"] = world * C / g
        plan["recv_pieces"] = 2 * (world - 1) * (256 << 20) / g
    plan["total"] = sum(plan.values())
    return {k: round(v, 1) for k, v in plan.items()}


def main():
    # Exactly one line goes to stdout: the JSON.  Libraries print to the C-level stdout as well (RCCL writes a version banner
    # when a communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the JSON is written to
    # the original stdout at the end.
    # The HIP runtime spreads a process's streams over four hardware queues unless told otherwise; this process has the NULL stream
    # (torch), the library's own, one per pipeline and -- with N > 1 -- the gather's and RCCL's.  Streams that share a queue
    # run one behind the other (found the hard way twice: DESIGN 6.4); sixteen queues leave every stream its own (eight still paired the two pipeline streams of an N > 1 rank).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("PLZ4_BENCH_BLOCKS", "6144")),
                    help="4 MiB blocks per GPU per step")
    ap.add_argument("--kind", default="T")
    ap.add_argument("--level", type=int, default=1, help="1 (configs[1]/[2], the headline) or 2..12 (LZ4_compress_HC; configs[3] is level 12)")
    ap.add_argument("--pipe", type=int, default=int(os.environ.get("PLZ4_BENCH_PIPE", "1")),
                    help="parts per step; decode of part p overlaps encode of part p+1 on a second stream (1 = serial)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--duplex", type=int, default=int(os.environ.get("PLZ4_BENCH_DUPLEX", "1")),
                    help="level 1 only.  1 (default): a step is ONE duplex call (plz4hip_dev_duplex_records): the level-1 encode of this step's batch beside the "
                         "decode of the frame body the previous step produced (the decoder's waves share every CU with the parser's), "
                         "then scan + compact; 0: encode -> frame body -> decode of the same batch, one after the other")
    ap.add_argument("--pipelines", type=int, default=int(os.environ.get("PLZ4_BENCH_PIPELINES", "0")),
                    help="level-1 duplex steps on one GPU: the steps alternate over this many HIP streams, each with its own frame bodies, "
                         "plaintext buffer and (inside the library) record workspace; a step's call decodes the body the same stream's "
                         "previous call wrote.  The emit kernels of one step then run beside the parse of the next.  0 (default): 2 on "
                         "one GPU, 1 with N > 1 (rank 0 has the assembled frame to hold)")
    ap.add_argument("--gather", choices=("rank0", "none"), default=os.environ.get("PLZ4_BENCH_GATHER", "rank0"),
                    help="N > 1: rank0 (default) = the framed output is gathered to rank 0 over RCCL and interleaved into ONE frame body "
                         "there (a single io.Writer); none = every rank keeps its frame body, block i on rank i mod N (SURVEY 8d config 3: "
                         "'or left sharded -- report both'): no exchange at all")
    ap.add_argument("--decode-only", action="store_true",
                    help="configs[2]: the step is the decode of the (already framed, resident) records alone; blocks stay sharded "
                         "block i -> rank i mod N, the plaintext stays on the rank that decoded it")
    ap.add_argument("--crossover", action="store_true",
                    help="second mode, never the headline: enc+dec MiB/s of the HOST-buffer ABI (PCIe included) against blocks in "
                         "flight, next to the CPU path on the same blocks -- where the drop-in engine starts to pay")
    args = ap.parse_args()
    if args.crossover:
        line = crossover(args)
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        return

    if args.level > 1:
        # the HC levels live on blocks in flight, i.e. on workspace; the library takes a quarter of free memory by default, a
        # machine that is there for this job says so
        os.environ.setdefault("PLZ4HIP_HC_BUDGET_GIB", "192")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("note: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    from plz4_amd import synth
    from plz4_amd._native import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # PLZ4_BENCH_FORCE_GATHER=1 runs the N > 1 code path (process group, gather stream, interleave) with a single rank: a
    # dry run of everything but the peer transfers, for boxes with one GPU
    multi = world > 1 or bool(os.environ.get("PLZ4_BENCH_FORCE_GATHER"))
    gather_rank0 = multi and args.gather == "rank0" and not args.decode_only
    if multi and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        dist.init_process_group("nccl", device_id=dev)
    eng = Engine(local)

    B = args.blocks
    S = B * BSZ
    t0 = time.time()
    NPIPE = args.pipelines if args.pipelines > 0 else 2
    if not (bool(args.duplex) and args.level == 1 and not args.decode_only and args.pipe <= 1):
        NPIPE = 1
    # N > 1: rank 0 has the assembled frame to hold and no room for a second full-size workspace, so there the two streams take one
    # HALF of the step's blocks each (a step = one call per stream; buffers and workspaces are the one-stream step's, cut in two):
    # 3072 + 3072 blocks over two streams run at 78 300 MiB/s on one GPU where 6144 on one stream run at 73 300.
    SPLIT = multi and NPIPE > 1
    if SPLIT:
        NPIPE = 2
    if rank == 0:
        free0, total0 = torch.cuda.mem_get_info(dev)
        dup0 = bool(args.duplex) and args.level == 1 and not args.decode_only
        plan = memory_plan(B, world, gather_rank0, dup0, 0.40, args.level, 1 if SPLIT else NPIPE)
        log("rank 0 memory plan, GiB (%d rank(s), gather %s): %s; device %.1f GiB, free %.1f GiB" %
            (world, args.gather if multi else "-", plan, total0 / 2**30, free0 / 2**30))
        if world == 1 and multi:                                               # the dry run also says what rank 0 of a full node would hold
            log("rank 0 memory plan, GiB, for 8 ranks with the gather: %s" % memory_plan(B, 8, True, dup0, 0.40, args.level, 1))
        # a fifth of the card stays free for the allocator and for whatever else lives there: a plan that does not leave it is
        # refused here rather than found out in the middle of a timed step (the level-1 workspace falling back to groups)
        # (the memory of a process that has just exited comes back asynchronously: a bench started right behind another GPU job
        # waits for what it plans to hold instead of finding a workspace refused in the middle of its steps)
        t_w = time.time()
        while free0 / 2**30 < plan["total"] + 8 and time.time() - t_w < 60:
            time.sleep(1.0); free0, total0 = torch.cuda.mem_get_info(dev)
        if time.time() - t_w > 1:
            log("rank 0: waited %.0f s for device memory, free now %.1f GiB" % (time.time() - t_w, free0 / 2**30))
        assert plan["total"] <= 0.8 * total0 / 2**30 or os.environ.get("PLZ4_BENCH_NO_HEADROOM_CHECK"), \
            "rank 0 would hold %.1f GiB of %.1f: less than 20 %% headroom (fewer --blocks, or --gather none)" % (plan["total"], total0 / 2**30)
    pool = synth.make(args.kind, POOL_BLOCKS * BSZ, BSZ)
    d_pool = torch.from_numpy(pool).to(dev)
    d_src = torch.empty(S, dtype=torch.uint8, device=dev)
    psz = d_pool.numel()
    for r in range((S + psz - 1) // psz):                                    # rotated replicas: every block distinct
        g = rank + world * r                                                 # global replica id (block i -> rank i mod N)
        shift = (g * 1000003) % psz
        rep = torch.roll(d_pool, -shift) if shift else d_pool
        lo = r * psz; n = min(psz, S - lo)
        d_src[lo:lo + n] = rep[:n]
    del rep
    stride = eng.stage_stride(BSZ)
    d_out = torch.empty(S, dtype=torch.uint8, device=dev)
    # The batch is processed as `pipe` consecutive parts (like a writer emitting batch after batch): part p+1 is encoded
    # while part p is decoded on a second HIP stream -- the decoder needs no LDS and fills issue slots the LDS-bound
    # encoder leaves idle.  Every part has its own staging / sizes / offsets / body.
    NP = 2 if (SPLIT and B >= 2) else max(1, min(args.pipe, B))
    bounds = [(B * i) // NP for i in range(NP + 1)]
    parts = []
    for i in range(NP):
        b0, b1 = bounds[i], bounds[i + 1]; nb = b1 - b0
        parts.append({
            "b0": b0, "nb": nb, "bytes": nb * BSZ,
            "src": d_src[b0 * BSZ:b1 * BSZ], "out": d_out[b0 * BSZ:b1 * BSZ],
            # (levels 1 and 2 write their records straight into the frame body: plz4hip_dev_encode_body / _duplex_body)
            "stage": torch.empty(nb * stride, dtype=torch.uint8, device=dev) if args.level > 2 else None,
            "len": torch.zeros(nb, dtype=torch.int32, device=dev),
            "off": torch.zeros(nb + 1, dtype=torch.int64, device=dev),
            # the frame body is sized for the data at hand, not for the worst case (24 GiB per 6144 blocks): 0.55 of the plaintext
            # to begin with (T text stores 0.38), the worst case only if the parity gate's compaction reports that it did not fit
            "body": torch.empty(min(nb * (BSZ + 8), int(nb * BSZ * float(os.environ.get("PLZ4_BENCH_BODY_FRAC", "0.55"))) + (1 << 20)), dtype=torch.uint8, device=dev),
            "res": torch.zeros(nb, dtype=torch.int32, device=dev),
            "st": torch.zeros(nb, dtype=torch.int32, device=dev),
        })
    SPLIT = SPLIT and NP == 2
    duplex = bool(args.duplex) and args.level == 1 and (NP == 1 or SPLIT) and not args.decode_only
    if duplex:
        # two frame bodies: the decode side of a duplex call reads the one the previous step compacted while this step's goes to the other
        for pt in (parts if SPLIT else parts[:1]):
            pt["bodies"] = [pt["body"], None]                                # (the second one once the gate has told how large a body is)
            pt["offs"] = [pt["off"], torch.zeros_like(pt["off"])]
            pt["cur"] = 0
            pt["gat_ev"] = [None, None]                                      # N > 1: the exchange that last read each body
            pt["stream"] = None                                              # (the current stream, below)
    else:
        NPIPE = 1
    # the pipelines: one stream -> the one part; N > 1 -> the two halves; one GPU -> further full-size ones once the gate has sized a body
    pipes = (list(parts) if SPLIT else [parts[0]]) if duplex else []
    calls = {"n": 0}
    s_enc = torch.cuda.current_stream()
    s_dec = torch.cuda.Stream(device=dev) if NP > 1 else s_enc
    # N > 1: the framed-output gather (RCCL send/recv + the interleave on rank 0) runs on its own stream, next to the decode
    # of the same records -- the decoder reads the local body and does not wait for the exchange
    s_gat = torch.cuda.Stream(device=dev) if gather_rank0 else None
    log("rank %d: %d blocks (%.1f GiB) in %d part(s) ready in %.1fs" % (rank, B, S / 2**30, NP, time.time() - t0))

    gather = {}

    def scatter(src, src_off, lens, dst_off, n, max_len, dst):
        eng.dev_scatter_records(src.data_ptr(), src_off.data_ptr(), lens.data_ptr(), dst_off.data_ptr(), n, max_len,
                                dst.data_ptr(), dst.numel(), torch.cuda.current_stream().cuda_stream)
        gather.setdefault("live", []).append((src_off, lens, dst_off))      # keep operands alive until the step ends

    def frame_gather(pt, which=None):
        """N > 1: rank 0 owns the io.Writer.  All-gather the record sizes (tiny), send every body to rank 0 over xGMI
        (RCCL send/recv), interleave there: global block g = j*N + r is record j of rank r (plz4_amd/shard.py).
        which: the one of the part's two bodies to send (the duplex steps gather a step late); None: the current one."""
        from plz4_amd import shard
        body = pt["body"] if which is None else pt["bodies"][which]
        off = pt["off"] if which is None else pt["offs"][which]
        total_local = int(off[-1].item())
        lens = (off[1:] - off[:-1]).to(torch.int32) if which is not None else pt["len"]   # (pt["len"] is the LAST call's)
        shard.gather_frame_body(body[:max(total_local, 1)], lens, rank, world, scatter, gather, BSZ + 8)

    def step_decode(ev=None):
        """configs[2]: decode alone, over the frame bodies the parity gate's full step left in place"""
        for i, pt in enumerate(parts):
            e = ev[i] if ev else None
            if e:
                for k in (0, 1, 2, 3, 4): e[k].record(s_enc)
            eng.dev_decode_records(pt["body"].data_ptr(), pt["off"].data_ptr(), pt["nb"], BSZ, True, pt["out"].data_ptr(), BSZ, BSZ,
                                   pt["res"].data_ptr(), pt["st"].data_ptr(), s_enc.cuda_stream)
            if e: e[5].record(s_enc)

    def step_duplex(ev=None):
        """one duplex call: encode of this step's batch straight into a frame body + decode of the frame body the same pipeline's
        previous call wrote (one pipeline: the previous step's).  Calls alternate over the pipelines, each on its own stream."""
        gather["live"] = []
        done = [duplex_call(ev[h] if ev else None) for h in range(len(pipes) if SPLIT else 1)]
        # The exchanges of the step BEFORE run now, behind this step's calls: sizing an exchange reads the body's length on the
        # host, which waits for that call's emit kernels -- with this step's calls already enqueued the GPU has work meanwhile, and
        # the next parse starts beside the last emit.  (A body is not written again before the exchange that read it is over:
        # gat_ev in duplex_call.)  flush_gathers() runs what is left: before the timed region starts and before it ends.
        if gather_rank0:
            flush_gathers(); pending_gathers.extend(done)
        else:
            pending_gathers.extend(done); flush_gathers()                   # (no exchange: only the step's events to record)

    pending_gathers = []

    def flush_gathers():
        todo = list(pending_gathers); del pending_gathers[:]
        for pt, cur, packed, e in todo:
            if gather_rank0:
                s_gat.wait_event(packed)
                with torch.cuda.stream(s_gat):
                    frame_gather(pt, cur)
                    if e: e[3].record(s_gat)
                    pt["gat_ev"][cur] = torch.cuda.Event(); pt["gat_ev"][cur].record(s_gat)
            elif e:
                e[3].record(pt["stream"] if pt["stream"] is not None else s_enc)
            if e:
                st_ = pt["stream"] if pt["stream"] is not None else s_enc
                e[4].record(st_); e[5].record(st_)

    def duplex_call(e):
        pt = pipes[calls["n"] % len(pipes)]; calls["n"] += 1
        st = pt["stream"] if pt["stream"] is not None else s_enc
        prv = pt["cur"]; cur = 1 - prv
        if e: e[0].record(st)
        # (two bodies alternate: the one this call writes is the one the exchange of the step BEFORE LAST read -- that exchange has
        # to be over; the last step's, which reads the body this call decodes, runs on under this call)
        if s_gat is not None and pt["gat_ev"][cur] is not None:
            st.wait_event(pt["gat_ev"][cur])
        eng.dev_duplex_body(pt["src"].data_ptr(), pt["bytes"], BSZ, True, pt["bodies"][cur].data_ptr(), pt["bodies"][cur].numel(),
                            pt["offs"][cur].data_ptr(), pt["len"].data_ptr(),
                            pt["bodies"][prv].data_ptr(), pt["offs"][prv].data_ptr(), pt["nb"], BSZ, True,
                            pt["out"].data_ptr(), BSZ, BSZ, pt["res"].data_ptr(), pt["st"].data_ptr(), st.cuda_stream)
        if e: e[1].record(st)
        pt["cur"] = cur; pt["body"] = pt["bodies"][cur]; pt["off"] = pt["offs"][cur]
        packed = torch.cuda.Event(enable_timing=False) if e is None else e[2]
        packed.record(st)
        return pt, cur, packed, e

    def step(ev=None, serial=False):
        """ev: per part [enc0, enc1, cmp1, gat1, dec0, dec1] events."""
        if ev is not None and args.decode_only:
            return step_decode(ev)
        if ev is not None and duplex and not serial:
            return step_duplex(ev)
        gather["live"] = []
        for i, pt in enumerate(parts):
            e = ev[i] if ev else None
            if e: e[0].record(s_enc)
            if args.level <= 2:
                if s_gat is not None:
                    s_enc.wait_stream(s_gat)                   # the previous step's sends (and interleave) still read this body
                eng.dev_encode_body(pt["src"].data_ptr(), pt["bytes"], BSZ, True, pt["body"].data_ptr(), pt["body"].numel(),
                                    pt["off"].data_ptr(), pt["len"].data_ptr(), s_enc.cuda_stream, level=args.level)
                if e: e[1].record(s_enc)
            else:
                eng.dev_encode_records(pt["src"].data_ptr(), pt["bytes"], BSZ, True, pt["stage"].data_ptr(), pt["len"].data_ptr(), s_enc.cuda_stream,
                                       level=args.level)
                if e: e[1].record(s_enc)
                if s_gat is not None:
                    s_enc.wait_stream(s_gat)                   # the previous step's sends (and interleave) still read this body
                eng.dev_compact_records(pt["stage"].data_ptr(), stride, pt["len"].data_ptr(), pt["nb"], pt["off"].data_ptr(),
                                        pt["body"].data_ptr(), pt["body"].numel(), s_enc.cuda_stream)
            packed = torch.cuda.Event(enable_timing=False) if e is None else e[2]
            packed.record(s_enc)
            if gather_rank0:
                s_gat.wait_event(packed)
                with torch.cuda.stream(s_gat):                 # collectives and the interleave kernels take the current stream
                    frame_gather(pt)
                    if e: e[3].record(s_gat)
            elif e:
                e[3].record(s_enc)
            if s_dec is not s_enc:
                s_dec.wait_event(packed)
            if e: e[4].record(s_dec)
            eng.dev_decode_records(pt["body"].data_ptr(), pt["off"].data_ptr(), pt["nb"], BSZ, True, pt["out"].data_ptr(), BSZ, BSZ,
                                   pt["res"].data_ptr(), pt["st"].data_ptr(), s_dec.cuda_stream)
            if e: e[5].record(s_dec)
        if s_dec is not s_enc:
            done = torch.cuda.Event(); done.record(s_dec); s_enc.wait_event(done)
        # N > 1: the exchange of this step's records runs on under the next step's encode (it is done long before that encode
        # reaches its compact, which waits for it above); the barrier + synchronize that close the timed region wait for the last
        # one, so K steps are K encodes, K decodes and K assembled frames

    # ---- correctness gate before any timing: round trip bit-exact, every block status OK (records vs the reference: cpu_baseline leg)
    d_out.zero_()
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    for pt in parts:
        if int(pt["off"][-1].item()) > pt["body"].numel():                   # (compaction skips what does not fit and says so here)
            log("rank %d: frame body of %.1f GiB too small for this data, taking the worst case" % (rank, pt["body"].numel() / 2**30))
            pt["body"] = None
            pt["body"] = torch.empty(pt["nb"] * (BSZ + 8), dtype=torch.uint8, device=dev)
            if duplex: pt["bodies"][0] = pt["body"]
            d_out.zero_(); torch.cuda.synchronize(); step(); torch.cuda.synchronize()
            break
    assert sum(int(pt["st"].abs().sum().item()) for pt in parts) == 0, "decode status != OK"
    assert sum(int(pt["res"].to(torch.int64).sum().item()) for pt in parts) == S, "decoded size mismatch"
    assert same_bytes(d_out, d_src), "round trip mismatch"
    C_bytes = sum(int(pt["off"][-1].item()) for pt in parts)
    d_off, d_body = parts[0]["off"], parts[0]["body"]
    check_pairs = []
    if rank == 0:
        offs = d_off[:3].cpu().tolist()
        check_pairs = [(d_src[i * BSZ:(i + 1) * BSZ].cpu().numpy().copy(), d_body[offs[i]:offs[i + 1]].cpu().numpy().copy()) for i in range(2)]
        log("parity gate ok: round trip exact, every block status OK, stored/plain ratio %.4f "
            "(two records are compared with the reference encoder in the cpu_baseline leg)" % (C_bytes / S))

    for pt in parts:                                                         # the gate has told how large a body is: keep that + 2 %
        c_pt = int(pt["off"][-1].item())
        want = min(pt["body"].numel(), int(c_pt * 1.02) + (1 << 20))
        if want < pt["body"].numel():
            nb_ = torch.empty(want, dtype=torch.uint8, device=dev)
            nb_[:c_pt] = pt["body"][:c_pt]
            pt["body"] = nb_
            if duplex: pt["bodies"][0] = nb_
    d_body = parts[0]["body"]
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    if duplex:
        d_out.zero_()                                                        # what the timed steps decode is checked again below
        # the same input every step, so the same body: the second one is that size + 2 % (compact skips what would not fit and
        # the check after the timed region would notice), which leaves rank 0 of an 8-GPU run room for the assembled frame
        for pt in pipes:
            pt["bodies"][1] = torch.empty(pt["bodies"][0].numel(), dtype=torch.uint8, device=dev)
            if NPIPE > 1:
                pt["stream"] = torch.cuda.Stream(device=dev)                 # (a stream of its own like the others', not the default one)
        bcap = parts[0]["bodies"][1].numel()
        for i in range(1, NPIPE if not SPLIT else 1):
            # a further pipeline: its own stream, plaintext buffer, two bodies and offsets; primed with one encode of the batch, so
            # that its first duplex call has a body to decode (the library gives the second stream a record workspace of its own)
            q = {"b0": 0, "nb": B, "bytes": S, "src": d_src, "out": torch.zeros(S, dtype=torch.uint8, device=dev),
                 "bodies": [torch.empty(bcap, dtype=torch.uint8, device=dev) for _ in range(2)],
                 "offs": [torch.zeros(B + 1, dtype=torch.int64, device=dev) for _ in range(2)],
                 "len": torch.zeros(B, dtype=torch.int32, device=dev), "res": torch.zeros(B, dtype=torch.int32, device=dev),
                 "st": torch.zeros(B, dtype=torch.int32, device=dev), "cur": 0, "gat_ev": [None, None],
                 "stream": torch.cuda.Stream(device=dev)}
            q["body"] = q["bodies"][0]; q["off"] = q["offs"][0]
            eng.dev_encode_body(q["src"].data_ptr(), q["bytes"], BSZ, True, q["body"].data_ptr(), q["body"].numel(),
                                q["off"].data_ptr(), q["len"].data_ptr(), q["stream"].cuda_stream, level=1)
            pipes.append(q)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step_decode() if args.decode_only else (step_duplex() if duplex else step())
    pipe_check = None
    if duplex and len(pipes) > 1 and not SPLIT:
        # (untimed, whatever --warmup says: every further pipeline makes one call, which also puts its record workspace in place)
        torch.cuda.synchronize()
        while calls["n"] % len(pipes) != 0 or calls["n"] < len(pipes):
            step_duplex()
        torch.cuda.synchronize()

        # Still untimed: do the pipelines pay on this process's streams?  Calls over several streams overlap only if the runtime
        # has put the streams on different hardware queues and the launches stay staggered; two processes of about forty were seen
        # to run them at 495 ms per step instead of 305 (every step of the run, the serial steps after it at their usual speed).
        # So a few steps are clocked both ways before the timed region; a set of streams that does not pay is replaced once, and
        # if the new one does not pay either the timed steps run on one stream (config.pipelines says what ran).
        def clock(ps, n):
            keep = list(pipes); pipes[:] = ps
            torch.cuda.synchronize(); t_ = time.perf_counter()
            for _ in range(n): step_duplex()
            torch.cuda.synchronize(); dt_ = (time.perf_counter() - t_) / n
            pipes[:] = keep
            return dt_
        skip_check = bool(os.environ.get("PLZ4_BENCH_NO_PIPE_CHECK"))                # (tests with a handful of blocks: nothing to clock)
        one = clock(pipes[:1], 2) if not skip_check else 1.0
        both = clock(pipes, 2 * len(pipes)) if not skip_check else 0.0
        first_both = both
        tries = 0
        force_bad = int(os.environ.get("PLZ4_BENCH_TEST_PIPE_CHECK", "0"))       # (tests: pretend the first N checks fail)
        while (both > 0.97 * one or tries < force_bad) and tries < 2:
            tries += 1
            if tries == 1:
                for pt_ in pipes: pt_["stream"] = torch.cuda.Stream(device=dev)
                both = clock(pipes, 2 * len(pipes))
            else:
                log("rank %d: %d streams do not pay here (%.1f ms per step against %.1f on one): the timed steps run on one stream" % (rank, len(pipes), both * 1e3, one * 1e3))
                del pipes[1:]; NPIPE = 1; pt_ = None
                torch.cuda.synchronize(); torch.cuda.empty_cache()
        pipe_check = {"one_stream_ms": round(one * 1e3, 1), "pipelined_ms": round(both * 1e3, 1), "first_pipelined_ms": round(first_both * 1e3, 1), "streams_replaced": tries}
        log("rank %d: pipeline check, ms per step: one stream %.1f, %d streams %.1f%s" % (rank, one * 1e3, NPIPE, both * 1e3, " (streams replaced)" if tries else ""))
        calls["n"] = 0
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    evs = [[[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in parts] for _ in range(args.steps)]
    if duplex: flush_gathers()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    if duplex: flush_gathers()                                               # (the last step's frame: K steps are K assembled frames)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    free_t, total_t = torch.cuda.mem_get_info(dev)
    held_timed = round((total_t - free_t) / 2**30, 1)
    if duplex:
        # every timed step decoded the body its pipeline's previous call wrote (the first ones the warm-up's or the gate's): the
        # plaintext must be back, and so must the plaintext of each pipeline's last body, decoded here outside the timed region
        for pt in pipes:
            assert int(pt["st"].abs().sum().item()) == 0 and same_bytes(pt["out"], pt["src"]), "duplex: round trip mismatch"
            pt["out"].zero_()
            eng.dev_decode_records(pt["body"].data_ptr(), pt["off"].data_ptr(), pt["nb"], BSZ, True, pt["out"].data_ptr(), BSZ, BSZ,
                                   pt["res"].data_ptr(), pt["st"].data_ptr(), s_enc.cuda_stream)
            torch.cuda.synchronize()
            assert int(pt["st"].abs().sum().item()) == 0 and same_bytes(pt["out"], pt["src"]), "duplex: last body does not decode to the input"
        if len(pipes) > 1 and not SPLIT:
            # the further pipelines' buffers are not needed for the serial leg below (nor is the library's second workspace)
            del pipes[1:]
            pt = q = None
            torch.cuda.synchronize(); torch.cuda.empty_cache()
    serial_leg = None
    if duplex and not multi:
        # beside the headline: the same K steps as encode -> frame body -> decode of the same batch, one call after the other (the
        # step of rounds 1-2), which is also where the encode call and k_decode_rec are timed by themselves
        evs2 = [[[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in parts] for _ in range(args.steps)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(evs2[k], serial=True)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        seg2 = np.array([[sum(e[0].elapsed_time(e[1]) for e in st_), sum(e[1].elapsed_time(e[2]) for e in st_),
                          sum(e[2].elapsed_time(e[3]) for e in st_), sum(e[4].elapsed_time(e[5]) for e in st_)] for st_ in evs2])
        serial_leg = (el2, seg2.mean(axis=0).tolist())
    # ms per step, summed over the parts: encode kernel, scan+compact, gather, decode kernel (decode overlaps the next encode)
    seg = np.array([[sum(e[0].elapsed_time(e[1]) for e in st_), sum(e[1].elapsed_time(e[2]) for e in st_),
                     sum(e[2].elapsed_time(e[3]) for e in st_), sum(e[4].elapsed_time(e[5]) for e in st_)] for st_ in evs])
    log("rank %d: call begin-to-end on its stream, ms per step: %s" % (rank, " ".join("%.1f" % x for x in seg[:, 0])))
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        enc_ms, cmp_ms, gat_ms, dec_ms = seg.mean(axis=0).tolist()
        mib = S / 2**20
        # level 12 is one ABI call of its phases (hist, chain, search, then the parser in segments: seg, stitch, gather) and the emit
        # kernels it shares with level 1 (lz4hc12_device.inl): they are timed together
        # level 1 is one ABI call of five kernels (parse, sizes, scan, write, finish; lz4_seq_device.inl), timed together as well
        enc_kernel = "k_l1_parse+k_l1_sizes+k_l1_scan+k_l1_write+k_l1_finish" if args.level == 1 else (
                      "k_hc12_hist+k_hc12_chain+k_hc12_search+k_hc12_seg+k_hc12_stitch+k_hc_gather+k_l1_sizes+k_l1_scan+k_l1_write+k_l1_finish" if args.level >= 12 else
                      ("k_hc12_hist+k_hc12_chain+k_hc_lazy+k_hc_stitch+k_hc_gather+k_l1_sizes+k_l1_scan+k_l1_write+k_l1_finish" if 3 <= args.level <= 11 else
                       ("k_hc_mid+k_l1_sizes+k_l1_scan+k_l1_write+k_l1_finish" if args.level == 2 else "k_encode_rec_hc")))
        ach_enc = (S + C_bytes) / (enc_ms * 1e-3) / 1e9
        dec_ms = max(dec_ms, 1e-6)                                              # (duplex: no decode launch of its own)
        ach_dec = (S + C_bytes) / (dec_ms * 1e-3) / 1e9
        if args.decode_only:
            line = {
                "metric": "MiB/s decompress-only, 4MiB independent blocks, block-checksum verified",
                "value": round(world * mib / (ms_step * 1e-3), 1), "unit": "MiB/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "configs[2]: decompress-only stream of %d x 4MiB independent blocks per GPU (level-%d records of synthetic "
                                       "%s text, block checksum verified), block i -> rank i mod N, records and plaintext resident in HBM, "
                                       "the plaintext stays on the rank that decoded it" % (B, args.level, args.kind),
                           "blocks_per_gpu": B, "block_bytes": BSZ, "stored_ratio": round(C_bytes / S, 4),
                           "sharding": "block i -> rank i mod N" if world > 1 else "single GPU"},
                "ms": {"decode_kernel": round(dec_ms, 3)},
                "roofline": {"bound": "hbm", "kernel": "k_decode_rec", "achieved": round(ach_dec, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach_dec / HBM_PEAK_GBS, 5), "traffic": None},
            }
            t = committed_traffic("k_decode_rec", B, args.level)
            if t:
                line["roofline"]["traffic"] = t["bytes"]; line["roofline"]["traffic_source"] = t["source"]
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
            if multi:
                dist.barrier(); dist.destroy_process_group()
            eng.close()
            return
        out = {
            "metric": "MiB/s enc+dec, 4MiB independent blocks, level %d, block-checksum on%s" % (args.level, (" (duplex steps over %d streams)" % NPIPE if NPIPE > 1 else " (duplex step)") if duplex else ""),
            "value": round(world * mib / (ms_step * 1e-3), 1),
            "unit": "MiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[1]+[2]" if args.level == 1 else "configs[3] (HC)") + ": %d x 4MiB independent blocks per GPU of synthetic %s text "
                                   "(64 MiB PCG64/Zipf pool, rotated replicas), level %d, block checksum on, content "
                                   "checksum off; step = encode->frame body->decode, inputs resident in HBM"
                                   % (B, args.kind, args.level),
                       "blocks_per_gpu": B, "block_bytes": BSZ, "pipeline_parts": NP, "stored_ratio": round(C_bytes / S, 4),
                       "sharding": "block i -> rank i mod N" if world > 1 else "single GPU",
                       **({"hc_workspace_budget_gib": int(os.environ["PLZ4HIP_HC_BUDGET_GIB"])} if args.level > 1 else {})},
            "enc_MiBps_per_gpu": round(mib / (enc_ms * 1e-3), 1),
            "dec_MiBps_per_gpu": round(mib / (dec_ms * 1e-3), 1),
            "ms": {"encode_kernel": round(enc_ms, 3), "scan_compact": round(cmp_ms, 3),
                   "frame_gather": round(gat_ms, 3), "decode_kernel": round(dec_ms, 3)},
            "roofline": {"bound": "hbm", "kernel": enc_kernel, "achieved": round(ach_enc, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach_enc / HBM_PEAK_GBS, 5), "traffic": None},
            "roofline_decode": {"bound": "hbm", "kernel": "k_decode_rec", "achieved": round(ach_dec, 2),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_dec / HBM_PEAK_GBS, 5),
                                "traffic": None},
        }
        if duplex:
            # one call does both directions: its algorithmic bytes are the encode's (S read + C written) plus the decode's (C read + S
            # written), over the call's time (k_l1_duplex + the four emit kernels)
            enc_kernel = "k_l1_duplex+k_l1_sizes+k_l1_scan+k_l1_write+k_l1_finish"
            # (several pipelines: the calls of different streams overlap, so a call's own begin-to-end time on its stream covers
            # the other stream's kernels too -- the rate is then taken over the timed region: K calls' bytes over the K steps' time)
            ach = 2 * (S + C_bytes) / ((ms_step if NPIPE > 1 else enc_ms) * 1e-3) / 1e9
            out["config"]["workload"] = out["config"]["workload"].replace(
                "step = encode->frame body->decode",
                "step = ONE duplex call (encode of this step's batch beside the decode of the frame body the previous step produced; every "
                "step is one full encode and one full decode, the last body is decoded and checked after the timed region) -> frame body")
            out["config"]["duplex"] = True
            out["config"]["pipelines"] = NPIPE
            if SPLIT:
                out["config"]["calls_per_step"] = 2; out["config"]["blocks_per_call"] = [pt["nb"] for pt in parts]
            if pipe_check: out["config"]["pipeline_check"] = pipe_check
            if NPIPE > 1:
                out["config"]["workload"] = out["config"]["workload"].replace(
                    "the frame body the previous step produced", "the frame body the same stream's previous call produced (the steps alternate "
                    "over %d HIP streams, each with its own bodies, plaintext buffer and record workspace, so the emit kernels of one "
                    "step run beside the parse of the next)" % NPIPE)
            out["roofline"] = {"bound": "hbm", "kernel": enc_kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None}
            out["ms"] = {"duplex_call": round(enc_ms, 3), "scan_compact": round(cmp_ms, 3), "frame_gather": round(gat_ms, 3)}
            if NPIPE > 1:
                out["ms"]["duplex_call_note"] = "begin to end on the call's own stream, %d calls in flight: not a kernel time (serial_step has those)" % NPIPE
                out["roofline"]["over"] = "the timed region (overlapping launches): 2 (S + C) x steps / elapsed"
            del out["roofline_decode"], out["enc_MiBps_per_gpu"], out["dec_MiBps_per_gpu"]
            if serial_leg:
                el2, (e2, c2, g2, d2) = serial_leg
                ms2 = el2 / args.steps * 1e3
                a_e = (S + C_bytes) / (e2 * 1e-3) / 1e9; a_d = (S + C_bytes) / (d2 * 1e-3) / 1e9
                k_e = "k_l1_parse+k_l1_sizes+k_l1_scan+k_l1_write+k_l1_finish"
                out["serial_step"] = {
                    "what": "the same %d steps as encode -> frame body -> decode of the same batch, one call after the other" % args.steps,
                    "value": round(world * mib / (ms2 * 1e-3), 1), "unit": "MiB/s", "ms_per_step": round(ms2, 3),
                    "enc_MiBps_per_gpu": round(mib / (e2 * 1e-3), 1), "dec_MiBps_per_gpu": round(mib / (d2 * 1e-3), 1),
                    "ms": {"encode_kernel": round(e2, 3), "scan_compact": round(c2, 3), "decode_kernel": round(d2, 3)},
                    "roofline": {"bound": "hbm", "kernel": k_e, "achieved": round(a_e, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(a_e / HBM_PEAK_GBS, 5), "traffic": None},
                    "roofline_decode": {"bound": "hbm", "kernel": "k_decode_rec", "achieved": round(a_d, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(a_d / HBM_PEAK_GBS, 5), "traffic": None}}
                for key, kern in (("roofline", k_e), ("roofline_decode", "k_decode_rec")):
                    t = committed_traffic(kern, B, 1)
                    if t:
                        out["serial_step"][key]["traffic"] = t["bytes"]; out["serial_step"][key]["traffic_source"] = t["source"]
        for key, kern in (("roofline", enc_kernel), ("roofline_decode", "k_decode_rec")):
            t = committed_traffic(kern, B, args.level) if key in out else None
            if t:
                out[key]["traffic"] = t["bytes"]; out[key]["traffic_source"] = t["source"]; out[key]["traffic_per"] = t["per"]
        free1, total1 = torch.cuda.mem_get_info(dev)
        out["config"]["gather"] = (args.gather if multi else None)
        out["memory_gib"] = {"plan": plan, "held_in_timed_region": held_timed, "held_at_end": round((total1 - free1) / 2**30, 1), "device": round(total1 / 2**30, 1)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pool, level=args.level, check=check_pairs)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
