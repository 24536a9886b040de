"""Experiment: two duplex pipelines alternating on two streams (two contexts, so two record workspaces) -- the emit kernels of
step k then run beside the duplex launch of step k+1, and each pipeline decodes the body its own previous call wrote.
usage: exp_pingpong.py BLOCKS STEPS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plz4_amd import synth
from plz4_amd._native import Engine

BSZ = 4 << 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
PIPES = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2]
SHARED_OUT = torch.empty(B * BSZ, dtype=torch.uint8, device="cuda:0") if os.environ.get("EXP_SHARED_OUT") else None
dev = torch.device("cuda", 0)
pool = synth.make("T", 16 * BSZ, BSZ)
d_pool = torch.from_numpy(pool).to(dev)
S = B * BSZ
d_src = torch.empty(S, dtype=torch.uint8, device=dev)
psz = d_pool.numel()
for r in range((S + psz - 1) // psz):
    shift = (r * 1000003) % psz
    rep = torch.roll(d_pool, -shift) if shift else d_pool
    lo = r * psz; n = min(psz, S - lo); d_src[lo:lo + n] = rep[:n]
del rep


ONE = Engine(0) if os.environ.get("EXP_ONE_CTX") else None


def pipeline():
    eng = ONE if ONE is not None else Engine(0)
    cap = int(S * 0.45) + (1 << 20)
    return {"eng": eng, "stream": torch.cuda.Stream(device=dev), "out": SHARED_OUT if SHARED_OUT is not None else torch.empty(S, dtype=torch.uint8, device=dev),
            "bodies": [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)],
            "offs": [torch.zeros(B + 1, dtype=torch.int64, device=dev) for _ in range(2)],
            "len": torch.zeros(B, dtype=torch.int32, device=dev), "res": torch.zeros(B, dtype=torch.int32, device=dev),
            "st": torch.zeros(B, dtype=torch.int32, device=dev), "cur": 0, "primed": False}


def call(p):
    prv = p["cur"]; cur = 1 - prv
    s = p["stream"].cuda_stream
    if not p["primed"]:
        p["eng"].dev_encode_body(d_src.data_ptr(), S, BSZ, True, p["bodies"][cur].data_ptr(), p["bodies"][cur].numel(),
                                 p["offs"][cur].data_ptr(), p["len"].data_ptr(), s, level=1)
        p["primed"] = True
    else:
        p["eng"].dev_duplex_body(d_src.data_ptr(), S, BSZ, True, p["bodies"][cur].data_ptr(), p["bodies"][cur].numel(),
                                 p["offs"][cur].data_ptr(), p["len"].data_ptr(),
                                 p["bodies"][prv].data_ptr(), p["offs"][prv].data_ptr(), B, BSZ, True,
                                 p["out"].data_ptr(), BSZ, BSZ, p["res"].data_ptr(), p["st"].data_ptr(), s)
    p["cur"] = cur


for npipes in PIPES:
    pipes = [pipeline() for _ in range(npipes)]
    for w in range(2 * npipes):                       # prime + one duplex call each
        call(pipes[w % npipes])
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(K):
        call(pipes[k % npipes])
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    ok = all(bool((p["st"] == 0).all().item()) and bool((p["res"] == BSZ).all().item()) and torch.equal(p["out"], d_src) for p in pipes)
    free, total = torch.cuda.mem_get_info(dev)
    print("pipelines %d, %d blocks per step: %.1f ms per step = %.0f MiB/s enc+dec; round trip %s; %.0f GiB in use" % (npipes, B, dt * 1e3, (S / 2**20) / dt, "ok" if ok else "BAD", (total - free) / 2**30), flush=True)
    for p in pipes:
        if ONE is None: p["eng"].close()
    if ONE is not None: ONE.trim()
    del pipes
    torch.cuda.empty_cache()
