"""Experiment: how often do two duplex pipelines of one context fall out of their stagger?  Segments of [synchronise, N steps
alternating over two streams, synchronise], clocked one by one.   usage: exp_pipe_phase.py BLOCKS SEGMENTS STEPS_PER_SEGMENT"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plz4_amd import synth
from plz4_amd._native import Engine

BSZ = 4 << 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
SEG = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
pool = synth.make("T", 16 * BSZ, BSZ)
d_pool = torch.from_numpy(pool).to(dev)
S = B * BSZ
d_src = d_pool.repeat((B + 15) // 16)[:S].contiguous()
eng = Engine(0)
cap = int(S * 0.42) + (1 << 20)
pipes = []
for p in range(2):
    q = {"stream": torch.cuda.Stream(device=dev), "out": torch.empty(S, dtype=torch.uint8, device=dev),
         "bodies": [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)],
         "offs": [torch.zeros(B + 1, dtype=torch.int64, device=dev) for _ in range(2)],
         "len": torch.zeros(B, dtype=torch.int32, device=dev), "res": torch.zeros(B, dtype=torch.int32, device=dev),
         "st": torch.zeros(B, dtype=torch.int32, device=dev), "cur": 0}
    eng.dev_encode_body(d_src.data_ptr(), S, BSZ, True, q["bodies"][0].data_ptr(), cap, q["offs"][0].data_ptr(), q["len"].data_ptr(), q["stream"].cuda_stream, level=1)
    pipes.append(q)
torch.cuda.synchronize()


def call(p):
    prv = p["cur"]; cur = 1 - prv
    eng.dev_duplex_body(d_src.data_ptr(), S, BSZ, True, p["bodies"][cur].data_ptr(), cap, p["offs"][cur].data_ptr(), p["len"].data_ptr(),
                        p["bodies"][prv].data_ptr(), p["offs"][prv].data_ptr(), B, BSZ, True,
                        p["out"].data_ptr(), BSZ, BSZ, p["res"].data_ptr(), p["st"].data_ptr(), p["stream"].cuda_stream)
    p["cur"] = cur


n = 0
out = []
for seg in range(SEG):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evs = []
    for k in range(N):
        p = pipes[n % 2]; n += 1
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(p["stream"]); call(p); e1.record(p["stream"]); evs.append((e0, e1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    out.append(dt)
    print("segment %2d: %.1f ms per step; calls begin-to-end %s" % (seg, dt, " ".join("%.0f" % a.elapsed_time(b) for a, b in evs)), flush=True)
print("segments over 1.2 x the best: %d of %d" % (sum(1 for x in out if x > 1.2 * min(out)), len(out)))
