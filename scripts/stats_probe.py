"""Diagnostics: build the engine with -DPLZ4_STATS into scripts/_build/libplz4hip_stats.so, run an encode over B blocks
of T text and print the per-phase counters.  Not part of the product or of the tests."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plz4_amd import synth, _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
so = os.path.join(ROOT, "scripts", "_build", "libplz4hip_stats.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared", "-DPLZ4_STATS",
                       "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-o", so,
                       os.path.join(ROOT, "plz4_amd", "csrc", "plz4hip.hip")])
_native.LIB_PATH = so
eng = _native.Engine(0)
L = eng.L
names = ["grid", "generic", "misorder", "seq_grid", "seq_gen", "twinstop", "sat", "longback", "memlit", "walkiter",
         "cyc_total", "cyc_load", "cyc_walk", "cyc_fix", "cyc_gen", "cyc_sat", "cyc_memlit", "blocks", "lanes_exec",
         "dbatch", "dmemb", "dseq", "dseq_ml15", "dseq_ll15"]
bsz = 4 << 20
pool = synth.text(16 * bsz)
dev = torch.device("cuda:0")
d_pool = torch.from_numpy(pool).to(dev)
d_src = d_pool.repeat((B + 15) // 16)[:B * bsz].contiguous()
stride = eng.stage_stride(bsz)
d_stage = torch.empty(B * stride, dtype=torch.uint8, device=dev)
d_len = torch.zeros(B, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 24)()
for chk in (True, False):
    L.plz4hip_debug_stats(out)
    torch.cuda.synchronize(); t0 = time.time()
    eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, chk, d_stage.data_ptr(), d_len.data_ptr(), 0)
    torch.cuda.synchronize(); dt = time.time() - t0
    L.plz4hip_debug_stats(out)
    v = dict(zip(names, list(out)))
    nb = max(v["blocks"], 1)
    print("checksum=%s  B=%d  %.1f ms  -> %.1f MiB/s" % (chk, B, dt * 1e3, B * 4 / dt))
    for k in names:
        print("   %-10s %14d   per block %12.1f" % (k, v[k], v[k] / nb))
    g = max(v["grid"], 1)
    print("   per grid batch: phase1 %.0f cyc, hop %.0f cyc, emit %.0f cyc, fix %.0f cyc, seqs %.2f, lanes exec %.1f, walk iters %.2f"
          % (v["cyc_load"] / g, v["cyc_walk"] / g, v["cyc_sat"] / g, v["cyc_fix"] / g, v["seq_grid"] / g, v["lanes_exec"] / g, v["walkiter"] / g))
    print("   hop split per grid batch: nextHit %.0f, walk %.0f, derive %.0f, repair %.0f cyc"
          % (v["dbatch"] / g, v["dmemb"] / g, v["dseq"] / g, v["dseq_ml15"] / g))
    print("   per generic batch: %.0f cyc;  total cyc/block %.3e" % (v["cyc_gen"] / max(v["generic"], 1), v["cyc_total"] / nb))

# decode stats
d_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
d_body = torch.empty(B * (bsz + 8), dtype=torch.uint8, device=dev)
d_out = torch.empty(B * bsz, dtype=torch.uint8, device=dev)
d_res = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
eng.dev_compact_records(d_stage.data_ptr(), stride, d_len.data_ptr(), B, d_off.data_ptr(), d_body.data_ptr(), d_body.numel(), 0)
L.plz4hip_debug_stats(out)
torch.cuda.synchronize(); t0 = time.time()
eng.dev_decode_records(d_body.data_ptr(), d_off.data_ptr(), B, bsz, False, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), 0)
torch.cuda.synchronize(); dt = time.time() - t0
L.plz4hip_debug_stats(out)
v = dict(zip(names, list(out)))
print("decode B=%d %.1f ms -> %.1f MiB/s; per block: batches %.0f, members %.0f, sequential steps %.0f (ml15 %.0f, ll15 %.0f)"
      % (B, dt * 1e3, B * 4 / dt, v["dbatch"] / B, v["dmemb"] / B, v["dseq"] / B, v["dseq_ml15"] / B, v["dseq_ll15"] / B))
