"""Diagnostics: build the engine with -DPLZ4_STATS into scripts/_build/libplz4hip_stats.so, run a level-1 encode over B blocks
of T text and print the parse kernel's per-section cycle counters (lz4_seq_device.inl).  Not part of the product or the tests."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plz4_amd import synth, _native, build as B_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
so = os.path.join(ROOT, "scripts", "_build", "libplz4hip_stats.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared", "-DPLZ4_STATS",
                       "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-o", so] + B_.sources())
_native.LIB_PATH = so
eng = _native.Engine(0)
L = eng.L
names = ["batch", "cyc_mem", "cyc_lds", "cyc_refresh", "cyc_cmp", "cyc_walk", "cyc_tail", "repair", "prime", "generic", "cyc_total", "blocks", "seq", "cyc_gen", "cyc_nh", "cyc_hop", "cyc_e", "cyc_slow", "hops"]
bsz = 4 << 20
pool = synth.text(16 * bsz)
dev = torch.device("cuda:0")
d_pool = torch.from_numpy(pool).to(dev)
d_src = d_pool.repeat((B + 15) // 16)[:B * bsz].contiguous()
stride = eng.stage_stride(bsz)
d_stage = torch.empty(B * stride, dtype=torch.uint8, device=dev)
d_len = torch.zeros(B, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 24)()
for rep in range(2):
    L.plz4hip_debug_stats(out)
    torch.cuda.synchronize(); t0 = time.time()
    eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), 0)
    torch.cuda.synchronize(); dt = time.time() - t0
    L.plz4hip_debug_stats(out)
    v = dict(zip(names, list(out)))
    nb = max(v["blocks"], 1); g = max(v["batch"], 1)
    print("B=%d  %.1f ms  -> %.1f MiB/s (instrumented)" % (B, dt * 1e3, B * 4 / dt))
    print("   per block: batches %.0f primes %.1f generic %.1f repairs %.0f seqs %.0f cycles %.3e" % (v["batch"] / nb, v["prime"] / nb, v["generic"] / nb, v["repair"] / nb, v["seq"] / nb, v["cyc_total"] / nb))
    print("   per grid batch (cycles): memory wait %.0f | peek + compare %.0f | [commit + verify + gather + compare %.0f, inside the rounds] | requests %.0f | rounds (walk, commit, verify; second rounds) %.0f | records + state %.0f | sum of the outer sections %.0f"
          % (v["cyc_mem"] / g, v["cyc_lds"] / g, v["cyc_refresh"] / g, v["cyc_cmp"] / g, v["cyc_walk"] / g, v["cyc_tail"] / g,
             (v["cyc_mem"] + v["cyc_lds"] + v["cyc_cmp"] + v["cyc_walk"] + v["cyc_tail"]) / g))
    print("   inside the rounds: hops %.0f (%.1f hop slots) | executed lanes %.0f | batches that took a second round %.1f%%"
          % (v["cyc_hop"] / g, v["hops"] / g, v["cyc_e"] / g, 100.0 * v["repair"] / g))
