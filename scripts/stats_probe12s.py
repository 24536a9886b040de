"""Diagnostics: the level-12 parser's cycles by part (engine built with -DPLZ4_STATS into scripts/_build/).  Not part of the
product or of the tests."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plz4_amd import synth, _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
so = os.path.join(ROOT, "scripts", "_build", "libplz4hip_stats_s.so")
src = [os.path.join(ROOT, "plz4_amd", "csrc", f) for f in ("plz4hip.hip", "plz4hip_mgpu.cpp", "host/plz4_host.cpp", "host/host_capi.cpp")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared", "-DPLZ4_STATS", "-DPLZ4_STATS_SEARCH",
                       "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-o", so] + src)
_native.LIB_PATH = so
eng = _native.Engine(0)
L = eng.L
bsz = 4 << 20
pool = synth.text(16 * bsz)
dev = torch.device("cuda:0")
d_src = torch.from_numpy(pool).to(dev).repeat((B + 15) // 16)[:B * bsz].contiguous()
stride = eng.stage_stride(bsz)
d_stage = torch.empty(B * stride, dtype=torch.uint8, device=dev)
d_len = torch.zeros(B, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 24)()
names_parse = ["outer scan", "window init", "dp scan", "dp update", "traversal", "flush", "#dp updates", "#windows", "total", "#sequences", "#dp scans", "#blocks", "upd: select", "upd: reads", "upd: stores"]
for rep in range(2):
    L.plz4hip_debug_stats(out)
    torch.cuda.synchronize(); t0 = time.time()
    eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), 0, level=12)
    torch.cuda.synchronize(); dt = time.time() - t0
    L.plz4hip_debug_stats(out)
v = list(out)
# NOTE: the search kernel and the parser share the counter array: this probe reads the search kernel's slots (waves of 16 per block)
nw = max(v[16], 1)
names = ["filter runs", "filter lanes", "filter cyc", "count runs", "count lanes", "count cyc", "scan runs", "scan lanes", "scan cyc",
         "rank runs", "rank lanes", "rank cyc", "queue cyc", "queue runs", "trips", "wave cyc", "waves"]
print("B=%d %.0f ms; per WAVE of the search kernel (16 waves per block):" % (B, dt * 1e3))
for i, k in enumerate(names):
    print("  %-12s %14.0f" % (k, v[i] / nw))
