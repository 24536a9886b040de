"""Diagnostics: -DPLZ4_STATS build, cycle stamps around the sections of the record decoder's vector batch (lz4_device.inl)."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plz4_amd import synth, _native, build as B_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
so = os.path.join(ROOT, "scripts", "_build", "libplz4hip_stats.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared", "-DPLZ4_STATS",
                       "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-o", so] + B_.sources())
_native.LIB_PATH = so
eng = _native.Engine(0); L = eng.L
bsz = 4 << 20
pool = synth.text(16 * bsz); dev = torch.device("cuda:0")
d_src = torch.from_numpy(pool).to(dev).repeat((B + 15) // 16)[:B * bsz].contiguous()
stride = eng.stage_stride(bsz)
d_stage = torch.empty(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int32, device=dev)
eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), 0)
d_off = torch.zeros(B + 1, dtype=torch.int64, device=dev); d_body = torch.empty(B * (bsz + 8), dtype=torch.uint8, device=dev)
d_out = torch.empty(B * bsz, dtype=torch.uint8, device=dev); d_res = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
eng.dev_compact_records(d_stage.data_ptr(), stride, d_len.data_ptr(), B, d_off.data_ptr(), d_body.data_ptr(), d_body.numel(), 0)
out = (C.c_ulonglong * 24)()
torch.cuda.synchronize(); L.plz4hip_debug_stats(out)
t0 = time.time()
eng.dev_decode_records(d_body.data_ptr(), d_off.data_ptr(), B, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), 0)
torch.cuda.synchronize(); dt = time.time() - t0
L.plz4hip_debug_stats(out); v = list(out); g = max(v[6], 1)
print("decode B=%d %.1f ms (instrumented); batches per block %.0f, members per batch %.1f, far matches per batch %.1f, near rounds per batch %.2f"
      % (B, dt * 1e3, v[6] / B, v[20] / max(v[19], 1), v[7] / g, v[8] / g))
print("   per batch (cycles): window + token parse %.0f | chain walk + offsets %.0f | requests + literals %.0f | far matches %.0f | near rounds %.0f | sweep + tail %.0f"
      % (v[0] / g, v[1] / g, v[2] / g, v[3] / g, v[4] / g, v[5] / g))
