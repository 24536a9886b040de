"""A/B of k_decode_rec between builds of the library: dec_ab.py <lib.so>   (6144 x 4 MiB T records, device resident)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plz4_amd import synth, _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from plz4_amd._native import Engine
BSZ = 4 << 20; B = 6144; S = B * BSZ
dev = torch.device("cuda", 0)
d_pool = torch.from_numpy(synth.make("T", 16 * BSZ, BSZ)).to(dev)
d_src = d_pool.repeat(B // 16)[:S].contiguous()
eng = Engine(0)
cap = int(S * 0.42)
body = torch.empty(cap, dtype=torch.uint8, device=dev); off = torch.zeros(B + 1, dtype=torch.int64, device=dev); ln = torch.zeros(B, dtype=torch.int32, device=dev)
out = torch.empty(S, dtype=torch.uint8, device=dev); res = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(device=dev)
eng.dev_encode_body(d_src.data_ptr(), S, BSZ, True, body.data_ptr(), cap, off.data_ptr(), ln.data_ptr(), s.cuda_stream, level=1)
ts = []
for rep in range(4):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s); eng.dev_decode_records(body.data_ptr(), off.data_ptr(), B, BSZ, True, out.data_ptr(), BSZ, BSZ, res.data_ptr(), st.data_ptr(), s.cuda_stream); e1.record(s)
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("%s: k_decode_rec over 6144 blocks: %s ms; ok %s" % (os.path.basename(_native.LIB_PATH), " ".join("%.1f" % t for t in ts), bool((st == 0).all().item()) and torch.equal(out[:1 << 28], d_src[:1 << 28])))
