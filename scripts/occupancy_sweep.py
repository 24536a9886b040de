"""Diagnostics: encode / decode kernel time against the number of blocks in flight (one wave per block), to see where the
chip saturates and how the tail round behaves.  Not part of the product or of the tests."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plz4_amd import synth, _native

eng = _native.Engine(0)
bsz = 4 << 20
pool = synth.text(16 * bsz)
dev = torch.device("cuda:0")
d_pool = torch.from_numpy(pool).to(dev)
counts = [int(x) for x in sys.argv[1:]] or [256, 512, 1024, 1280, 2048, 2304, 2560, 2816, 3072, 5120, 6144, 7680]
stride = eng.stage_stride(bsz)
print("resident waves: encoder", eng.resident_waves(False), "decoder", eng.resident_waves(True))
for B in counts:
    d_src = d_pool.repeat((B + 15) // 16)[:B * bsz].contiguous()
    d_stage = torch.empty(B * stride, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(B, dtype=torch.int32, device=dev)
    d_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    d_body = torch.empty(B * (bsz // 2), dtype=torch.uint8, device=dev)
    d_out = torch.empty(B * bsz, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    best_e = best_d = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), 0)
        torch.cuda.synchronize(); best_e = min(best_e, time.time() - t0)
        eng.dev_compact_records(d_stage.data_ptr(), stride, d_len.data_ptr(), B, d_off.data_ptr(), d_body.data_ptr(), d_body.numel(), 0)
        torch.cuda.synchronize(); t0 = time.time()
        eng.dev_decode_records(d_body.data_ptr(), d_off.data_ptr(), B, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), 0)
        torch.cuda.synchronize(); best_d = min(best_d, time.time() - t0)
    print("B=%5d  encode %7.1f ms  %6.2f GB/s   decode %7.1f ms  %6.2f GB/s" % (B, best_e * 1e3, B * bsz / best_e / 1e9, best_d * 1e3, B * bsz / best_d / 1e9), flush=True)
    del d_src, d_stage, d_body, d_out
