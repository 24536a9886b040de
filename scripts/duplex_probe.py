"""Diagnostics: what each role of k_l1_duplex costs by itself at the duplex kernel's residency (9 parser + 9 decoder waves per CU):
the full call, the call with an empty decode side, the call with a one-block encode side.  Times are HIP-event times of the whole
call (the encode side's emit kernels included: subtract them to get the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plz4_amd import synth
from plz4_amd._native import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
bsz = 4 << 20
eng = Engine(0); dev = torch.device("cuda:0")
pool = synth.make("T", 16 * bsz, bsz)
d_pool = torch.from_numpy(pool).to(dev)
d_src = torch.empty(B * bsz, dtype=torch.uint8, device=dev)
for r in range((B + 15) // 16):
    n = min(16, B - r * 16) * bsz
    d_src[r * 16 * bsz:r * 16 * bsz + n] = torch.roll(d_pool, -((r * 1000003) % d_pool.numel()))[:n]
stride = eng.stage_stride(bsz)
d_stage = torch.empty(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int32, device=dev)
d_off = torch.zeros(B + 1, dtype=torch.int64, device=dev); d_body = torch.empty(B * (bsz + 8), dtype=torch.uint8, device=dev)
d_out = torch.empty(B * bsz, dtype=torch.uint8, device=dev); d_res = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), s)
eng.dev_compact_records(d_stage.data_ptr(), stride, d_len.data_ptr(), B, d_off.data_ptr(), d_body.data_ptr(), d_body.numel(), s)
torch.cuda.synchronize()

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def duplex(src_bytes, ebsz, ndec):
    eng.dev_duplex_records(d_src.data_ptr(), src_bytes, ebsz, True, d_stage.data_ptr(), d_len.data_ptr(),
                           d_body.data_ptr(), d_off.data_ptr(), ndec, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), s)

print("B = %d blocks of 4 MiB each way" % B)
print("encode call alone (k_l1_parse<10> + emit)          %8.2f ms" % timed(lambda: eng.dev_encode_records(d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), s)))
print("k_decode_rec alone (24 waves per CU)               %8.2f ms" % timed(lambda: eng.dev_decode_records(d_body.data_ptr(), d_off.data_ptr(), B, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), s)))
print("duplex call, both sides                            %8.2f ms" % timed(lambda: duplex(B * bsz, bsz, B)))
print("duplex call, one 64 KiB block to encode            %8.2f ms   (the decode role by itself at 9 waves per CU)" % timed(lambda: duplex(64 << 10, 64 << 10, B)))
os.environ["PLZ4HIP_DUPLEX"] = "1,1"
# the parse role by itself at 9 waves per CU: one record to decode
one = torch.zeros(2, dtype=torch.int64, device=dev); one[1] = d_off[1]
print("duplex call, one record to decode                  %8.2f ms   (the parse role by itself at 9 waves per CU, + emit)" % timed(lambda: eng.dev_duplex_records(
    d_src.data_ptr(), B * bsz, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), d_body.data_ptr(), one.data_ptr(), 1, bsz, True,
    d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), s)))
eng.close()
