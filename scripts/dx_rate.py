"""The few-block decode path (lz4_dx_device.inl) against one wave per block: plz4hip_decompress_batch of 1 / 4 / 16 / 64 blocks of
4 MiB through host buffers (staging copies and PCIe included, warm), and the kernels alone on device-resident blocks (HIP events).
    python scripts/dx_rate.py            (PLZ4HIP_DX_MAX_BLOCKS=0 in a second process gives the old path)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from plz4_amd import synth
from plz4_amd._native import Engine
import orclib

bsz = 4 << 20
orc = orclib.Oracle()
pool = synth.text(16 * bsz)
blocks = []
for i in range(16):
    src = pool[i * bsz:(i + 1) * bsz]
    c, comp = orc.compress_fast(src, orc.bound(bsz))
    blocks.append((src, np.ascontiguousarray(comp[:c])))
eng = Engine(0)
out = {"dx_max_blocks": os.environ.get("PLZ4HIP_DX_MAX_BLOCKS", "128 (default)"), "host_buffers_ms": {}, "device_resident_ms": {}}
for nb in (1, 4, 16, 64):
    comps = [blocks[i % 16][1] for i in range(nb)]
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        res, outs = eng.decompress_batch(comps, [bsz + 8] * nb)
        best = min(best, time.perf_counter() - t0)
    assert all(int(r) == bsz for r in res) and np.array_equal(outs[nb - 1], blocks[(nb - 1) % 16][0])
    out["host_buffers_ms"][str(nb)] = round(best * 1e3, 2)
    # device-resident: the kernels alone
    dev = torch.device("cuda:0")
    stride = 5 << 20
    d_src = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    for i, c in enumerate(comps): d_src[i * stride:i * stride + c.size] = torch.from_numpy(c).to(dev)
    d_len = torch.tensor([c.size for c in comps], dtype=torch.int32, device=dev)
    d_cap = torch.full((nb,), bsz + 8, dtype=torch.int32, device=dev)
    d_dst = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(nb, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream()
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        eng._chk(eng.L.plz4hip_dev_decompress(eng.h, nb, d_src.data_ptr(), stride, d_len.data_ptr(), d_dst.data_ptr(), stride, d_cap.data_ptr(), d_res.data_ptr(), s.cuda_stream))
        e1.record(s); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert int(d_res.sum().item()) == nb * bsz
    assert np.array_equal(d_dst[(nb - 1) * stride:(nb - 1) * stride + bsz].cpu().numpy(), blocks[(nb - 1) % 16][0])
    out["device_resident_ms"][str(nb)] = round(best, 3)
out["records_host_buffers_ms"] = {}
for nb in (1, 16):                                              # frame records with block checksums: what plz4's reader hands over
    recs = [np.ascontiguousarray(orc.block_record(blocks[i % 16][0], bsz, True)) for i in range(nb)]
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        res, st, outs = eng.decode_records(recs, bsz, True)
        best = min(best, time.perf_counter() - t0)
    assert not any(int(k) for k in st) and np.array_equal(outs[nb - 1], blocks[(nb - 1) % 16][0])
    out["records_host_buffers_ms"][str(nb)] = round(best * 1e3, 2)
out["MiBps_host_16"] = round(16 * 4 / (out["host_buffers_ms"]["16"] * 1e-3), 1)
out["MiBps_device_16"] = round(16 * 4 / (out["device_resident_ms"]["16"] * 1e-3), 1)
print(json.dumps(out))
eng.close()
