"""BASELINE configs[4]: linked-block mode + 64 KiB dictionary prefix, through the host-buffer ABI (B').  The encoder shards by
block (block i needs only the last 64 KiB of block i-1's source); the decoder is one serial chain per frame, so decode
throughput comes from decoding many frames at once -- here: F frames of N blocks each, one decode call per frame is what the
ABI offers today, so the decode figure below is the single-chain rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from plz4_amd import synth
from plz4_amd._native import Engine

bsz = 4 << 20
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pool = synth.text(16 * bsz)
srcs = [pool[(i % 16) * bsz:(i % 16 + 1) * bsz] for i in range(nblk)]
dct = synth.text(65536, seed=99)
eng = Engine(0)
d = eng.dict_create(dct)
eng.encode_records_ex(srcs[:4], bsz, True, linked=True, d=d)
t0 = time.perf_counter(); recs = eng.encode_records_ex(srcs, bsz, True, linked=True, d=d); t1 = time.perf_counter()
ndec = min(nblk, 32)
win = np.zeros(65536, dtype=np.uint8); win[:dct.size] = dct[-65536:]
t2 = time.perf_counter()
res, st, outs, wl = eng.decode_records_ex([np.ascontiguousarray(r) for r in recs[:ndec]], bsz, True, linked=True, window=win, window_len=min(dct.size, 65536))
t3 = time.perf_counter()
assert all(int(s) == 0 for s in st) and all(np.array_equal(o, s) for o, s in zip(outs, srcs[:ndec]))
ratio = sum(r.size for r in recs) / (nblk * bsz)
print("config 5 (linked + 64 KiB dictionary), %d x 4MiB, ratio %.4f: encode_records_ex %.0f MiB/s (host buffers); "
      "decode chain of %d blocks %.0f MiB/s" % (nblk, ratio, nblk * 4 / (t1 - t0), ndec, ndec * 4 / (t3 - t2)))
eng.dict_destroy(d); eng.close()
