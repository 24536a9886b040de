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
# the C ABI itself: caller-owned buffers allocated (and touched) beforehand, a warm call first (staging is grown on demand)
from plz4_amd._native import _ptr_array, _i32, _i32p
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lens = _i32([s.size for s in srcs]); rl = np.zeros(nblk, dtype=np.int32)
rbuf = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(nblk)]
sp, rp = _ptr_array(srcs), _ptr_array(rbuf)
for rep in range(2):
    t0 = time.perf_counter()
    eng._chk(eng.L.plz4hip_encode_records_ex(eng.h, nblk, sp, _i32p(lens), bsz, level, 1, 1, d, None, -1, rp, _i32p(rl)))
    t1 = time.perf_counter()
recs = [r[:int(k)] for r, k in zip(rbuf, rl)]
ndec = min(nblk, 32)
win = np.zeros(65536, dtype=np.uint8); win[:dct.size] = dct[-65536:]
t2 = time.perf_counter()
res, st, outs, wl = eng.decode_records_ex([np.ascontiguousarray(r) for r in recs[:ndec]], bsz, True, linked=True, window=win, window_len=min(dct.size, 65536))
t3 = time.perf_counter()
assert all(int(s) == 0 for s in st) and all(np.array_equal(o, s) for o, s in zip(outs, srcs[:ndec]))
ratio = sum(r.size for r in recs) / (nblk * bsz)
print("config 5 (linked + 64 KiB dictionary), level %d, %d x 4MiB, ratio %.4f: encode_records_ex %.0f MiB/s (host buffers, warm); "
      "decode chain of %d blocks %.0f MiB/s" % (level, nblk, ratio, nblk * 4 / (t1 - t0), ndec, ndec * 4 / (t3 - t2)))
# many linked frames in one call: one wavefront per frame ("replicas only") -- F frames of 4 blocks each
nf = max(nblk // 4, 1)
frames = []
enc4 = [np.ascontiguousarray(r) for r in eng.encode_records_ex(srcs[:4], bsz, True, linked=True, d=d)]
for f in range(nf):
    frames.append(enc4)
wins = np.zeros((nf, 65536), dtype=np.uint8); wins[:] = dct[-65536:]
wl = np.full(nf, 65536, dtype=np.int32)
recs_all = [r for fr in frames for r in fr]
nrec = len(recs_all)
first = np.arange(0, nrec + 1, 4, dtype=np.int32)
rlen = _i32([r.size for r in recs_all]); res = np.zeros(nrec, dtype=np.int32); st = np.zeros(nrec, dtype=np.int32)
obuf = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(nrec)]                      # caller buffers ready before the clock starts
rp2, op2 = _ptr_array(recs_all), _ptr_array(obuf)
for rep in range(2):
    w2 = wins.copy(); l2 = wl.copy()
    t4 = time.perf_counter()
    eng._chk(eng.L.plz4hip_decode_records_chains(eng.h, nf, _i32p(first), rp2, _i32p(rlen), bsz, 1, w2.ctypes.data, _i32p(l2), op2, _i32p(res), _i32p(st)))
    t5 = time.perf_counter()
assert not st.any() and all(np.array_equal(o[:bsz], s) for o, s in zip(obuf[-4:], srcs[:4]))
print("config 5 decode, %d linked frames x 4 blocks x 4MiB in one call (plz4hip_decode_records_chains): %.0f MiB/s (host buffers, warm)"
      % (nf, nf * 16 / (t5 - t4)))
eng.dict_destroy(d); eng.close()
