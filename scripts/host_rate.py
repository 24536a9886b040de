"""PCIe-inclusive rate of the host-buffer ABI (pinned staging + H2D + kernels + D2H, synchronous), for DESIGN.md §6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from plz4_amd import synth, host
from plz4_amd._native import Engine

bsz = 4 << 20
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pool = synth.text(16 * bsz)
srcs = [pool[(i % 16) * bsz:(i % 16 + 1) * bsz] for i in range(nblk)]
eng = Engine(0)
eng.encode_records(srcs[:8], bsz, True)
t0 = time.perf_counter(); recs = eng.encode_records(srcs, bsz, True); t1 = time.perf_counter()
res, st, outs = eng.decode_records([np.ascontiguousarray(r) for r in recs], bsz, True); t2 = time.perf_counter()
assert all(int(s) == 0 for s in st)
mib = nblk * 4
print("ABI B host buffers, %d x 4MiB: encode_records %.0f MiB/s, decode_records %.0f MiB/s, enc+dec %.0f MiB/s"
      % (nblk, mib / (t1 - t0), mib / (t2 - t1), mib / (t2 - t0)))
# the C ABI itself: caller-owned buffers allocated (and touched) beforehand, a warm call first (staging is grown on demand)
import ctypes as C
from plz4_amd._native import _ptr_array, _i32, _i32p
lens = _i32([s.size for s in srcs]); rl = np.zeros(nblk, dtype=np.int32)
rbuf = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(nblk)]
sp, rp = _ptr_array(srcs), _ptr_array(rbuf)
for rep in range(2):
    t0 = time.perf_counter()
    eng._chk(eng.L.plz4hip_encode_records(eng.h, nblk, sp, _i32p(lens), bsz, 1, 1, rp, _i32p(rl)))
    t1 = time.perf_counter()
rlen = _i32([int(k) for k in rl]); res = np.zeros(nblk, dtype=np.int32); st = np.zeros(nblk, dtype=np.int32)
obuf = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(nblk)]
op = _ptr_array(obuf)
for rep in range(2):
    t2 = time.perf_counter()
    eng._chk(eng.L.plz4hip_decode_records(eng.h, nblk, rp, _i32p(rlen), bsz, 1, op, _i32p(res), _i32p(st)))
    t3 = time.perf_counter()
assert all(int(x) == 0 for x in st) and all(np.array_equal(o[:bsz], s) for o, s in zip(obuf[:4], srcs[:4]))
print("C ABI, caller buffers ready, warm: encode_records %.0f MiB/s, decode_records %.0f MiB/s" % (mib / (t1 - t0), mib / (t3 - t2)))
if len(sys.argv) > 2 and sys.argv[2] == "abi":
    sys.exit(0)
e = host.hip_engine(0)
data = np.concatenate(srcs).tobytes()
t0 = time.perf_counter()
w = host.Writer(e, parallel=4, block_checksum=True, content_checksum=False, gpu_batch=nblk); w.write(data); w.close(); f = w.output()
t1 = time.perf_counter()
n, out, err = host.Reader(e, f, gpu_batch=nblk).write_to()
t2 = time.perf_counter()
assert not err and out == data
print("host layer NewWriter/NewReader over memory buffers: write %.0f MiB/s, read %.0f MiB/s" % (mib / (t1 - t0), mib / (t2 - t1)))
