"""A/B of the host-buffer ABI's rate between two builds of the library: host_rate_ab.py <lib.so> [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from plz4_amd import synth, _native
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from plz4_amd._native import Engine, _ptr_array, _i32, _i32p
bsz = 4 << 20
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
pool = synth.text(16 * bsz)
srcs = [pool[(i % 16) * bsz:(i % 16 + 1) * bsz] for i in range(nblk)]
eng = Engine(0)
lens = _i32([s.size for s in srcs]); rl = np.zeros(nblk, dtype=np.int32)
rbuf = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(nblk)]
sp, rp = _ptr_array(srcs), _ptr_array(rbuf)
te = []
for rep in range(3):
    t0 = time.perf_counter(); eng._chk(eng.L.plz4hip_encode_records(eng.h, nblk, sp, _i32p(lens), bsz, 1, 1, rp, _i32p(rl))); te.append(time.perf_counter() - t0)
rlen = _i32([int(k) for k in rl]); res = np.zeros(nblk, dtype=np.int32); st = np.zeros(nblk, dtype=np.int32)
obuf = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(nblk)]
op = _ptr_array(obuf)
td = []
for rep in range(3):
    t2 = time.perf_counter(); eng._chk(eng.L.plz4hip_decode_records(eng.h, nblk, rp, _i32p(rlen), bsz, 1, op, _i32p(res), _i32p(st))); td.append(time.perf_counter() - t2)
print("%s: %d x 4 MiB: encode_records %s ms, decode_records %s ms" % (os.path.basename(_native.LIB_PATH), nblk, " ".join("%.0f" % (x * 1e3) for x in te), " ".join("%.0f" % (x * 1e3) for x in td)))
