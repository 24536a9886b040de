"""Host-buffer ABI (plz4hip_encode_records from pageable caller memory) at the HC levels: MiB/s for one warm call.
   python scripts/host_hc_rate.py [blocks] [levels...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plz4_amd import synth
from plz4_amd._native import Engine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
levels = [int(x) for x in sys.argv[2:]] or [5, 9, 11, 12]
bsz = 4 << 20
base = synth.make("T", 64 * bsz, bsz)
srcs = [base[(i % 64) * bsz:(i % 64 + 1) * bsz] for i in range(nb)]
eng = Engine(0)
for lvl in levels:
    eng.encode_records(srcs[:64], bsz, True, level=lvl)           # warm: workspaces and staging
    for rep in range(2):
        t = time.time(); recs = eng.encode_records(srcs, bsz, True, level=lvl); dt = time.time() - t
    print("level %d: %d x 4 MiB through host memory: %.0f MiB/s encode (%.2f s), ratio %.3f" % (lvl, nb, nb * 4 / dt, dt, sum(r.size for r in recs) / (nb * bsz)), flush=True)
eng.close()
