"""Fuzz for the dictionary / linked-block encoder and decoder (BASELINE config 5) on the GPU, through the C ABI, against the
oracle's stream emulation (itself pinned to the real liblz4 in tests/test_oracle_vs_ref.py).  Run from the repo root on a GPU
box: python tests/fuzz/fuzz_dict.py [iters] [seed]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "fuzz")]
import fuzz_encode                 # noqa: E402
from orclib import Oracle          # noqa: E402
from plz4_amd._native import Engine  # noqa: E402


def record(orc, ret, comp, src, checksum):
    payload, word = (src, 0x80000000 | src.size) if ret == 0 else (comp, comp.size)
    rec = np.uint32(word).tobytes() + payload.tobytes()
    return rec + (np.uint32(orc.xxh32(payload)).tobytes() if checksum else b"")


def main(iters=40, seed=5):
    orc, eng = Oracle(), Engine(0)
    rng = np.random.default_rng(seed)
    bad = tot = 0
    for it in range(iters):
        big = fuzz_encode.make(rng, it)
        dlen = int(rng.choice([0, 5, 8, 9, 100, 4000, 30000, 65535, 65536, 70000]))
        user = np.ascontiguousarray(big[:dlen]) if rng.random() < 0.5 else rng.integers(0, 256, dlen, dtype=np.uint8)
        # independent blocks against one dictionary: sizes on both sides of the 4 KiB switch
        sizes = [int(x) for x in rng.choice([0, 1, 12, 13, 100, 4095, 4096, 4097, 20000, 65536, 70000, 150000], 6)]
        srcs = [np.ascontiguousarray(big[o:o + n]) for o, n in zip(rng.integers(0, big.size // 2, len(sizes)), sizes)]
        sizes = [s.size for s in srcs]
        dctx = orc.dict_ctx(user); d = eng.dict_create(np.ascontiguousarray(user))
        for caps in ([orc.bound(n) for n in sizes], [max(n, 1) for n in sizes], [max(n // 3, 1) for n in sizes]):
            res, outs = eng.compress_batch_dict(srcs, caps, d)
            for s, c, r, o in zip(srcs, caps, res, outs):
                a, da = orc.compress_indie_dict(s, c, dctx); tot += 1
                if int(r) != a or not np.array_equal(o, da):
                    bad += 1; print("DICT MISMATCH", it, user.size, s.size, c, a, int(r))
        comps = [np.ascontiguousarray(orc.compress_indie_dict(s, orc.bound(s.size), dctx)[1]) for s in srcs]
        dd = np.ascontiguousarray(user[-65536:])
        for caps in ([n + 8 for n in sizes], sizes, [max(n - 1, 0) for n in sizes]):
            res, outs = eng.decompress_batch_dict(comps, caps, d)
            for cp, cap, r, o in zip(comps, caps, res, outs):
                a, da = orc.decompress_safe_dict(cp, cap, dd); tot += 1
                if int(r) != a or (a >= 0 and not np.array_equal(o, da[:a])):
                    bad += 1; print("DICT DECODE MISMATCH", it, user.size, cp.size, cap, a, int(r))
        # a linked frame (with or without the dictionary), records against the oracle's stream, then the decode chain
        bsz = int(rng.choice([64 << 10, 256 << 10]))
        nb = int(rng.integers(2, 6))
        data = big[:min(big.size, nb * bsz - int(rng.integers(0, bsz // 2)))]
        blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
        with_dict = rng.random() < 0.6 and user.size > 0
        want, prev = [], None
        for b in blocks:
            tail = None if prev is None else prev[-65536:].copy()
            r, c = orc.compress_linked(b, bsz, tail, dctx if (prev is None and with_dict) else None)
            want.append(record(orc, r, c, b, True)); prev = b
        got = eng.encode_records_ex(blocks, bsz, True, linked=True, d=d if with_dict else None)
        for i, (g, w) in enumerate(zip(got, want)):
            tot += 1
            if g.tobytes() != w:
                bad += 1; print("LINKED MISMATCH", it, bsz, i, blocks[i].size, with_dict, user.size)
        window = np.zeros(65536, dtype=np.uint8); wl = 0
        if with_dict:
            wl = min(user.size, 65536); window[:wl] = user[-wl:]
        res, st, outs, wl = eng.decode_records_ex([np.ascontiguousarray(g) for g in got], bsz, True, linked=True, window=window, window_len=wl)
        for i, (b, k, o) in enumerate(zip(blocks, st, outs)):
            tot += 1
            if int(k) != 0 or not np.array_equal(o, b):
                # the reference itself garbles a linked frame after a stored block (DictT.Update quirk): only flag when no block is stored
                if not any(g[3] & 0x80 for g in got):
                    bad += 1; print("LINKED DECODE MISMATCH", it, bsz, i, int(k))
        eng.dict_destroy(d)
    eng.close()
    print("total", tot, "bad", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(x) for x in sys.argv[1:])) else 0)
