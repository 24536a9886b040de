"""Fuzz for the level-1 encoder in its external-segment modes (linked blocks, dictionaries: BASELINE config 5), which run the
grid batches since the segment is laid out right before the block: the device source (compiled for the CPU by tests/emu,
ascending and descending lane order) against the oracle's stream emulation (pinned to the real liblz4 in
tests/test_oracle_vs_ref.py).  (tests/test_emu_kernels.py::test_emu_both_dictionary_encoders_agree cross-checks it against the
one-sequence-per-batch dictionary encoder kept for the <= 4 KiB lookup mode.)
Not part of the test-suite (minutes); run from the repo root:  python tests/fuzz/fuzz_dict_emu.py [iters] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "fuzz")]
import fuzz_encode                 # noqa: E402
from emulib import Emu             # noqa: E402
from orclib import Oracle          # noqa: E402


def dict_table(dctx):
    return np.ctypeslib.as_array(dctx.table).astype(np.uint32).copy()


def main(iters=40, seed=5):
    orc, emu = Oracle(), Emu()
    rng = np.random.default_rng(seed)
    bad = tot = 0
    os.environ.setdefault("FUZZ_MAXN", "300000")
    for it in range(iters):
        big = fuzz_encode.make(rng, it)
        dlen = int(rng.choice([0, 5, 8, 9, 100, 4000, 30000, 65535, 65536, 70000]))
        user = np.ascontiguousarray(big[:dlen]) if rng.random() < 0.6 else rng.integers(0, 256, dlen, dtype=np.uint8)
        dct = np.ascontiguousarray(user[-65536:])
        dctx = orc.dict_ctx(user); tab = dict_table(dctx)
        for desc in (False, True):
            emu.set_descending(desc)
            # independent blocks against one dictionary (table copied in above 4 KiB)
            sizes = [int(x) for x in rng.choice([4097, 5000, 20000, 65536, 70000, 150000], 3)]
            for n in sizes:
                o = int(rng.integers(0, max(big.size - n, 1)))
                s = np.ascontiguousarray(big[o:o + n])
                for cap in (orc.bound(s.size), s.size, max(s.size // 3, 1)):
                    a, da = orc.compress_indie_dict(s, cap, dctx)
                    mode = 4 if dct.size < 8 else 2
                    b, db = emu.compress_dict(s, cap, dct if mode != 4 else None, mode, tab)
                    tot += 1
                    if a != b or not np.array_equal(da, db):
                        bad += 1; print("DICT MISMATCH it", it, "desc", desc, "dict", user.size, "n", s.size, "cap", cap, a, b)
            # a linked frame: block 0 fresh or under the dictionary, later blocks after LZ4_loadDict(previous tail)
            bsz = int(rng.choice([64 << 10, 100000, 256 << 10]))
            data = big[:min(big.size, 4 * bsz - int(rng.integers(0, bsz // 2)))]
            blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
            if rng.random() < 0.3 and len(blocks) > 2:
                blocks.insert(1, np.ascontiguousarray(big[:int(rng.choice([0, 3, 7, 8, 20, 500]))]))      # a short flush block
            prev = None
            for k, b in enumerate(blocks):
                tail = None if prev is None else prev[-65536:].copy()
                with_dict = prev is None and dct.size > 0 and rng.random() < 0.5
                a, da = orc.compress_linked(b, bsz, tail, dctx if with_dict else None)
                if tail is not None:
                    mode, seg = (1, tail) if tail.size >= 8 else (4, None)
                elif with_dict:
                    mode, seg = (4, None) if dct.size < 8 else ((2, dct) if b.size > 4096 else (3, dct))
                else:
                    mode, seg = 0, None
                r, dr = emu.compress_dict(b, bsz, seg, mode, tab if mode in (2, 3) else None)
                tot += 1
                if a != r or not np.array_equal(da, dr):
                    bad += 1; print("LINKED MISMATCH it", it, "desc", desc, "bsz", bsz, "k", k, "n", b.size, "mode", mode, a, r)
                prev = b
        emu.set_descending(False)
        if it % 5 == 4:
            print("iter", it + 1, "total", tot, "bad", bad, flush=True)
    print("total", tot, "bad", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(x) for x in sys.argv[1:])) else 0)
