"""One-off fuzz: the device HC source (compiled for the CPU by tests/emu) against the real LZ4_compress_HC in oracle/_ref,
levels 2..12, three capacities per input.  Not part of the test-suite (takes minutes); run from the repo root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import corpus                      # noqa: E402
from emulib import Emu             # noqa: E402
from orclib import Oracle, Ref     # noqa: E402
from plz4_amd import synth         # noqa: E402


def main(iters=400, seed=7):
    emu, ref, orc = Emu(), Ref(), Oracle()
    rng = np.random.default_rng(seed)
    bad = tot = 0
    for it in range(iters):
        n = int(rng.integers(0, 60000))
        kind = it % 5
        if kind == 0:
            src = corpus.structured(n, it)
        elif kind == 1:                                  # runs of short repeated patterns (pattern analysis, level >= 9)
            parts, have = [], 0
            while have < n:
                pat = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8)
                parts += [np.tile(pat, int(rng.integers(1, 3000))), rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)]
                have += parts[-1].size + parts[-2].size
            src = np.concatenate(parts)[:n].copy() if parts else np.zeros(0, np.uint8)
        elif kind == 2:
            src = rng.integers(0, 4, n, dtype=np.uint8)
        elif kind == 3:
            src = synth.text(n + 1)[:n].copy()
        else:
            base = rng.integers(0, 256, max(n // 7, 1), dtype=np.uint8)
            src = np.tile(base, 8)[:n].copy()
            if n:
                src[rng.integers(0, n, n // 50)] = 0
        for lvl in range(2, 13):
            for cap in (orc.bound(src.size), src.size, max(src.size // 2, 1)):
                a, da = ref.compress_hc(src, cap, lvl)
                b, db = emu.compress_hc(src, cap, lvl)
                tot += 1
                if a != b or not np.array_equal(da, db):
                    bad += 1
                    print("MISMATCH", it, kind, n, lvl, cap, a, b)
                if 3 <= lvl <= 11:                       # the same parsers on the chain built up front (HcWork::pre: what the kernels run)
                    b, db = emu.compress_hc_pre(src, cap, lvl)
                    tot += 1
                    if a != b or not np.array_equal(da, db):
                        bad += 1
                        print("MISMATCH pre", it, kind, n, lvl, cap, a, b)
                if 4 <= lvl <= 12:                       # the 63-candidates-per-round finder on the per-hash lists (hc_find_wider_lists)
                    b, db = emu.compress_hc_lists(src, cap, lvl)
                    tot += 1
                    if a != b or not np.array_equal(da, db):
                        bad += 1
                        print("MISMATCH lists", it, kind, n, lvl, cap, a, b)
                if 3 <= lvl <= 11:                       # round 3: segments walked at once and stitched, one candidate per lane, records + emit
                    for segs, mseg in ((1, 65536), (16, 256 + 64 * (it % 9))):
                        b, db = emu.compress_hc_lazy(src, cap, lvl, segs, mseg)
                        tot += 1
                        if a != b or not np.array_equal(da, db):
                            bad += 1
                            print("MISMATCH lazy", it, kind, n, lvl, cap, segs, mseg, a, b)
                if lvl == 2:                             # round 3: level 2 in batches over its two tables
                    b, db = emu.compress_hc_mid(src, cap)
                    tot += 1
                    if a != b or not np.array_equal(da, db):
                        bad += 1
                        print("MISMATCH mid", it, kind, n, cap, a, b)
                if lvl == 12:                            # the three-phase level-12 path (lz4hc12_device.inl): chain, per-position search, parser
                    for nc, nl in ((0, 1024), (5, 48)):  # nc: positions left to the parser's own search; nl: price-table entries in "LDS"
                        b, db = emu.compress_hc12(src, cap, nc, nl)
                        tot += 1
                        if a != b or not np.array_equal(da, db):
                            bad += 1
                            print("MISMATCH hc12", it, kind, n, cap, nc, nl, a, b)
    print("total", tot, "bad", bad)
    return bad


def main_gpu(iters=120, seed=7):
    """The kernels through the C ABI: one batch per level and capacity rule."""
    from plz4_amd._native import Engine
    ref, orc, eng = Ref(), Oracle(), Engine(0)
    rng = np.random.default_rng(seed)
    srcs = []
    for it in range(iters):
        n = int(rng.integers(0, 60000))
        kind = it % 3
        if kind == 0:
            srcs.append(corpus.structured(n, it))
        elif kind == 1:
            srcs.append(synth.text(n + 1)[:n].copy())
        else:
            pat = rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8)
            a = np.tile(pat, n // pat.size + 1)[:n].copy()
            if n:
                a[rng.integers(0, n, n // 60)] = 1
            srcs.append(a)
    bad = tot = 0
    for lvl in range(2, 13):
        for rule in (lambda n: orc.bound(n), lambda n: n, lambda n: max(n // 2, 1)):
            caps = [rule(s.size) for s in srcs]
            res, outs = eng.compress_batch(srcs, caps, level=lvl)
            for it, (s, cap, r, o) in enumerate(zip(srcs, caps, res, outs)):
                a, da = ref.compress_hc(s, cap, lvl); tot += 1
                if int(r) != a or not np.array_equal(o, da):
                    bad += 1; print("GPU MISMATCH", it, s.size, lvl, cap, a, int(r))
    eng.close()
    print("total", tot, "bad", bad)
    return bad


if __name__ == "__main__":
    if "--gpu" in sys.argv:
        sys.exit(1 if main_gpu(*(int(x) for x in sys.argv[1:] if x != "--gpu")) else 0)
    sys.exit(1 if main(*(int(x) for x in sys.argv[1:])) else 0)
