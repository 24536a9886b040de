"""One-off fuzz: the device HC source with dictionaries / linked blocks (compiled for the CPU by tests/emu) against the real
liblz4 in oracle/_ref driven the way clz4.go drives it (StreamCtxHC, StreamLinkedCtxHC), levels 2..12.
Not part of the test-suite (takes minutes); run from the repo root:  python tests/fuzz/fuzz_hc_dict.py [iters] [seed]
PLZ4_FUZZ_LISTS=1: the blocks behind an external segment (linked blocks, blocks > 4 KiB under a dictionary) at levels 3..12 take
the path the kernels run since round 4 -- chain and lists over segment + block, the walk in 1..16 segments, stitched, record emit
(emu_compress_hc_lazy_ext) -- instead of the one-thread parsers of lz4hc_device.inl.  `--gpu`: the kernels through the C ABI."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import corpus                      # noqa: E402
from emulib import Emu             # noqa: E402
from orclib import Oracle, Ref     # noqa: E402
from plz4_amd import synth         # noqa: E402


def gen(rng, n, kind, it):
    if n == 0:
        return np.zeros(0, np.uint8)
    if kind == 0:
        return corpus.structured(n, it)
    if kind == 1:                                      # runs of short repeated patterns (pattern analysis across the boundary)
        parts, have = [], 0
        while have < n:
            pat = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8)
            parts += [np.tile(pat, int(rng.integers(1, 3000))), rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)]
            have += parts[-1].size + parts[-2].size
        return np.concatenate(parts)[:n].copy()
    if kind == 2:
        return rng.integers(0, 4, n, dtype=np.uint8)
    if kind == 3:
        return synth.text(n + 1 + it % 97)[it % 97:it % 97 + n].copy()
    base = rng.integers(0, 256, max(n // 7, 1), dtype=np.uint8)
    a = np.tile(base, 8)[:n].copy()
    a[rng.integers(0, n, n // 50)] = 0
    return a


def mode_for(n):
    return 1 if n > 4096 else 2


LISTS = os.environ.get("PLZ4_FUZZ_LISTS") == "1"


def with_segment(emu, rng, b, cap, lvl, seg, mode):
    """One block under the kernels' priming rule `mode` (1: external segment, 2: dictionary context)."""
    if LISTS and mode == 1 and lvl >= 2:
        segs, minseg = [(1, 65536), (4, 1000), (16, 300), (64, 100)][int(rng.integers(0, 4))]
        return emu.compress_hc_lazy_ext(b, cap, lvl, seg, segs, minseg)
    return emu.compress_hc_dict(b, cap, lvl, seg, mode)


def main(iters=200, seed=11, levels=tuple(range(2, 13))):
    emu, ref, orc = Emu(), Ref(), Oracle()
    rng = np.random.default_rng(seed)
    bad = tot = 0
    for it in range(iters):
        kind = it % 5
        # a stream whose pieces share content, so that dictionary / previous-block matches exist
        dlen = int(rng.choice([0, 1, 3, 4, 5, 8, 9, 100, 4000, 40000, 65536, 70000]))
        nblk = 3
        sizes = [int(rng.choice([0, 1, 11, 13, 200, 4096, 4097, 5000, 20000, 66000])) for _ in range(nblk)]
        whole = gen(rng, dlen + sum(sizes), kind, it)
        if kind == 1 and whole.size > 40:               # make one run straddle the first boundary
            whole[max(dlen - 20, 0):dlen + 20] = whole[max(dlen - 20, 0)]
        dct = whole[:dlen].copy()
        blocks, o = [], dlen
        for sz in sizes:
            blocks.append(whole[o:o + sz].copy()); o += sz
        dtrunc = dct[-65536:] if dct.size > 65536 else dct
        for lvl in levels:
            keep, daddr = ref.new_dict_ctx_hc(dtrunc, lvl)
            # --- independent blocks + dictionary: StreamCtxHC, one reused stream
            comp = ref.stream_ctx_hc(lvl, daddr)
            for b in blocks:
                for cap in (orc.bound(b.size), b.size, max(b.size // 2, 1)):
                    a, da = comp(b, cap)
                    r, dr = with_segment(emu, rng, b, cap, lvl, dtrunc, mode_for(b.size))
                    tot += 1
                    if a != r or not np.array_equal(da, dr):
                        bad += 1; print("INDIE MISMATCH it", it, "kind", kind, "lvl", lvl, "dlen", dlen, "n", b.size, "cap", cap, a, r)
            # --- linked blocks, with and without the dictionary: StreamLinkedCtxHC, capacity = block size as in the frame path
            for use_dict in (True, False):
                lcomp = ref.stream_linked_ctx_hc(lvl, daddr if use_dict else None)
                prev = None
                for k, b in enumerate(blocks):
                    tail = None if prev is None else prev[-65536:]
                    cap = max(b.size, 1)
                    a, da = lcomp(b, cap, tail)
                    if tail is not None:
                        r, dr = with_segment(emu, rng, b, cap, lvl, np.ascontiguousarray(tail), 1)
                    elif use_dict:
                        r, dr = with_segment(emu, rng, b, cap, lvl, dtrunc, mode_for(b.size))
                    else:
                        r, dr = emu.compress_hc(b, cap, lvl)
                    tot += 1
                    if a != r or not np.array_equal(da, dr):
                        bad += 1; print("LINKED MISMATCH it", it, "kind", kind, "lvl", lvl, "dict", use_dict, "k", k, "n", b.size, a, r)
                    prev = b
        if it % 10 == 9:
            print("iter", it + 1, "total", tot, "bad", bad, flush=True)
    print("total", tot, "bad", bad)
    return bad


def main_gpu(iters=40, seed=31, levels=tuple(range(2, 13))):
    """The kernels through the C ABI: compress_batch_dict (StreamCtxHC: a dictionary, independent blocks, both sides of the 4 KiB
    switch in one call) and encode_records_ex (StreamLinkedCtxHC: linked frames with and without a dictionary), every level."""
    from plz4_amd._native import Engine
    import hcdict
    ref, orc, eng = Ref(), Oracle(), Engine(0)
    rng = np.random.default_rng(seed)
    bad = tot = 0
    for it in range(iters):
        kind = it % 5
        dlen = int(rng.choice([0, 3, 5, 100, 4000, 40000, 65536, 70000]))
        sizes = [int(rng.choice([0, 13, 200, 4096, 4097, 5000, 20000, 66000, 150000])) for _ in range(5)]
        whole = gen(rng, dlen + sum(sizes), kind, it)
        if kind == 1 and whole.size > 40:
            whole[max(dlen - 20, 0):dlen + 20] = whole[max(dlen - 20, 0)]
        dct = whole[:dlen].copy()
        blocks, o = [], dlen
        for sz in sizes:
            blocks.append(np.ascontiguousarray(whole[o:o + sz])); o += sz
        dtrunc = np.ascontiguousarray(dct[-65536:] if dct.size > 65536 else dct)
        d = eng.dict_create(np.ascontiguousarray(dct))
        for lvl in levels:
            keep, daddr = ref.new_dict_ctx_hc(dtrunc, lvl)
            comp = ref.stream_ctx_hc(lvl, daddr)
            for rule in (lambda n: orc.bound(n), lambda n: max(n, 1), lambda n: max(n // 2, 1)):
                caps = [rule(b.size) for b in blocks]
                res, outs = eng.compress_batch_dict(blocks, caps, d, level=lvl)
                for k, (b, cap, r, out) in enumerate(zip(blocks, caps, res, outs)):
                    a, da = comp(b, cap); tot += 1
                    if int(r) != a or not np.array_equal(out, da):
                        bad += 1; print("GPU INDIE MISMATCH it", it, "kind", kind, "lvl", lvl, "dlen", dlen, "n", b.size, "cap", cap, a, int(r))
            bsz = max(max(sizes), 1)
            for use_dict in (True, False):
                want, _ = hcdict.ref_records(ref, orc, blocks, bsz, lvl, True, dct if use_dict else None)
                got = eng.encode_records_ex(blocks, bsz, True, linked=True, d=d if use_dict else None, level=lvl)
                for k, (g_, w) in enumerate(zip(got, want)):
                    tot += 1
                    if g_.tobytes() != w:
                        bad += 1; print("GPU LINKED MISMATCH it", it, "kind", kind, "lvl", lvl, "dict", use_dict, "k", k, "n", blocks[k].size)
        eng.dict_destroy(d)
        if it % 5 == 4:
            print("iter", it + 1, "total", tot, "bad", bad, flush=True)
    eng.close()
    print("total", tot, "bad", bad)
    return bad


if __name__ == "__main__":
    if "--gpu" in sys.argv:
        sys.exit(1 if main_gpu(*(int(x) for x in sys.argv[1:] if x != "--gpu")) else 0)
    sys.exit(1 if main(*(int(x) for x in sys.argv[1:])) else 0)
