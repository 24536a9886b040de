"""Fuzz for the level-1 encoder: the device source (compiled for the CPU by tests/emu, ascending and descending lane order)
against the real LZ4_compress_fast in oracle/_ref on inputs above the byU16 limit (so the grid batches run), three
capacities per input.  Not part of the test-suite (minutes); run from the repo root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import corpus                      # noqa: E402
from emulib import Emu             # noqa: E402
from orclib import Oracle, Ref     # noqa: E402
from plz4_amd import synth         # noqa: E402


def make(rng, it):
    n = int(rng.integers(66000, int(os.environ.get("FUZZ_MAXN", "400000"))))
    kind = it % 8
    if kind == 6:                                    # noise with sparse repeats: long searches, stride > 1, then a match
        a = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(int(rng.integers(1, 60))):
            ln = int(rng.integers(4, 400)); s0 = int(rng.integers(0, n - ln)); d0 = int(rng.integers(0, n - ln))
            a[d0:d0 + ln] = a[s0:s0 + ln].copy()
        return a
    if kind == 7:                                    # compressible and incompressible stretches taking turns
        parts, have = [], 0
        while have < n:
            ln = int(rng.integers(50, 30000))
            parts.append(rng.integers(0, 256, ln, dtype=np.uint8) if rng.random() < 0.5 else synth.text(ln + 1, seed=int(rng.integers(1, 1 << 30)))[:ln])
            have += ln
        return np.concatenate(parts)[:n].copy()
    if kind == 0:
        return corpus.structured(n, it)
    if kind == 1:                                    # short periodic runs: twins inside a window, long matches
        parts, have = [], 0
        while have < n:
            pat = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
            parts += [np.tile(pat, int(rng.integers(1, 200))), rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8)]
            have += parts[-1].size + parts[-2].size
        return np.concatenate(parts)[:n].copy()
    if kind == 2:
        return rng.integers(0, 3, n, dtype=np.uint8)
    if kind == 3:
        return synth.text(n + 1, seed=it)[:n].copy()
    if kind == 4:                                    # a small vocabulary of 4..9-byte words: dense hash collisions and twins
        words = [rng.integers(97, 123, int(rng.integers(4, 10)), dtype=np.uint8) for _ in range(int(rng.integers(3, 60)))]
        out, have = [], 0
        while have < n:
            w = words[int(rng.integers(0, len(words)))]; out.append(w); have += w.size
        return np.concatenate(out)[:n].copy()
    base = rng.integers(0, 256, max(n // 9, 1), dtype=np.uint8)
    a = np.tile(base, 10)[:n].copy()
    a[rng.integers(0, n, n // 40)] = 7
    return a


def main_gpu(iters, seed):
    """The same inputs through the C ABI in one batch per capacity rule, against the real reference."""
    from plz4_amd._native import Engine
    ref, orc, eng = Ref(), Oracle(), Engine(0)
    rng = np.random.default_rng(seed)
    srcs = [make(rng, it) for it in range(iters)]
    bad = tot = 0
    for rule in (lambda n: orc.bound(n), lambda n: n, lambda n: max(n // 3, 1)):
        caps = [rule(s.size) for s in srcs]
        res, outs = eng.compress_batch(srcs, caps)
        for it, (s, cap, r, o) in enumerate(zip(srcs, caps, res, outs)):
            a, da = ref.compress_fast(s, cap)
            tot += 1
            if int(r) != a or not np.array_equal(o, da):
                bad += 1; print("GPU MISMATCH", it, it % 8, s.size, cap, a, int(r))
    # capacities around the size the block really takes: liblz4's limitedOutput tests are conservative, reproduce them exactly
    sizes = [ref.compress_fast(s, orc.bound(s.size))[0] for s in srcs]
    for delta in (-9, -1, 0, 1, 4, 5, 6, 12, 13):
        caps = [max(c + delta, 1) for c in sizes]
        res, outs = eng.compress_batch(srcs, caps)
        for it, (s, cap, r, o) in enumerate(zip(srcs, caps, res, outs)):
            a, da = ref.compress_fast(s, cap)
            tot += 1
            if int(r) != a or not np.array_equal(o, da):
                bad += 1; print("GPU MISMATCH (tight cap)", it, it % 8, s.size, cap, delta, a, int(r))
    eng.close()
    print("total", tot, "bad", bad)
    return bad


def main_gpu_body(iters, seed):
    """The staged kernels on device-resident data, both builds of the parser: the fuzz inputs back to back, cut into blocks of three
    sizes; plz4hip_dev_encode_body (k_l1_parse) and plz4hip_dev_duplex_body (k_l1_duplex, which also decodes the body the call before
    it wrote) against the oracle's records."""
    import torch
    from plz4_amd._native import Engine
    orc, eng = Oracle(), Engine(0)
    rng = np.random.default_rng(seed)
    data = np.concatenate([make(rng, it) for it in range(iters)])
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    bad = tot = 0
    for bsz in (65536, 1 << 20, 4 << 20):
        want = [orc.block_record(data[o:o + bsz], bsz, True) for o in range(0, data.size, bsz)]
        body = np.concatenate(want); nb = len(want)
        d_b = [torch.zeros(body.size + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
        d_off = [torch.zeros(nb + 1, dtype=torch.int64, device=dev) for _ in range(2)]
        d_len = torch.zeros(nb, dtype=torch.int32, device=dev)
        d_out = torch.zeros(nb * bsz, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(nb, dtype=torch.int32, device=dev); d_st = torch.full((nb,), -9, dtype=torch.int32, device=dev)
        eng.dev_encode_body(d_src.data_ptr(), data.size, bsz, True, d_b[0].data_ptr(), d_b[0].numel(), d_off[0].data_ptr(), d_len.data_ptr(), s, level=1)
        eng.dev_duplex_body(d_src.data_ptr(), data.size, bsz, True, d_b[1].data_ptr(), d_b[1].numel(), d_off[1].data_ptr(), d_len.data_ptr(),
                            d_b[0].data_ptr(), d_off[0].data_ptr(), nb, bsz, True, d_out.data_ptr(), bsz, bsz, d_res.data_ptr(), d_st.data_ptr(), s)
        torch.cuda.synchronize()
        for k in range(2):
            tot += 1
            got = d_b[k][:body.size].cpu().numpy()
            if int(d_off[k][nb].item()) != body.size or not np.array_equal(got, body):
                bad += 1
                offs = d_off[k].cpu().numpy(); o = 0
                for i, w in enumerate(want):
                    if int(offs[i + 1] - offs[i]) != w.size or not np.array_equal(got[o:o + w.size], w):
                        print("GPU BODY MISMATCH", "encode_body" if k == 0 else "duplex_body", "bsz", bsz, "block", i); break
                    o += w.size
        tot += 1
        if int(d_st.abs().sum().item()) != 0 or not np.array_equal(d_out[:data.size].cpu().numpy(), data):
            bad += 1; print("GPU BODY: the duplex call's decode side, bsz", bsz)
    eng.close()
    print("body: blocks of 64 KiB / 1 MiB / 4 MiB over", data.size, "bytes: checks", tot, "bad", bad)
    return bad


def main(iters=120, seed=3):
    emu, ref, orc = Emu(), Ref(), Oracle()
    rng = np.random.default_rng(seed)
    bad = tot = 0
    for it in range(iters):
        src = make(rng, it)
        for desc in (False, True):
            emu.set_descending(desc)
            for cap in (orc.bound(src.size), src.size, max(src.size // 3, 1)):
                a, da = ref.compress_fast(src, cap)
                b, db = emu.compress_fast(src, cap)
                tot += 1
                if a != b or not np.array_equal(da, db):
                    bad += 1; print("MISMATCH", it, it % 8, src.size, cap, desc, a, b)
    emu.set_descending(False)
    print("total", tot, "bad", bad)
    return bad


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[1:] if x != "--gpu"]
    if "--gpu" in sys.argv:
        a2 = args + [120, 3][len(args):]
        sys.exit(1 if (main_gpu(*a2) + main_gpu_body(*a2)) else 0)
    sys.exit(1 if main(*args) else 0)
