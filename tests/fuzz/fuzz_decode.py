"""Fuzz for the decoder's vector path: LZ4 blocks built sequence by sequence (not by a compressor), with literal and match
lengths and offsets drawn around every boundary the vector path cares about (13/14/15 literals, extension bytes 254/255,
18/19/273/274 match bytes, offsets below the match length, sequences straddling the 64-byte window), decoded by the lane-
emulated device code (both builds of the vector path, compared inside tests/emu) and by the oracle.  `--gpu` sends the same
blocks through the C ABI instead.  Not part of the test-suite (minutes); run from the repo root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from orclib import Oracle          # noqa: E402

LL = [0, 0, 1, 2, 3, 5, 7, 12, 13, 14, 15, 16, 17, 30, 45, 47, 48, 49, 62, 63, 64, 100, 254 + 15, 255 + 15, 300, 600]
ML = [4, 4, 5, 6, 8, 12, 17, 18, 19, 20, 33, 64, 100, 272, 273, 274, 275, 528, 529, 1000]


def put_len(out, v):
    while v >= 255:
        out.append(255); v -= 255
    out.append(v)


def make_block(rng, nseq):
    """Returns (compressed bytes, plaintext) of a valid block that ends the way liblz4 requires (last 5 bytes literals,
    last match starts >= 12 bytes before the end)."""
    comp = bytearray(); plain = bytearray()
    for _ in range(nseq):
        ll = int(rng.choice(LL)) if rng.random() < 0.7 else int(rng.integers(0, 40))
        ml = int(rng.choice(ML)) if rng.random() < 0.6 else int(rng.integers(4, 40))
        if not plain and ll == 0:
            ll = 1
        lits = rng.integers(0, 256, ll, dtype=np.uint8).tobytes()
        have = len(plain) + ll
        kind = rng.random()
        if kind < 0.25:
            off = int(rng.integers(1, min(have, 8) + 1))                   # overlapping / run-length
        elif kind < 0.55:
            off = int(rng.integers(1, min(have, 64) + 1))                  # near: inside the current batch
        elif kind < 0.8:
            off = int(rng.integers(1, min(have, 2000) + 1))
        else:
            off = int(rng.integers(1, min(have, 65535) + 1))
        tok = (min(ll, 15) << 4) | min(ml - 4, 15)
        comp.append(tok)
        if ll >= 15:
            put_len(comp, ll - 15)
        comp += lits
        comp += bytes([off & 0xFF, off >> 8])
        if ml - 4 >= 15:
            put_len(comp, ml - 4 - 15)
        plain += lits
        start = len(plain) - off
        for i in range(ml):
            plain.append(plain[start + i])
    tail = int(rng.integers(12, 40))                                      # closing literal run
    lits = rng.integers(0, 256, tail, dtype=np.uint8).tobytes()
    comp.append(min(tail, 15) << 4)
    if tail >= 15:
        put_len(comp, tail - 15)
    comp += lits; plain += lits
    return np.frombuffer(bytes(comp), dtype=np.uint8).copy(), np.frombuffer(bytes(plain), dtype=np.uint8).copy()


def main_dx(iters=60, seed=11, gpu=False):
    orc = Oracle()
    rng = np.random.default_rng(seed)
    blocks = [make_block(rng, int(rng.integers(1500, 9000))) for _ in range(iters)]
    bad = taken = 0
    if gpu:
        from plz4_amd._native import Engine
        eng = Engine(0)
        for lo in range(0, len(blocks), 12):
            part = blocks[lo:lo + 12]
            for spare in (0, 100):
                res, outs = eng.decompress_batch([c for c, _ in part], [p.size + spare for _, p in part])
                for i, ((c, p), r, o) in enumerate(zip(part, res, outs)):
                    if int(r) != p.size or not np.array_equal(o[:p.size], p):
                        bad += 1; print("GPU MISMATCH (dx)", lo + i, c.size, p.size, int(r))
            # damaged copies: the call's results are the oracle's, codes included
            dam = []
            for c, p in part:
                d = c.copy(); i = int(rng.integers(0, d.size)); d[i] ^= 1 << int(rng.integers(0, 8)); dam.append((d, p.size + 8))
            res, outs = eng.decompress_batch([d for d, _ in dam], [cap for _, cap in dam])
            for i, ((d, cap), r, o) in enumerate(zip(dam, res, outs)):
                a, da = orc.decompress_safe(d, cap)
                if int(r) != a or (a >= 0 and not np.array_equal(o, da)):
                    bad += 1; print("GPU MISMATCH (dx, damaged)", lo + i, d.size, int(r), a)
        eng.close()
    else:
        from emulib import Emu
        emu = Emu()
        for i, (c, p) in enumerate(blocks):
            for cap in (p.size, p.size + 100, max(p.size - 1, 0)):
                a, da = orc.decompress_safe(c, cap)
                r, dr, rounds = emu.dx_decode(c, cap)
                if r == -999999:
                    continue
                taken += 1
                if r != a or a < 0 or not np.array_equal(dr, da):
                    bad += 1; print("MISMATCH (dx)", i, c.size, p.size, cap, a, r)
            d = c.copy(); k = int(rng.integers(0, d.size)); d[k] ^= 1 << int(rng.integers(0, 8))
            a, da = orc.decompress_safe(d, p.size + 8)
            r, dr, rounds = emu.dx_decode(d, p.size + 8)
            if r != -999999 and (r != a or a < 0 or not np.array_equal(dr, da)):
                bad += 1; print("MISMATCH (dx, damaged)", i, d.size, a, r)
    print("dx blocks", len(blocks), "answered", taken, "bad", bad)
    return bad


def main(iters=300, seed=11, gpu=False):
    orc = Oracle()
    rng = np.random.default_rng(seed)
    blocks = [make_block(rng, int(rng.integers(1, 400))) for _ in range(iters)]
    bad = 0
    if gpu:
        from plz4_amd._native import Engine
        eng = Engine(0)
        res, outs = eng.decompress_batch([c for c, _ in blocks], [p.size for _, p in blocks])
        for i, ((c, p), r, o) in enumerate(zip(blocks, res, outs)):
            if int(r) != p.size or not np.array_equal(o, p):
                bad += 1; print("GPU MISMATCH", i, c.size, p.size, int(r))
        res, outs = eng.decompress_batch([c for c, _ in blocks], [p.size + 100 for _, p in blocks])   # spare capacity: other exit path
        for i, ((c, p), r, o) in enumerate(zip(blocks, res, outs)):
            if int(r) != p.size or not np.array_equal(o[:p.size], p):
                bad += 1; print("GPU MISMATCH (spare)", i, c.size, p.size, int(r))
        # the record path (k_decode_rec) runs the LDS-staged build of the vector path
        bsz = 1 << 20
        recs = [np.concatenate([np.frombuffer(np.uint32(c.size).tobytes(), dtype=np.uint8), c]) for c, _ in blocks]
        res, st, outs = eng.decode_records(recs, bsz, False)
        for i, ((c, p), r, k, o) in enumerate(zip(blocks, res, st, outs)):
            if int(k) != 0 or int(r) != p.size or not np.array_equal(o, p):
                bad += 1; print("GPU MISMATCH (records)", i, c.size, p.size, int(r), int(k))
        eng.close()
    else:
        from emulib import Emu
        emu = Emu()
        for i, (c, p) in enumerate(blocks):
            for cap in (p.size, p.size + 100, max(p.size - 1, 0)):
                a, da = orc.decompress_safe(c, cap)
                b, db = emu.decompress_safe(c, cap)
                ok = (a == b) and (a <= 0 or np.array_equal(da[:a], db[:a]))
                if cap >= p.size:
                    ok = ok and a == p.size and np.array_equal(db[:a], p)
                if not ok:
                    bad += 1; print("MISMATCH", i, c.size, p.size, cap, a, b)
    print("blocks", len(blocks), "bad", bad)
    return bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a not in ("--gpu", "--dx")]
    fn = main_dx if "--dx" in sys.argv else main
    sys.exit(1 if fn(*(int(x) for x in args), gpu="--gpu" in sys.argv) else 0)
