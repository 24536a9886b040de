#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference (liblz4 v1.10.0 as vendored by plz4, compiled from
/root/reference by `make -C oracle ref`).  Run in the build container only:

    python tests/golden/make_golden.py

Outputs (data only -- inputs by generator seed or hex, expected bytes or their SHA-256):
    tests/golden/small_vectors.json   small inputs: compressed bytes at cap=bound and cap=n, decode results incl. error codes
    tests/golden/block_digests.json   4 MiB / 64 KiB blocks of the synthetic corpora: sizes + SHA-256 of the compressed bytes
    tests/golden/config1_frame.json   BASELINE config 1: 16 MiB T, 64 KiB blocks, sync frame: per-block sizes + frame SHA-256
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import corpus  # noqa: E402
from orclib import Oracle, Ref  # noqa: E402
from plz4_amd import synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref, orc = Ref(), Oracle()
    # ---- small vectors
    small = []
    cases = [c for c in corpus.small_cases() if c[1].size <= 300][::3] + [("hello", np.frombuffer(b"hello", dtype=np.uint8))]
    for name, src in cases:
        n = src.size
        rb, cb = ref.compress_fast(src, ref.L.LZ4_compressBound(n))
        rn, cn = ref.compress_fast(src, n)
        small.append({"name": name, "src": src.tobytes().hex(), "bound": cb.tobytes().hex(), "cap_n_ret": rn,
                      "cap_n": cn.tobytes().hex()})
    rng = np.random.default_rng(2024)
    dec = []
    for seed in range(12):
        n = int(rng.integers(40, 400))
        src = corpus.structured(n, 9000 + seed)
        _, comp = ref.compress_fast(src, ref.L.LZ4_compressBound(n))
        variants = [comp.copy()]
        for t in range(6):
            bad = comp.copy(); k = t % 3
            if k == 0: bad = bad[:int(rng.integers(1, bad.size))]
            elif k == 1: bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
            else: bad[int(rng.integers(0, bad.size))] = 0xFF
            variants.append(bad)
        for v in variants:
            for cap in (n, n + 8):
                r, out = ref.decompress_safe(np.ascontiguousarray(v), cap)
                dec.append({"comp": v.tobytes().hex(), "cap": cap, "ret": r, "out_sha": sha(out) if r >= 0 else None})
    json.dump({"reference": "liblz4 1.10.0 (plz4 internal/pkg/clz4)", "encode": small, "decode": dec},
              open(os.path.join(HERE, "small_vectors.json"), "w"), indent=0)

    # ---- block digests
    blocks = []
    for kind, bsz, nblk in (("T", 4 << 20, 4), ("M", 4 << 20, 4), ("Z", 4 << 20, 1), ("R", 4 << 20, 1), ("T", 64 << 10, 8), ("M", 64 << 10, 8)):
        data = synth.make(kind, nblk * bsz, bsz)
        for i in range(nblk):
            blk = data[i * bsz:(i + 1) * bsz]
            r, c = ref.compress_fast(blk, bsz)                       # frame path: cap == bsz
            blocks.append({"kind": kind, "bsz": bsz, "index": i, "src_sha": sha(blk), "ret": r, "comp_sha": sha(c) if r else None})
    json.dump({"reference": "liblz4 1.10.0", "blocks": blocks}, open(os.path.join(HERE, "block_digests.json"), "w"), indent=0)

    # ---- HC levels 2..12 (config 4 = level 12): LZ4_compress_HC of the real reference
    hc = []
    ALL = tuple(range(2, 13))
    for kind, bsz, nblk, levels in (("T", 64 << 10, 4, ALL), ("M", 256 << 10, 3, ALL), ("T", 4 << 20, 1, (2, 5, 9, 12))):
        data = synth.make(kind, nblk * bsz, bsz)
        for i in range(nblk):
            blk = data[i * bsz:(i + 1) * bsz]
            for lvl in levels:
                r, c = ref.compress_hc(blk, bsz, lvl)
                hc.append({"kind": kind, "bsz": bsz, "index": i, "level": lvl, "src_sha": sha(blk), "ret": r, "comp_sha": sha(c) if r else None})
    json.dump({"reference": "liblz4 1.10.0 LZ4_compress_HC", "blocks": hc}, open(os.path.join(HERE, "hc_digests.json"), "w"), indent=0)

    # ---- config 1 frame (sync writer semantics; per-block arithmetic checked against the reference here)
    data = synth.text(16 << 20)
    bsz = 64 << 10
    frame = orc.frame_encode(data, 4, block_checksum=False, content_checksum=True)
    sizes = []
    off = 7
    for i in range(data.size // bsz):
        word = int.from_bytes(frame[off:off + 4].tobytes(), "little"); sz = word & 0x7FFFFFFF
        r, c = ref.compress_fast(data[i * bsz:(i + 1) * bsz], bsz)
        assert r == sz and np.array_equal(c, frame[off + 4:off + 4 + sz]), i
        sizes.append(sz); off += 4 + sz
    json.dump({"input": "synth.text(16 MiB, seed 0x504C5A34)", "src_sha": sha(data), "block_idx": 4, "content_checksum": True,
               "block_checksum": False, "block_sizes": sizes, "frame_len": int(frame.size), "frame_sha": sha(frame)},
              open(os.path.join(HERE, "config1_frame.json"), "w"), indent=0)
    print("golden vectors written")


if __name__ == "__main__":
    main()
