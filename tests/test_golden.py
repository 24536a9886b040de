"""Committed golden vectors (tests/golden/*.json, generated from the compiled reference by tests/golden/make_golden.py):
the oracle must reproduce them on CPU, and the HIP path must reproduce them on the GPU (-m gpu)."""
import hashlib
import json
import os

import numpy as np
import pytest

from plz4_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load(name):
    return json.load(open(os.path.join(G, name)))


def _hex(h):
    return np.frombuffer(bytes.fromhex(h), dtype=np.uint8)


def test_oracle_small_vectors(orc):
    g = _load("small_vectors.json")
    for v in g["encode"]:
        src = _hex(v["src"])
        r, c = orc.compress_fast(src, orc.bound(src.size)); assert c.tobytes().hex() == v["bound"], v["name"]
        r, c = orc.compress_fast(src, src.size); assert r == v["cap_n_ret"] and c.tobytes().hex() == v["cap_n"], v["name"]
    for v in g["decode"]:
        r, out = orc.decompress_safe(_hex(v["comp"]), v["cap"])
        assert r == v["ret"]
        if r >= 0:
            assert sha(out) == v["out_sha"]


def _blocks(g):
    cache = {}
    for b in g["blocks"]:
        key = (b["kind"], b["bsz"])
        if key not in cache:
            n = max(x["index"] for x in g["blocks"] if (x["kind"], x["bsz"]) == key) + 1
            cache[key] = synth.make(b["kind"], n * b["bsz"], b["bsz"])
        blk = cache[key][b["index"] * b["bsz"]:(b["index"] + 1) * b["bsz"]]
        if sha(blk) != b["src_sha"]:
            pytest.skip("synthetic generator differs on this numpy build; golden inputs cannot be reproduced")
        yield b, blk


def test_oracle_block_digests(orc):
    for b, blk in _blocks(_load("block_digests.json")):
        r, c = orc.compress_fast(blk, b["bsz"])
        assert r == b["ret"] and (not r or sha(c) == b["comp_sha"]), (b["kind"], b["bsz"], b["index"])


def test_oracle_config1_frame(orc):
    """BASELINE config 1: 16 MiB, 64 KiB blocks, level 1, sync CPU path: frame SHA-256 + per-block sizes, decode back."""
    g = _load("config1_frame.json")
    data = synth.text(16 << 20)
    if sha(data) != g["src_sha"]:
        pytest.skip("synthetic generator differs on this numpy build")
    frame = orc.frame_encode(data, g["block_idx"], g["block_checksum"], g["content_checksum"])
    assert frame.size == g["frame_len"] and sha(frame) == g["frame_sha"]
    n, out = orc.frame_decode(frame, data.size)
    assert n == data.size and sha(out) == g["src_sha"]


@pytest.mark.gpu
def test_gpu_small_vectors():
    from plz4_amd._native import Engine
    eng = Engine(0)
    g = _load("small_vectors.json")
    srcs = [_hex(v["src"]) for v in g["encode"]]
    res, outs = eng.compress_batch(srcs, [Engine.compress_bound(s.size) for s in srcs])
    for v, o in zip(g["encode"], outs):
        assert o.tobytes().hex() == v["bound"], v["name"]
    res, outs = eng.compress_batch(srcs, [s.size for s in srcs])
    for v, r, o in zip(g["encode"], res, outs):
        assert int(r) == v["cap_n_ret"] and o.tobytes().hex() == v["cap_n"], v["name"]
    comps = [_hex(v["comp"]) for v in g["decode"]]
    res, outs = eng.decompress_batch(comps, [v["cap"] for v in g["decode"]])
    for v, r, o in zip(g["decode"], res, outs):
        assert int(r) == v["ret"]
        if v["ret"] >= 0:
            assert sha(o) == v["out_sha"]
    eng.close()


@pytest.mark.gpu
def test_gpu_block_digests_and_config1():
    from plz4_amd import host
    from plz4_amd._native import Engine
    eng = Engine(0)
    items = list(_blocks(_load("block_digests.json")))
    for bsz in sorted({b["bsz"] for b, _ in items}):
        sel = [(b, blk) for b, blk in items if b["bsz"] == bsz]
        res, outs = eng.compress_batch([blk for _, blk in sel], [bsz] * len(sel))
        for (b, _), r, o in zip(sel, res, outs):
            assert int(r) == b["ret"] and (not b["ret"] or sha(o) == b["comp_sha"]), (b["kind"], bsz, b["index"])
    eng.close()
    g = _load("config1_frame.json")
    data = synth.text(16 << 20)
    if sha(data) != g["src_sha"]:
        pytest.skip("synthetic generator differs on this numpy build")
    e = host.hip_engine(0)
    w = host.Writer(e, parallel=0, block_size=g["block_idx"], content_checksum=True)
    assert w.write(data)[1] == 0 and not w.close()
    f = w.output()
    assert len(f) == g["frame_len"] and hashlib.sha256(f).hexdigest() == g["frame_sha"]
    n, out, err = host.Reader(e, f).write_to()
    assert not err and hashlib.sha256(out).hexdigest() == g["src_sha"]
    e.close()
