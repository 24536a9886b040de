"""HC levels 2..12 with a dictionary and/or linked blocks (clz4.StreamCtxHC, clz4.StreamLinkedCtxHC).  The
restatement is the device source (plz4_amd/csrc/lz4hc_device.inl): on CPU it is compiled by the emulation harness and
checked against the REAL liblz4 streams (oracle/_ref, driven as clz4.go drives them) and the committed digests; on the GPU
(-m gpu) the same checks run through the C ABI and the host layer."""
import hashlib
import json
import os

import numpy as np
import pytest

import corpus
import hcdict
from plz4_amd import synth

LEVELS = tuple(range(2, 13))
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def emu():
    from emulib import Emu
    return Emu()


def _emu_records(emu, orc, blocks, bsz, level, linked, dct):
    """The kernels' per-block priming rule (plz4hip.hip hc_dict_of) on the emulation build."""
    d = None if dct is None else np.ascontiguousarray(dct[-65536:])
    out, prev = [], None
    for b in blocks:
        if linked and prev is not None:
            r, c = emu.compress_hc_dict(b, bsz, level, np.ascontiguousarray(prev[-65536:]), 1)
        elif d is not None:
            r, c = emu.compress_hc_dict(b, bsz, level, d, 1 if b.size > 4096 else 2)
        else:
            r, c = emu.compress_hc(b, bsz, level)
        out.append(hcdict.record(orc, r, c, b, True)); prev = b
    return out


def _golden():
    g = json.load(open(os.path.join(G, "hc_dict_digests.json")))
    user, bsz, frame, indie = hcdict.golden_inputs()
    if sha(b"".join(b.tobytes() for b in [user] + frame + indie)) != g["src_sha"]:
        pytest.skip("synthetic generator differs on this numpy build")
    return g, user, bsz, frame, indie


def test_emu_hc_dict_vs_reference_streams(ref, orc, emu):
    user = synth.text(70000, seed=9)
    pat = np.tile(np.frombuffer(b"abcabcab", np.uint8), 9000)            # a periodic run across dictionary, blocks and levels >= 9
    for name, dct, whole in (("text", user, synth.text(150000, seed=10)), ("pattern", pat[:50001], pat[50001:]),
                             ("structured", corpus.structured(30000, 3), corpus.structured(100000, 3)), ("tiny-dict", user[:3], synth.text(50000, seed=11))):
        for bsz in (4096, 40000):
            blocks = [np.ascontiguousarray(whole[o:o + bsz]) for o in range(0, min(whole.size, 3 * bsz + 700), bsz)]
            for lvl in (2, 3, 6, 9, 10, 12):
                for linked, d in ((True, dct), (True, None), (False, dct)):
                    want, _ = hcdict.ref_records(ref, orc, blocks, bsz, lvl, linked, d)
                    got = _emu_records(emu, orc, blocks, bsz, lvl, linked, d)
                    for i, (g_, w) in enumerate(zip(got, want)):
                        assert g_ == w, (name, bsz, lvl, linked, d is not None, i)


def _emu_records_lists(emu, orc, blocks, bsz, level, linked, dct, segs, min_seg):
    """As _emu_records, with the blocks behind an external segment on the path the kernels run since round 4 at levels 3..12: chain
    and lists over segment + block, the level's walk in `segs` segments, stitched, records through the emit stage."""
    d = None if dct is None else np.ascontiguousarray(dct[-65536:])
    out, prev = [], None
    for b in blocks:
        if linked and prev is not None:
            r, c = emu.compress_hc_lazy_ext(b, bsz, level, np.ascontiguousarray(prev[-65536:]), segs, min_seg)
        elif d is not None and b.size > 4096:
            r, c = emu.compress_hc_lazy_ext(b, bsz, level, d, segs, min_seg)
        elif d is not None:
            r, c = emu.compress_hc_dict(b, bsz, level, d, 2)
        else:
            r, c = emu.compress_hc_lazy_ext(b, bsz, level, np.zeros(0, np.uint8), segs, min_seg)
        out.append(hcdict.record(orc, r, c, b, True)); prev = b
    return out


def test_emu_hc_dict_on_the_lists_vs_reference_streams(ref, orc, emu):
    """Levels 3..12 with a dictionary / linked blocks as the kernels run them since round 4 (k_hc_ext_prep -> k_hc12_hist ->
    k_hc12_chain -> k_hc_lazy<true> -> k_hc_stitch<true> -> emit): against the real liblz4 streams, one walk and sixteen segments of
    300 bytes (walks that meet late or never).  tests/fuzz/fuzz_hc_dict.py with PLZ4_FUZZ_LISTS=1 is the long version."""
    user = synth.text(70000, seed=9)
    pat = np.tile(np.frombuffer(b"abcabcab", np.uint8), 9000)
    for name, dct, whole in (("text", user, synth.text(150000, seed=10)), ("pattern", pat[:50001], pat[50001:]),
                             ("structured", corpus.structured(30000, 3), corpus.structured(100000, 3)), ("tiny-dict", user[:3], synth.text(50000, seed=11))):
        for bsz in (4096, 40000):
            blocks = [np.ascontiguousarray(whole[o:o + bsz]) for o in range(0, min(whole.size, 3 * bsz + 700), bsz)]
            for lvl in (3, 5, 9, 10, 12):
                for linked, d in ((True, dct), (True, None), (False, dct)):
                    want, _ = hcdict.ref_records(ref, orc, blocks, bsz, lvl, linked, d)
                    for segs, min_seg in ((1, 65536), (16, 300)):
                        got = _emu_records_lists(emu, orc, blocks, bsz, lvl, linked, d, segs, min_seg)
                        for i, (g_, w) in enumerate(zip(got, want)):
                            assert g_ == w, (name, bsz, lvl, linked, d is not None, segs, i)


def test_emu_hc_dict_golden_digests(orc, emu):
    g, user, bsz, frame, indie = _golden()
    for lvl in (2, 4, 9, 11):                                            # the GPU test covers every level
        e = g["levels"][str(lvl)]
        for name, blocks, linked, d in (("linked_dict", frame, True, user), ("linked", frame, True, None), ("indie_dict", indie, False, user)):
            got = _emu_records(emu, orc, blocks, bsz, lvl, linked, d)
            assert [sha(r) for r in got] == e[name]["sha"], (lvl, name)


@pytest.fixture(scope="module")
def eng():
    from plz4_amd._native import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
def test_gpu_hc_dict_golden_digests(eng):
    g, user, bsz, frame, indie = _golden()
    d = eng.dict_create(np.ascontiguousarray(user))
    for lvl in LEVELS:
        e = g["levels"][str(lvl)]
        got = eng.encode_records_ex(frame[:2], bsz, True, linked=True, d=d, level=lvl)          # the frame in two calls
        got += eng.encode_records_ex(frame[2:], bsz, True, linked=True, d=d, prev_tail=frame[1][-65536:].copy(), level=lvl)
        assert [sha(r.tobytes()) for r in got] == e["linked_dict"]["sha"], lvl
        got = eng.encode_records_ex(frame, bsz, True, linked=True, level=lvl)
        assert [sha(r.tobytes()) for r in got] == e["linked"]["sha"], lvl
        got = eng.encode_records_ex(indie, bsz, True, linked=False, d=d, level=lvl)
        assert [sha(r.tobytes()) for r in got] == e["indie_dict"]["sha"], lvl
    eng.dict_destroy(d)


@pytest.mark.gpu
def test_gpu_hc_dict_vs_reference_streams(ref, orc, eng):
    """Block API with WithBlockDictionary at every level (both sides of the 4 KiB switch, three capacities), and linked
    frames over periodic data, against the real liblz4 streams."""
    user = synth.text(70000, seed=99)
    data = synth.text(120000, seed=7)
    sizes = (0, 5, 13, 100, 4095, 4096, 4097, 65536, 100000)
    srcs = [np.ascontiguousarray(data[:n]) for n in sizes]
    for dct_user in (user, user[:30000], user[:3]):
        d = eng.dict_create(np.ascontiguousarray(dct_user))
        dd = np.ascontiguousarray(dct_user[-65536:])
        for lvl in LEVELS:
            keep, daddr = ref.new_dict_ctx_hc(dd, lvl)
            comp = ref.stream_ctx_hc(lvl, daddr)
            for caps in ([orc.bound(n) for n in sizes], [max(n, 1) for n in sizes], [max(n // 3, 1) for n in sizes]):
                res, outs = eng.compress_batch_dict(srcs, caps, d, level=lvl)
                for s, c, r, o in zip(srcs, caps, res, outs):
                    a, da = comp(s, c)
                    assert int(r) == a and np.array_equal(o, da), (dct_user.size, lvl, s.size, c)
        eng.dict_destroy(d)
    pat = np.tile(np.frombuffer(b"abcabcab", np.uint8), 30000)
    bsz = 64 << 10
    blocks = [np.ascontiguousarray(pat[o:o + bsz]) for o in range(0, pat.size, bsz)]
    for lvl in (2, 9, 12):
        want, _ = hcdict.ref_records(ref, orc, blocks, bsz, lvl, True, None)
        got = eng.encode_records_ex(blocks, bsz, True, linked=True, level=lvl)
        assert [g_.tobytes() for g_ in got] == want, lvl


@pytest.mark.gpu
def test_gpu_hc_linked_full_size_blocks_on_the_lists(ref, orc, monkeypatch):
    """config 5 at the HC levels at its real size: 4 MiB linked blocks behind a 64 KiB dictionary (T and M data), levels 3, 9 and
    12, against the real liblz4 streams -- on the list path (default) and, level 3, on the one-thread parsers it replaces
    (PLZ4HIP_HC_EXT_OFF): the same bytes."""
    from plz4_amd._native import Engine
    bsz = 4 << 20
    user = synth.text(65536, seed=99)
    data = np.concatenate([synth.make("T", 2 * bsz, bsz), synth.make("M", 2 * bsz + 12345, bsz, seed=7)])
    blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
    for lvl, off in ((3, False), (3, True), (9, False), (12, False)):
        if off: monkeypatch.setenv("PLZ4HIP_HC_EXT_OFF", "1")
        else: monkeypatch.delenv("PLZ4HIP_HC_EXT_OFF", raising=False)
        want, _ = hcdict.ref_records(ref, orc, blocks, bsz, lvl, True, user)
        e = Engine(0)
        d = e.dict_create(np.ascontiguousarray(user))
        got = e.encode_records_ex(blocks, bsz, True, linked=True, d=d, level=lvl)
        assert [g_.tobytes() for g_ in got] == want, (lvl, off)
        e.dict_destroy(d); e.close()


@pytest.mark.gpu
def test_gpu_hc_dict_host_layer_round_trip():
    """WithLevel(9) + WithBlockLinked + WithDictionary through the host layer; the frame decodes with the same dictionary.
    And the reference's quirk: a linked frame above level 1 has no stored-block fallback (compress/linked.go:47-49)."""
    from plz4_amd import host
    e = host.hip_engine(0)
    user = synth.text(70000, seed=5).tobytes()
    payload = synth.text(5 * (64 << 10) + 321, seed=6).tobytes()
    for linked in (True, False):
        w = host.Writer(e, parallel=2, level=9, block_size=host.BlockIdx64KB, block_checksum=True, block_linked=linked, dictionary=user)
        assert w.write(payload)[1] == 0 and not w.close()
        frame = w.output()
        w1 = host.Writer(e, parallel=2, level=1, block_size=host.BlockIdx64KB, block_checksum=True, block_linked=linked, dictionary=user)
        w1.write(payload); w1.close()
        assert len(frame) < len(w1.output())
        n, out, err = host.Reader(e, frame, dictionary=user).write_to()
        assert not err and out == payload
    noise = synth.random_bytes(3 * (64 << 10), seed=8).tobytes()
    w = host.Writer(e, parallel=2, level=9, block_size=host.BlockIdx64KB, block_linked=True)
    w.write(noise)
    assert w.close() == host.ErrCompress
    w = host.Writer(e, parallel=2, level=9, block_size=host.BlockIdx64KB, block_linked=False)     # independent blocks: stored, no error
    w.write(noise)
    assert not w.close()
    e.close()
