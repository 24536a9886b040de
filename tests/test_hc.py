"""HC levels 2..12: lz4mid (2), hash chain (3..9), optimal parser (10..12; BASELINE config 4 = level 12).  The restatement is the device source itself
(plz4_amd/csrc/lz4hc_device.inl): on CPU it is compiled by the emulation harness and checked against the REAL reference
(oracle/_ref, LZ4_compress_HC) and the committed digests; on the GPU (-m gpu) the same checks run through the C ABI."""
import hashlib
import json
import os

import numpy as np
import pytest

import corpus
from plz4_amd import synth

LEVELS = tuple(range(2, 13))
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases():
    cases = [("T", synth.text(300000)), ("Z", np.zeros(100000, np.uint8)), ("M", synth.make("M", 400000, 65536))]
    cases += corpus.twin_cases()[:8]
    cases += [c for c in corpus.small_cases() if c[1].size in (0, 5, 12, 13, 14, 40, 300, 4097)]
    cases += [("S%d" % s, corpus.structured(120000, s)) for s in range(6)]
    return cases


def _golden_blocks():
    g = json.load(open(os.path.join(G, "hc_digests.json")))
    cache = {}
    for b in g["blocks"]:
        key = (b["kind"], b["bsz"])
        if key not in cache:
            n = max(x["index"] for x in g["blocks"] if (x["kind"], x["bsz"]) == key) + 1
            cache[key] = synth.make(b["kind"], n * b["bsz"], b["bsz"])
        blk = cache[key][b["index"] * b["bsz"]:(b["index"] + 1) * b["bsz"]]
        if sha(blk) != b["src_sha"]:
            pytest.skip("synthetic generator differs on this numpy build")
        yield b, blk


@pytest.fixture(scope="module")
def emu():
    from emulib import Emu
    return Emu()


def test_emu_hc_vs_reference(ref, orc, emu):
    for name, src in _cases():
        for lvl in LEVELS:
            for cap in (orc.bound(src.size), src.size, max(src.size // 3, 1)):
                a, da = ref.compress_hc(src, cap, lvl)
                b, db = emu.compress_hc(src, cap, lvl)
                assert a == b and np.array_equal(da, db), (name, src.size, lvl, cap, a, b)


def test_emu_hc_golden_digests(emu, orc):
    for b, blk in _golden_blocks():
        r, c = emu.compress_hc(blk, b["bsz"], b["level"])
        assert r == b["ret"] and (not r or sha(c) == b["comp_sha"]), (b["kind"], b["bsz"], b["index"], b["level"])
        if r and b["index"] == 0:
            n, out = orc.decompress_safe(np.ascontiguousarray(c), b["bsz"] + 8)     # and it is a valid LZ4 block
            assert n == blk.size and np.array_equal(out, blk)


@pytest.mark.gpu
def test_gpu_hc_vs_reference(ref, orc):
    from plz4_amd._native import Engine
    eng = Engine(0)
    cases = [c for c in _cases() if c[1].size <= 150000]
    for lvl in LEVELS:
        srcs = [s for _, s in cases]
        for capf in (lambda n: orc.bound(n), lambda n: n):
            caps = [capf(s.size) for s in srcs]
            res, outs = eng.compress_batch(srcs, caps, level=lvl)
            for (name, s), cap, r, o in zip(cases, caps, res, outs):
                a, da = ref.compress_hc(s, cap, lvl)
                assert int(r) == a and np.array_equal(o, da), (name, s.size, lvl, cap)
    eng.close()


@pytest.mark.gpu
def test_gpu_hc_lazy_levels_in_segments(ref, orc, monkeypatch):
    """Levels 3..11 walk a block in segments at once and stitch the walks (lz4hc_lazy_device.inl).  Small segments in small
    blocks: many stitches per block, walks that meet late or never (the pattern blocks); and the old one-wave path switched
    on instead (PLZ4HIP_HC_LAZY_OFF) still gives the same bytes."""
    from plz4_amd._native import Engine
    cases = [c for c in _cases() if c[1].size <= 150000] + [c for c in _lazy_cases() if c[0].startswith("P")]
    srcs = [s for _, s in cases]
    want = {lvl: [ref.compress_hc(s, orc.bound(s.size), lvl) for s in srcs] for lvl in (3, 4, 6, 9, 10, 11)}
    for env in ({"PLZ4HIP_HC_MIN_SEG": "2048", "PLZ4HIP_HC_SEGS": "16"}, {"PLZ4HIP_HC_MIN_SEG": "20000", "PLZ4HIP_HC_SEGS": "3"},
                {"PLZ4HIP_HC_LAZY_OFF": "1"}):
        for k in ("PLZ4HIP_HC_MIN_SEG", "PLZ4HIP_HC_SEGS", "PLZ4HIP_HC_LAZY_OFF"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(0)
        for lvl in (3, 4, 6, 9, 10, 11):
            res, outs = eng.compress_batch(srcs, [orc.bound(s.size) for s in srcs], level=lvl)
            for (name, s), r, o, (a, da) in zip(cases, res, outs, want[lvl]):
                assert int(r) == a and np.array_equal(o, da), (env, name, s.size, lvl)
        eng.close()


@pytest.mark.gpu
def test_gpu_hc_golden_and_frame(orc):
    from plz4_amd import host
    from plz4_amd._native import Engine
    eng = Engine(0)
    items = [(b, blk) for b, blk in _golden_blocks() if b["bsz"] <= (256 << 10)]      # 4 MiB: test_gpu_hc_config4_full_size
    for lvl in LEVELS:
        for bsz in sorted({b["bsz"] for b, _ in items}):
            sel = [(b, blk) for b, blk in items if b["bsz"] == bsz and b["level"] == lvl]
            recs = eng.encode_records([blk for _, blk in sel], bsz, True, level=lvl)
            for (b, blk), rec in zip(sel, recs):
                payload = rec[4:-4]
                assert (payload.size if not (rec[3] & 0x80) else 0) == b["ret"] or b["ret"] == 0
                if b["ret"]:
                    assert sha(payload) == b["comp_sha"], (b["kind"], bsz, b["index"], lvl)
    eng.close()
    e = host.hip_engine(0)                                                 # WithLevel(12) through the host layer
    payload = synth.text(3 * (64 << 10) + 500).tobytes()
    w = host.Writer(e, parallel=1, level=12, block_size=host.BlockIdx64KB, block_checksum=True)
    assert w.write(payload)[1] == 0 and not w.close()
    f12 = w.output()
    w = host.Writer(e, parallel=1, level=1, block_size=host.BlockIdx64KB, block_checksum=True)
    w.write(payload); w.close()
    assert len(f12) < len(w.output())
    n, out, err = host.Reader(e, f12).write_to()
    assert not err and out == payload
    e.close()


@pytest.mark.gpu
def test_gpu_hc_config4_full_size(ref, orc):
    """BASELINE config 4 at its real size: 4 MiB blocks through the HC kernels.  Every committed 4 MiB digest (T blocks,
    levels 2, 5, 9, 12 -- generated from the real LZ4_compress_HC by tests/golden/make_golden.py) and, at level 12, an M
    block (text / random / zeros mix) against the reference run here; records decode back to the plaintext."""
    from plz4_amd._native import Engine
    eng = Engine(0)
    bsz = 4 << 20
    items = [(b, blk) for b, blk in _golden_blocks() if b["bsz"] == bsz]
    assert {b["level"] for b, _ in items} >= {2, 5, 9, 12}
    mblk = synth.make("M", bsz, bsz)
    for lvl in sorted({b["level"] for b, _ in items}):
        sel = [(b, blk) for b, blk in items if b["level"] == lvl]
        srcs = [blk for _, blk in sel] + ([mblk] if lvl == 12 else [])
        recs = eng.encode_records(srcs, bsz, True, level=lvl)
        for (b, blk), rec in zip(sel, recs):
            assert b["ret"] and not (rec[3] & 0x80) and rec.size - 8 == b["ret"], (lvl, b["index"], rec.size, b["ret"])
            assert sha(rec[4:-4]) == b["comp_sha"], (b["kind"], b["index"], lvl)
        if lvl == 12:
            n, want = ref.compress_hc(mblk, bsz, 12)
            assert n and recs[-1].size - 8 == n and np.array_equal(recs[-1][4:-4], want[:n])
        res, st, outs = eng.decode_records([np.ascontiguousarray(r) for r in recs], bsz, True)
        for s, r, k, o in zip(srcs, res, st, outs):
            assert int(k) == 0 and int(r) == s.size and np.array_equal(o, s)
    # the levels without a committed 4 MiB digest: a T and the M block against the reference run here (level 3 on the chain
    # alone, the others on the per-hash lists, 10 and 11 with the chain swap)
    tblk = synth.make("T", bsz, bsz)
    for lvl in (3, 4, 6, 7, 8, 10, 11):
        recs = eng.encode_records([tblk, mblk], bsz, True, level=lvl)
        for blk, rec in zip((tblk, mblk), recs):
            n, want = ref.compress_hc(blk, bsz, lvl)
            assert n and rec.size - 8 == n and np.array_equal(rec[4:-4], want[:n]), (lvl, n, rec.size)
    eng.close()


def test_emu_hc_on_the_chain_built_up_front(ref, orc, emu):
    """Levels 3..11 as the kernels run them for independent blocks: the chain of the whole block built first (hc12_build_chain),
    the parsers of lz4hc_device.inl reading it instead of inserting into their own tables == LZ4_compress_HC."""
    cases = [("T", synth.text(90000)), ("Z", np.zeros(9000, np.uint8)), ("M", synth.make("M", 140000, 65536)[60000:])]
    cases += [(n, c[:6000]) for n, c in corpus.twin_cases()[:3]]
    cases += [c for c in corpus.small_cases() if c[1].size in (0, 5, 12, 13, 14, 40, 300, 4097)]
    cases += [("S%d" % s, corpus.structured(30000, s)) for s in range(3)]
    for name, src in cases:
        for lvl in range(3, 12):
            for cap in (orc.bound(src.size), src.size, max(src.size // 3, 1)):
                a, da = ref.compress_hc(src, cap, lvl)
                b, db = emu.compress_hc_pre(src, cap, lvl)
                assert a == b and np.array_equal(da, db), (name, src.size, lvl, cap, a, b)


def test_emu_hc_levels_4_to_11_on_the_lists(ref, orc, emu):
    """Levels 4..11 as the kernels run them when the per-hash lists fit the workspace: hc_find_wider_lists looks at up to 63
    candidates of a chain per round, one per lane, and must pick what the one-at-a-time walk picks == LZ4_compress_HC.
    The P cases are runs of short patterns: they drive the pattern analysis inside the rounds."""
    cases = [("T", synth.text(70000)), ("Z", np.zeros(9000, np.uint8)), ("M", synth.make("M", 140000, 65536)[60000:])]
    cases += [(n, c[:6000]) for n, c in corpus.twin_cases()[:3]]
    cases += [c for c in corpus.small_cases() if c[1].size in (0, 5, 12, 13, 14, 40, 300, 4097)]
    cases += [("S%d" % s, corpus.structured(30000, s)) for s in range(2)]
    rng = np.random.default_rng(5)
    for it in range(6):
        n = int(rng.integers(1000, 30000)); parts = []; have = 0
        while have < n:
            pat = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8)
            parts += [np.tile(pat, int(rng.integers(1, 3000))), rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)]
            have += parts[-1].size + parts[-2].size
        cases.append(("P%d" % it, np.concatenate(parts)[:n].copy()))
    for name, src in cases:
        for lvl in (4, 5, 6, 7, 8, 9, 10, 11):          # 10, 11: the optimal parser, its searches with the chain swap
            for cap in (orc.bound(src.size), src.size, max(src.size // 3, 1)) if lvl in (5, 9, 11) else (orc.bound(src.size),):
                a, da = ref.compress_hc(src, cap, lvl)
                b, db = emu.compress_hc_lists(src, cap, lvl)
                assert a == b and np.array_equal(da, db), (name, src.size, lvl, cap, a, b)


def _lazy_cases():
    cases = [("T", synth.text(70000)), ("Z", np.zeros(9000, np.uint8)), ("M", synth.make("M", 140000, 65536)[60000:])]
    cases += [(n, c[:6000]) for n, c in corpus.twin_cases()[:3]]
    cases += [c for c in corpus.small_cases() if c[1].size in (0, 5, 12, 13, 14, 40, 300, 4097)]
    cases += [("S%d" % s, corpus.structured(30000, s)) for s in range(2)]
    rng = np.random.default_rng(7)
    for it in range(5):
        n = int(rng.integers(1000, 30000)); parts = []; have = 0
        while have < n:
            pat = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8)
            parts += [np.tile(pat, int(rng.integers(1, 3000))), rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)]
            have += parts[-1].size + parts[-2].size
        cases.append(("P%d" % it, np.concatenate(parts)[:n].copy()))
    return cases


def test_emu_hc_lazy_levels_as_the_kernels_run_them(ref, orc, emu):
    """Levels 3..9 for independent blocks (lz4hc_lazy_device.inl): searches one candidate per lane (levels 3..6, first searches
    for 64 / attempts positions at once) or 63 per round (7..9), the deciding walk writing records, the block cut into
    segments that are walked independently and stitched (one; few; many tiny ones: walks that meet late or not at all), the
    emit stage of level 1 without catch-up == LZ4_compress_HC, for every capacity verdict."""
    for name, src in _lazy_cases():
        for lvl in range(3, 12):                          # 10, 11: the optimal parser in segments (walks meet on position AND anchor)
            caps = (orc.bound(src.size), src.size, max(src.size // 3, 1)) if lvl in (3, 5, 9, 11) else (orc.bound(src.size),)
            for cap in caps:
                a, da = ref.compress_hc(src, cap, lvl)
                for segs, mseg in ((1, 65536), (4, 8192), (16, 300)) if cap == caps[0] else ((3, 2000),):
                    b, db = emu.compress_hc_lazy(src, cap, lvl, segs, mseg)
                    assert a == b and np.array_equal(da, db), (name, src.size, lvl, cap, segs, mseg, a, b)


def test_emu_hc_level2_in_batches(ref, orc, emu):
    """Level 2 for independent blocks (hc_mid_parse): the positions of a literal run one per lane on the assumption that the
    run goes on, candidates from the lanes below or the tables, the first matching lane ends the batch == LZ4_compress_HC(2).
    The pattern blocks repeat hashes inside a batch; the long random block takes the steps above 1 (:668)."""
    cases = _lazy_cases() + [c for c in corpus.small_cases() if c[1].size <= 5000]
    cases += [("R", np.random.default_rng(3).integers(0, 256, 40000, dtype=np.uint8)), ("T2", synth.text(70000))]
    mixed = np.concatenate([np.random.default_rng(4).integers(0, 256, 3000, dtype=np.uint8), synth.text(5000)] * 6)
    cases.append(("RT", mixed))
    for name, src in cases:
        for cap in (orc.bound(src.size), src.size, max(src.size // 3, 1)) if src.size <= 20000 else (orc.bound(src.size),):
            a, da = ref.compress_hc(src, cap, 2)
            b, db = emu.compress_hc_mid(src, cap)
            assert a == b and np.array_equal(da, db), (name, src.size, cap, a, b)


# ---- level 12 in its three device phases (plz4_amd/csrc/lz4hc12_device.inl): chains + per-hash lists, F(p) per position, parser
def test_emu_hc12_vs_reference(ref, orc, emu):
    """The three phases back to back on the CPU == LZ4_compress_HC(level 12), incl. the parser's own search for positions the
    search phase leaves out (nc) and price-table entries beyond the LDS part (nl); the one-wave parser that writes bytes (blocks
    above 4 MiB) and the parser in segments with records (the frame path)."""
    cases = [("T", synth.text(70000)), ("Z", np.zeros(9000, np.uint8)), ("M", synth.make("M", 140000, 65536)[60000:])]
    cases += [(n, c[:6000 if i == 0 else 2500]) for i, (n, c) in enumerate(corpus.twin_cases()[:2])]     # (the second pattern costs the emulated search seconds per KiB)
    cases += [c for c in corpus.small_cases() if c[1].size in (0, 5, 12, 13, 14, 40, 300, 4097)]
    cases += [("S%d" % s, corpus.structured(30000, s)) for s in range(3)]
    for name, src in cases:
        for k, cap in enumerate((orc.bound(src.size), src.size, max(src.size // 3, 1))):
            a, da = ref.compress_hc(src, cap, 12)
            for nc, nl in (((0, 1024), (7, 40)) if k < 2 else ((0, 1024),)):
                b, db = emu.compress_hc12(src, cap, nc, nl)
                assert a == b and np.array_equal(da, db), (name, src.size, cap, nc, nl, a, b)
                # ... and with the parser walked in segments and stitched (walks meet on position AND anchor), records + emit
                for segs, mseg in ((1, 65536), (16, 300)) if k == 0 else ((4, 4096),):
                    b, db = emu.compress_hc12(src, cap, nc, 256 if nl > 256 else nl, segs, mseg)
                    assert a == b and np.array_equal(da, db), (name, src.size, cap, nc, nl, segs, mseg, a, b)


def test_emu_hc12_search_as_the_kernel_runs_it(emu):
    """Hc12Walk (lists, source window, phases: what k_hc12_search runs per lane) gives the plain chain walk's answer
    (Hc12Lane == LZ4HC_FindLongerMatch) for every position."""
    rng = np.random.default_rng(5)
    cases = [synth.text(80000), np.zeros(9000, np.uint8), synth.make("M", 140000, 65536)[50000:], rng.integers(0, 4, 20000, dtype=np.uint8)]
    for it in range(4):
        parts, have, n = [], 0, int(rng.integers(1000, 25000))
        while have < n:
            pat = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8)
            parts += [np.tile(pat, int(rng.integers(1, 3000))), rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)]
            have += parts[-1].size + parts[-2].size
        cases.append(np.concatenate(parts)[:n].copy())
    for src in cases:
        assert emu.hc12_search_check(src) == 0, src.size


def test_emu_hc12_other_atomic_order(ref, emu):
    """hc12_build_lists takes list slots and previous positions with LDS atomics and expects same-slot lanes to resolve in lane
    order; the emulation can resolve them in DEScending order: the kernel must detect that and put it right (same bytes out)."""
    emu.set_descending(True)
    try:
        for src in (synth.text(40000), np.zeros(5000, np.uint8), synth.make("M", 120000, 65536)[60000:100000], np.tile(np.arange(7, dtype=np.uint8), 3000)):
            a, da = ref.compress_hc(src, src.size, 12)
            b, db = emu.compress_hc12(src, src.size)
            assert a == b and np.array_equal(da, db), src.size
            assert emu.hc12_search_check(src) == 0
    finally:
        emu.set_descending(False)


def test_emu_hc12_golden_digests(emu):
    for b, blk in _golden_blocks():
        if b["level"] == 12 and b["bsz"] <= (64 << 10):
            r, c = emu.compress_hc12(blk, b["bsz"])
            assert r == b["ret"] and (not r or sha(c) == b["comp_sha"]), (b["kind"], b["bsz"], b["index"])


@pytest.mark.gpu
def test_gpu_hc12_groups_and_streams(ref, orc, monkeypatch):
    """Level 12 through the three-phase kernels: a call cut into several workspace groups gives the same records; and two HC
    jobs enqueued on two streams of one ctx (they share the ctx's HC workspaces) do not disturb each other."""
    import torch
    from plz4_amd._native import Engine
    bsz = 256 << 10
    data = synth.make("M", 11 * bsz + 999, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    want = []
    for s in srcs:
        n, c = ref.compress_hc(s, bsz, 12)
        want.append(c[:n] if n else None)
    monkeypatch.setenv("PLZ4HIP_HC12_GROUP", "4")
    eng = Engine(0)
    recs = eng.encode_records(srcs, bsz, True, level=12)
    for s, w, r in zip(srcs, want, recs):
        if w is None:
            assert r[3] & 0x80 and np.array_equal(r[4:-4], s)
        else:
            assert not (r[3] & 0x80) and np.array_equal(r[4:-4], w)
    # two streams, level 12 and level 9, same ctx
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    nblk = len(srcs)
    stride = eng.stage_stride(bsz)
    outs = []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for st, lvl in zip(streams, (12, 9)):
        stage = torch.zeros(nblk * stride, dtype=torch.uint8, device=dev)
        lens = torch.zeros(nblk, dtype=torch.int32, device=dev)
        eng.dev_encode_records(d_src.data_ptr(), data.size, bsz, True, stage.data_ptr(), lens.data_ptr(), st.cuda_stream, level=lvl)
        outs.append((lvl, stage, lens))
    torch.cuda.synchronize()
    for lvl, stage, lens in outs:
        h_stage = stage.cpu().numpy(); h_len = lens.cpu().numpy()
        for i, s in enumerate(srcs):
            rec = h_stage[i * stride:i * stride + int(h_len[i])]
            n, c = ref.compress_hc(s, bsz, lvl)
            if n:
                assert not (rec[3] & 0x80) and np.array_equal(rec[4:-4], c[:n]), (lvl, i)
            else:
                assert rec[3] & 0x80 and np.array_equal(rec[4:-4], s), (lvl, i)
    eng.close()


@pytest.mark.gpu
def test_gpu_hc_chain_and_lists_in_groups(ref, orc, monkeypatch):
    """Levels 3..11 on the chain / lists built up front, the call cut into groups of 3 blocks (what a call beyond the memory set
    aside does): records and raw blocks as the reference's, whichever group a block falls into."""
    from plz4_amd._native import Engine
    bsz = 256 << 10
    data = synth.make("M", 10 * bsz + 4321, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    monkeypatch.setenv("PLZ4HIP_HC_GROUP", "3")
    eng = Engine(0)
    for lvl in (3, 6, 9, 10, 11):
        recs = eng.encode_records(srcs, bsz, True, level=lvl)
        res, outs = eng.compress_batch(srcs, [orc.bound(s.size) for s in srcs], level=lvl)
        for i, (s, rec) in enumerate(zip(srcs, recs)):
            n, c = ref.compress_hc(s, bsz, lvl)
            if n:
                assert not (rec[3] & 0x80) and np.array_equal(rec[4:-4], c[:n]), (lvl, i)
            else:
                assert rec[3] & 0x80 and np.array_equal(rec[4:-4], s), (lvl, i)
            n2, c2 = ref.compress_hc(s, orc.bound(s.size), lvl)
            assert int(res[i]) == n2 and np.array_equal(outs[i], c2), (lvl, i)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nblk", [11, 2])
def test_gpu_hc_builder_beside_the_walk(ref, orc, monkeypatch, nblk):
    """Levels 3..11 of a large call run in four or more groups over the two halves of the workspace, the list builder of the
    next group on a second stream beside the walk of the current one (launch_hc).  The pipeline switched on for a handful of
    blocks (PLZ4HIP_HC_OVERLAP_MIN): records and raw blocks as the reference's, twice in a row on one ctx (the halves and
    the events are reused), device-resident call included."""
    import torch
    from plz4_amd._native import Engine
    bsz = 256 << 10
    data = synth.make("M", (nblk - 1) * bsz + 4321, bsz)
    srcs = [data[o:o + bsz] for o in range(0, data.size, bsz)]
    monkeypatch.setenv("PLZ4HIP_HC_OVERLAP_MIN", "2")
    eng = Engine(0)
    for lvl in (3, 9, 10, 11, 5):
        want = [ref.compress_hc(s, bsz, lvl) for s in srcs]
        for rep in range(2):
            recs = eng.encode_records(srcs, bsz, True, level=lvl)
            for i, (s, rec, (n, c)) in enumerate(zip(srcs, recs, want)):
                if n:
                    assert not (rec[3] & 0x80) and np.array_equal(rec[4:-4], c[:n]), (lvl, rep, i)
                else:
                    assert rec[3] & 0x80 and np.array_equal(rec[4:-4], s), (lvl, rep, i)
        res, outs = eng.compress_batch(srcs, [orc.bound(s.size) for s in srcs], level=lvl)
        for i, s in enumerate(srcs):
            n2, c2 = ref.compress_hc(s, orc.bound(s.size), lvl)
            assert int(res[i]) == n2 and np.array_equal(outs[i], c2), (lvl, i)
    # device-resident, on the caller's stream
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev); stride = eng.stage_stride(bsz)
    d_stage = torch.zeros(len(srcs) * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(len(srcs), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        eng.dev_encode_records(d_src.data_ptr(), data.size, bsz, True, d_stage.data_ptr(), d_len.data_ptr(), st.cuda_stream, level=9)
    st.synchronize()
    lens = d_len.cpu().numpy(); stage = d_stage.cpu().numpy()
    for i, s in enumerate(srcs):
        n, c = ref.compress_hc(s, bsz, 9)
        rec = stage[i * stride:i * stride + int(lens[i])]
        if n:
            assert not (rec[3] & 0x80) and np.array_equal(rec[4:-4], c[:n]), i
        else:
            assert rec[3] & 0x80 and np.array_equal(rec[4:-4], s), i
    eng.close()


@pytest.mark.gpu
def test_gpu_hc12_block_sizes_beyond_the_frame_path(ref, orc):
    """The raw block API takes blocks of any size: 5 MiB runs the three-phase kernels (positions beyond 22 bits), 9 MiB falls back
    to the one-thread-per-block kernel (the three-phase writer packs positions into 23 bits); a batch of ragged sizes too."""
    from plz4_amd._native import Engine
    eng = Engine(0)
    srcs = [synth.text((5 << 20) + 321, seed=5), synth.text(70001, seed=6), synth.text(13, seed=7), np.zeros(0, np.uint8), synth.text(12, seed=8)]
    caps = [orc.bound(s.size) for s in srcs]
    res, outs = eng.compress_batch(srcs, caps, level=12)
    for s, cap, r, o in zip(srcs, caps, res, outs):
        n, want = ref.compress_hc(s, cap, 12)
        assert int(r) == n and np.array_equal(o, want[:n]), s.size
    big = synth.text((9 << 20) + 5, seed=9)
    res, outs = eng.compress_batch([big], [orc.bound(big.size)], level=12)
    n, want = ref.compress_hc(big, orc.bound(big.size), 12)
    assert int(res[0]) == n and np.array_equal(outs[0], want[:n])
    eng.close()
