"""BASELINE config 5 on the device, through the C ABI: level-1 blocks with a 64 KiB dictionary and/or linked blocks, against
the oracle's stream emulation of clz4.StreamIndieCtx / StreamLinkedCtx / DecompressSafeWithDict (pinned to the real liblz4
in test_oracle_vs_ref.py::test_stream_linked_and_dict)."""
import numpy as np
import pytest

from plz4_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from plz4_amd._native import Engine
    e = Engine(0)
    yield e
    e.close()


def _record(orc, comp_ret, comp, src, checksum):
    """blk.CompressToBlk framing of one encoder result (blk.go:78-109)."""
    if comp_ret == 0:
        payload, word = src, 0x80000000 | src.size
    else:
        payload, word = comp, comp.size
    rec = np.uint32(word).tobytes() + payload.tobytes()
    if checksum:
        rec += np.uint32(orc.xxh32(payload)).tobytes()
    return rec


def test_gpu_block_api_with_dictionary(orc, eng):
    """CompressBlock / DecompressBlock + WithBlockDictionary (plz4_block.go:48-53): both sides of the 4 KiB switch,
    a short dictionary (checkOffset on), a < 8 byte dictionary (dropped), capacity too small."""
    user = synth.text(70000, seed=99)
    data = synth.text(300000, seed=7)
    for dct_user in (user, user[:30000], user[:5]):
        dctx = orc.dict_ctx(dct_user)
        d = eng.dict_create(np.ascontiguousarray(dct_user))
        sizes = (0, 5, 13, 100, 4095, 4096, 4097, 65536, 200000)
        srcs = [np.ascontiguousarray(data[:n]) for n in sizes]
        for caps in ([orc.bound(n) for n in sizes], [max(n, 1) for n in sizes], [max(n // 3, 1) for n in sizes]):
            res, outs = eng.compress_batch_dict(srcs, caps, d)
            for s, c, r, o in zip(srcs, caps, res, outs):
                a, da = orc.compress_indie_dict(s, c, dctx)
                assert int(r) == a and np.array_equal(o, da), (dct_user.size, s.size, c)
        comps = [np.ascontiguousarray(orc.compress_indie_dict(s, orc.bound(s.size), dctx)[1]) for s in srcs]
        dd = np.ascontiguousarray(dct_user[-65536:])
        for caps in ([s.size + 8 for s in srcs], [s.size for s in srcs], [max(s.size - 1, 0) for s in srcs]):
            res, outs = eng.decompress_batch_dict(comps, caps, d)
            for cp, cap, r, o in zip(comps, caps, res, outs):
                a, da = orc.decompress_safe_dict(cp, cap, dd)
                assert int(r) == a, (dct_user.size, cp.size, cap, int(r), a)
                if a >= 0:
                    assert np.array_equal(o, da)
        eng.dict_destroy(d)


@pytest.mark.parametrize("bsz", [64 << 10, 256 << 10])
@pytest.mark.parametrize("with_dict", [False, True])
def test_gpu_linked_records(orc, eng, bsz, with_dict):
    """WithBlockLinked (+ WithDictionary): records == oracle stream emulation, batch split in two calls (prevTail carried)."""
    user = synth.text(70000, seed=42)
    data = synth.make("M", 7 * bsz + 999, bsz, seed=5)
    blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
    dctx = orc.dict_ctx(user) if with_dict else None
    d = eng.dict_create(np.ascontiguousarray(user)) if with_dict else None
    want = []
    prev = None
    for b in blocks:
        tail = None if prev is None else prev[-65536:].copy()
        r, c = orc.compress_linked(b, bsz, tail, dctx if prev is None else None)
        want.append(_record(orc, r, c, b, True)); prev = b
    got = eng.encode_records_ex(blocks[:3], bsz, True, linked=True, d=d)
    got += eng.encode_records_ex(blocks[3:], bsz, True, linked=True, d=d, prev_tail=blocks[2][-65536:].copy())
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.tobytes() == w, i
    # decode the chain in two calls, window carried; starts as the dictionary's last 64 KiB (compress/dict.go:43-56)
    window = np.zeros(65536, dtype=np.uint8); wl = 0
    if with_dict:
        wl = min(user.size, 65536); window[:wl] = user[-wl:]
    recs = [np.ascontiguousarray(g) for g in got]
    res1, st1, out1, wl = eng.decode_records_ex(recs[:4], bsz, True, linked=True, window=window, window_len=wl)
    res2, st2, out2, wl = eng.decode_records_ex(recs[4:], bsz, True, linked=True, window=window, window_len=wl)
    outs = out1 + out2
    assert not any(st1) and not any(st2)
    for b, o in zip(blocks, outs):
        assert np.array_equal(b, o)
    if d is not None:
        eng.dict_destroy(d)


def test_gpu_linked_decode_follows_reference_window_rule(orc, eng):
    """The reference reader does not feed stored blocks into the window (sync/reader.go:75-78); a later block that refers
    to a stored block's bytes then decodes against the OLD window.  Reproduced, not fixed."""
    bsz = 64 << 10
    a = synth.text(bsz, seed=1); r_ = synth.random_bytes(bsz, seed=2); b = np.concatenate([r_[-30000:], synth.text(bsz - 30000, seed=3)])
    blocks = [a, r_, b]
    recs = eng.encode_records_ex(blocks, bsz, False, linked=True)
    assert recs[1][3] & 0x80                                            # the random block is stored
    window = np.zeros(65536, dtype=np.uint8)
    res, st, outs, wl = eng.decode_records_ex([np.ascontiguousarray(x) for x in recs], bsz, False, linked=True, window=window, window_len=0)
    # emulate the reference reader: window after block 0 = last 64 KiB of a; stored block skipped; block 2 against that window
    win = a[-65536:].copy()
    payload2 = np.ascontiguousarray(recs[2][4:])
    want_n, want = orc.decompress_safe_dict(payload2, bsz + 8, win)
    assert int(res[0]) == bsz and int(res[1]) == bsz and int(res[2]) == want_n
    if want_n >= 0:
        assert np.array_equal(outs[2], want)
        assert not np.array_equal(outs[2], b)                           # i.e. the reference's own reader garbles this frame


def test_gpu_indie_frame_with_dictionary(orc, eng):
    """Independent blocks + WithDictionary: every block against the same dictionary context."""
    user = synth.text(65536, seed=11)
    bsz = 64 << 10
    data = synth.text(5 * bsz + 3000, seed=12)
    blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)] + [np.ascontiguousarray(data[:3000])]
    dctx = orc.dict_ctx(user); d = eng.dict_create(user)
    got = eng.encode_records_ex(blocks, bsz, True, linked=False, d=d)
    for b, g in zip(blocks, got):
        r, c = orc.compress_indie_dict(b, bsz, dctx)
        assert g.tobytes() == _record(orc, r, c, b, True)
    res, st, outs, _ = eng.decode_records_ex([np.ascontiguousarray(g) for g in got], bsz, True, linked=False, d=d)
    assert not any(st)
    for b, o in zip(blocks, outs):
        assert np.array_equal(b, o)
    eng.dict_destroy(d)


@pytest.mark.parametrize("level", [1, 3])
def test_gpu_linked_encode_in_many_chunks(orc, level, monkeypatch):
    """A linked encode is cut into staging chunks like any other host call (the first block of a later chunk is primed with
    the tail of the block before it, taken from the caller's buffers): the records must not depend on the chunk size."""
    from plz4_amd._native import Engine
    bsz = 64 << 10
    user = synth.text(70000, seed=42)
    data = synth.make("M", 37 * bsz + 555, bsz, seed=15)
    blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
    blocks.insert(9, np.ascontiguousarray(data[:5]))                  # a short flush block: the next block's window is 5 bytes
    e = Engine(0); d = e.dict_create(np.ascontiguousarray(user))
    whole = e.encode_records_ex(blocks, bsz, True, linked=True, d=d, level=level)
    monkeypatch.setenv("PLZ4HIP_HOST_CHUNK_MB", "1")
    parts = e.encode_records_ex(blocks, bsz, True, linked=True, d=d, level=level)
    assert [r.tobytes() for r in parts] == [r.tobytes() for r in whole]
    if level == 1:
        prev = None
        for b, g in zip(blocks, whole):
            tail = None if prev is None else prev[-65536:].copy()
            r, c = orc.compress_linked(b, bsz, tail, orc.dict_ctx(user) if prev is None else None)
            assert g.tobytes() == _record(orc, r, c, b, True); prev = b
    e.dict_destroy(d); e.close()


@pytest.mark.parametrize("kind", ["T", "M"])
def test_gpu_linked_records_full_size_blocks(orc, eng, kind):
    """BASELINE configs[4] at its real size: 4 MiB linked blocks + a 64 KiB dictionary; records == oracle stream emulation,
    and the chain decodes back (the random parts of M are stored blocks, so the decode is checked on T only: the reference's
    reader does not feed stored blocks into its window)."""
    bsz = 4 << 20
    user = synth.text(65536, seed=77)
    data = synth.make(kind, 3 * bsz + 12345, bsz, seed=21)
    blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
    dctx = orc.dict_ctx(user); d = eng.dict_create(np.ascontiguousarray(user))
    got = eng.encode_records_ex(blocks, bsz, True, linked=True, d=d)
    prev = None
    for i, (b, g) in enumerate(zip(blocks, got)):
        tail = None if prev is None else prev[-65536:].copy()
        r, c = orc.compress_linked(b, bsz, tail, dctx if prev is None else None)
        assert g.tobytes() == _record(orc, r, c, b, True), (kind, i); prev = b
    if kind == "T":
        window = np.zeros(65536, dtype=np.uint8); window[:] = user
        res, st, outs, wl = eng.decode_records_ex([np.ascontiguousarray(g) for g in got], bsz, True, linked=True, window=window, window_len=65536)
        assert not any(st)
        for b, o in zip(blocks, outs):
            assert np.array_equal(b, o)
    eng.dict_destroy(d)


def test_gpu_linked_decode_of_several_frames_at_once(orc, eng):
    """plz4hip_decode_records_chains: every linked frame is its own chain (one wavefront), chains share nothing.  The result of
    each chain must be what plz4hip_decode_records_ex(linked=1) gives for it alone -- window state included -- and a corrupt
    block stops its own chain only."""
    bsz = 64 << 10
    user = synth.text(70000, seed=3)
    frames, windows, wlens = [], [], []
    for f, (nblk, with_dict) in enumerate([(5, True), (1, False), (0, False), (7, False), (3, True)]):
        data = synth.make("T" if f % 2 else "M", nblk * bsz - (777 if nblk else 0), bsz, seed=30 + f) if nblk else np.zeros(0, np.uint8)
        blocks = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
        d = eng.dict_create(np.ascontiguousarray(user)) if with_dict else None
        recs = [np.ascontiguousarray(r) for r in eng.encode_records_ex(blocks, bsz, True, linked=True, d=d)] if blocks else []
        if d is not None:
            eng.dict_destroy(d)
        w = np.zeros(65536, dtype=np.uint8); wl = 0
        if with_dict:
            wl = 65536; w[:] = user[-65536:]
        frames.append((blocks, recs)); windows.append(w); wlens.append(wl)
    bad = frames[3][1][2].copy(); bad[40] ^= 0x55                      # frame 3, block 2: checksum mismatch
    frames[3] = (frames[3][0], frames[3][1][:2] + [bad] + frames[3][1][3:])
    # each chain alone
    alone = []
    for (blocks, recs), w, wl in zip(frames, windows, wlens):
        if not recs:
            alone.append(([], [], [], wl, w.copy())); continue
        wcopy = w.copy()
        res, st, outs, wl2 = eng.decode_records_ex(recs, bsz, True, linked=True, window=wcopy, window_len=wl)
        alone.append((list(res), list(st), outs, wl2, wcopy))
    # all chains in one call
    wall = np.stack(windows).copy()
    got, wl_out = eng.decode_records_chains([recs for _, recs in frames], bsz, True, windows=wall, window_lens=np.array(wlens, dtype=np.int32))
    for k, ((res, st, outs), (ares, ast, aouts, awl, aw)) in enumerate(zip(got, alone)):
        assert list(res) == ares and list(st) == ast, k
        for o, ao in zip(outs, aouts):
            assert np.array_equal(o, ao), k
        assert int(wl_out[k]) == awl and np.array_equal(wall[k][:awl], aw[:awl]), k
    assert any(int(x) != 0 for x in got[3][1]) and not any(int(x) for x in got[0][1]) and not any(int(x) for x in got[4][1])
    for k in (0, 1, 4):                                                # the frames without stored / bad blocks decode to their plaintext
        if not any(r[3] & 0x80 for r in frames[k][1]):
            for b, o in zip(frames[k][0], got[k][2]):
                assert np.array_equal(b, o), k
