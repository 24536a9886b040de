"""What plz4 writes for HC levels with a dictionary and/or linked blocks, from the real liblz4 (oracle/_ref) driven the way
clz4.go drives it.  Shared by tests/test_hc_dict.py and tests/golden/make_hc_dict_golden.py.  Test infrastructure."""
import numpy as np

from plz4_amd import synth


def record(orc, ret, comp, src, checksum):
    """blk.CompressToBlk framing of one encoder result (blk/blk.go:69-109)."""
    payload, word = (src, 0x80000000 | src.size) if ret == 0 else (comp, comp.size)
    rec = np.uint32(word).tobytes() + payload.tobytes()
    if checksum:
        rec += np.uint32(orc.xxh32(payload)).tobytes()
    return rec


def ref_records(ref, orc, blocks, bsz, level, linked, dct, checksum=True):
    """Records of a frame: compress.NewCompressorFactory(level, !linked, dict).NewCompressor() (compress/compress.go:32-80)
    applied block by block with capacity bsz (blk.go:73); linked block k gets the last <= 64 KiB of block k-1
    (async/writer.go:412-437).  dct: the user dictionary (any length) or None."""
    keep = daddr = None
    if dct is not None:
        d = dct[-65536:] if dct.size > 65536 else dct                   # compress/dict.go:43-56
        keep, daddr = ref.new_dict_ctx_hc(np.ascontiguousarray(d), level)
    out, rets = [], []
    if linked:
        comp = ref.stream_linked_ctx_hc(level, daddr)
        prev = None
        for b in blocks:
            r, c = comp(b, bsz, None if prev is None else prev[-65536:].copy())
            out.append(record(orc, r, c, b, checksum)); rets.append(r); prev = b
    else:
        if daddr is None:                                               # newIndieCompressorDictHC is only used with a dictionary
            for b in blocks:
                r, c = ref.compress_hc(b, bsz, level)
                out.append(record(orc, r, c, b, checksum)); rets.append(r)
        else:
            comp = ref.stream_ctx_hc(level, daddr)
            for b in blocks:
                r, c = comp(b, bsz)
                out.append(record(orc, r, c, b, checksum)); rets.append(r)
    return out, rets


def golden_inputs():
    """Deterministic inputs of tests/golden/hc_dict_digests.json: a 70000-byte dictionary (truncated to 64 KiB by plz4) and
    two block lists that share vocabulary with it."""
    user = synth.text(70000, seed=42)
    bsz = 64 << 10
    data = synth.make("M", 4 * bsz + 999, bsz, seed=5)
    frame = [np.ascontiguousarray(data[o:o + bsz]) for o in range(0, data.size, bsz)]
    small = synth.text(40000, seed=43)
    indie = [np.ascontiguousarray(small[:n]) for n in (0, 13, 700, 4096, 4097, 30000)] + frame[:2]
    return user, bsz, frame, indie
