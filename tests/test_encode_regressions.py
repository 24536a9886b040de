"""Inputs that once broke the level-1 encoder, kept as regression tests (found by tests/fuzz/fuzz_encode.py against the real
LZ4_compress_fast).  CPU: the lane-emulated device code; -m gpu: the kernel through the C ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
import fuzz_encode  # noqa: E402


def _case_long_match_to_block_end_after_twin_repair():
    """A match that runs to the end of the block (lz4.c:1233 ends the parse there) in a batch whose walk is redone after a
    twin repair: the redo must still end the block (it once re-tested past the last probe position and emitted a match
    inside the last 5 bytes)."""
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "enc_case_block_end_after_repair.npz"))["src"]


def test_emu_long_match_to_block_end_after_twin_repair(ref, orc):
    from emulib import Emu
    emu = Emu()
    src = _case_long_match_to_block_end_after_twin_repair()
    for desc in (False, True):
        emu.set_descending(desc)
        for cap in (orc.bound(src.size), src.size):
            a, da = ref.compress_fast(src, cap)
            b, db = emu.compress_fast(src, cap)
            assert a == b and np.array_equal(da, db), (src.size, cap, desc, a, b)
    emu.set_descending(False)


@pytest.mark.gpu
def test_gpu_long_match_to_block_end_after_twin_repair(orc):
    from plz4_amd._native import Engine
    eng = Engine(0)
    src = _case_long_match_to_block_end_after_twin_repair()
    caps = [orc.bound(src.size), src.size]
    res, outs = eng.compress_batch([src, src], caps)
    for cap, r, o in zip(caps, res, outs):
        want_n, want = orc.compress_fast(src, cap)
        assert int(r) == want_n and np.array_equal(o, want[:want_n]), (cap, int(r), want_n)
    n, out = orc.decompress_safe(np.ascontiguousarray(outs[0]), src.size)
    assert n == src.size and np.array_equal(out[:n], src)
    eng.close()
