"""Blocks built sequence by sequence around the boundaries of the decoder's vector path (tests/fuzz/fuzz_decode.py): the
lane-emulated device code on CPU (both builds of the vector path, compared inside tests/emu) and, with -m gpu, the raw and
the record kernels through the C ABI, all against the oracle / the plaintext the generator kept."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
import fuzz_decode  # noqa: E402


def test_emu_decode_built_sequences(orc):
    from emulib import Emu
    emu = Emu()
    rng = np.random.default_rng(2024)
    for i in range(150):
        c, p = fuzz_decode.make_block(rng, int(rng.integers(1, 300)))
        for cap in (p.size, p.size + 100, max(p.size - 1, 0)):
            a, da = orc.decompress_safe(c, cap)
            b, db = emu.decompress_safe(c, cap)
            assert a == b and (a <= 0 or np.array_equal(da[:a], db[:a])), (i, c.size, p.size, cap, a, b)
        assert a is not None


@pytest.mark.gpu
def test_gpu_decode_built_sequences():
    assert fuzz_decode.main(600, 31, gpu=True) == 0
