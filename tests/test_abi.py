"""The C-ABI library loads and exports every symbol include/plz4hip.h declares (no compute without a GPU)."""
import os
import re

from plz4_amd import _native, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    build.build()
    L = _native.load()
    hdr = open(os.path.join(ROOT, "include", "plz4hip.h")).read()
    declared = set(re.findall(r"\b(plz4hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_native.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.plz4hip_abi_version() == _native.ABI_VERSION
    assert L.plz4hip_compress_bound(4 << 20) == (4 << 20) + (4 << 20) // 255 + 16
    assert L.plz4hip_compress_bound(0x7E000001) == 0
    assert L.plz4hip_dev_stage_stride(4 << 20) == (4 << 20) + 16
