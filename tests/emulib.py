"""ctypes binding for the lane-emulation build of the device code (tests/emu/emu_kernels.cpp).
Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from orclib import ROOT, _ptr, u8p

EMU_SRC = os.path.join(ROOT, "tests", "emu", "emu_kernels.cpp")
EMU_SO = os.path.join(ROOT, "tests", "emu", "_build", "libemu.so")
DEV_SRCS = [os.path.join(ROOT, "plz4_amd", "csrc", f) for f in ("lz4_dx_device.inl", "lz4_device.inl", "lz4_seq_device.inl", "lz4hc_device.inl", "lz4hc12_device.inl", "lz4hc_lazy_device.inl", "wave.h")]


def build_emu():
    newest = max(os.path.getmtime(p) for p in [EMU_SRC] + DEV_SRCS)
    if (not os.path.exists(EMU_SO)) or os.path.getmtime(EMU_SO) < newest:
        os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-parameter",
                               "-o", EMU_SO, EMU_SRC])


class Emu:
    def __init__(self):
        build_emu()
        L = self.L = C.CDLL(EMU_SO)
        L.emu_encode_block.restype = C.c_int
        L.emu_encode_block.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.emu_decode_block.restype = C.c_int
        L.emu_decode_block.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.emu_xxh32.restype = C.c_uint32
        L.emu_xxh32.argtypes = [u8p, C.c_int]

        L.emu_encode_block_dict.restype = C.c_int
        L.emu_encode_block_dict.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_void_p]
        L.emu_decode_block_dict.restype = C.c_int
        L.emu_decode_block_dict.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int]

    def compress_dict(self, src, cap, dct, mode, table=None):
        dst = np.empty(max(cap, 1) + 32, dtype=np.uint8)
        tp = table.ctypes.data if table is not None else None
        r = int(self.L.emu_encode_block_dict(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap,
                                             _ptr(dct) if dct is not None and dct.size else C.cast(None, u8p),
                                             0 if dct is None else dct.size, mode, tp))
        return r, dst[:max(r, 0)]

    def decompress_dict(self, src, cap, dct):
        dst = np.zeros(max(cap, 1) + 32, dtype=np.uint8)
        r = int(self.L.emu_decode_block_dict(_ptr(src), src.size, _ptr(dst), cap, _ptr(dct), dct.size))
        return r, dst[:max(r, 0)]

    def compress_hc(self, src, cap, level):
        self.L.emu_compress_hc.restype = C.c_int
        self.L.emu_compress_hc.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int]
        dst = np.empty(max(cap, 1) + 32, dtype=np.uint8)
        r = int(self.L.emu_compress_hc(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap, level))
        return r, dst[:max(r, 0)]

    def compress_hc_dict(self, src, cap, level, seg, mode):
        """mode 1: `seg` is the external segment in front of the block (linked tail / dictionary, block > 4 KiB);
        mode 2: `seg` is the dictionary of an attached context (block <= 4 KiB)."""
        self.L.emu_compress_hc_dict.restype = C.c_int
        self.L.emu_compress_hc_dict.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        dst = np.empty(max(cap, 1) + 32, dtype=np.uint8)
        nul = C.cast(None, u8p)
        r = int(self.L.emu_compress_hc_dict(_ptr(src) if src.size else nul, src.size, _ptr(dst), cap, level,
                                            _ptr(seg) if seg.size else nul, seg.size, mode))
        return r, dst[:max(r, 0)]

    def set_old_dict(self, d: bool):
        """True: every dictionary mode through the one-sequence-per-batch encoder (kept for the <= 4 KiB lookup mode), as a
        cross-check of the external-segment mode of the full encoder."""
        self.L.emu_set_old_dict(int(d))

    def set_descending(self, d: bool):
        self.L.emu_set_descending(int(d))

    def compress_fast(self, src: np.ndarray, cap: int):
        """Level 1 as the kernels run it: parse -> emit for blocks up to 4 MiB, the fused encoder above."""
        dst = np.empty(max(cap, 1) + 32, dtype=np.uint8)
        r = int(self.L.emu_encode_block(_ptr(src), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def compress_fast_fused(self, src: np.ndarray, cap: int):
        """The fused encoder of lz4_device.inl (blocks above 4 MiB; dictionary modes run it in its external-segment form)."""
        self.L.emu_encode_block_fused.restype = C.c_int
        self.L.emu_encode_block_fused.argtypes = [u8p, C.c_int, u8p, C.c_int]
        dst = np.empty(max(cap, 1) + 32, dtype=np.uint8)
        r = int(self.L.emu_encode_block_fused(_ptr(src), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def decompress_safe(self, src: np.ndarray, cap: int):
        dst = np.zeros(max(cap, 1) + 32, dtype=np.uint8)
        r = int(self.L.emu_decode_block(_ptr(src), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def dx_decode(self, src: np.ndarray, cap: int):
        """The few-block decoder (lz4_dx_device.inl) stage by stage as the kernels run it.  Returns (size, output, jump rounds);
        size -999999: the block is left to the one-wave decoder."""
        self.L.emu_dx_decode.restype = C.c_int
        self.L.emu_dx_decode.argtypes = [u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int)]
        dst = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
        rounds = C.c_int(0)
        r = int(self.L.emu_dx_decode(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap, C.byref(rounds)))
        return r, dst[:max(r, 0)], rounds.value

    def xxh32(self, a: np.ndarray) -> int:
        return int(self.L.emu_xxh32(_ptr(a) if a.size else C.cast(None, u8p), a.size))

    def compress_hc12(self, src, cap, nc_every=0, nl=1024, max_segs=0, min_seg=8192):
        """Level 12 through the three device phases of lz4hc12_device.inl (chain, per-position search, parser)."""
        self.L.emu_compress_hc12.restype = C.c_int
        self.L.emu_compress_hc12.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        dst = np.empty(max(cap, 1) + 64, dtype=np.uint8)
        r = int(self.L.emu_compress_hc12(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap, nc_every, nl, max_segs, min_seg))
        return r, dst[:max(r, 0)]

    def hc12_search_check(self, src):
        """Positions whose phased search (Hc12Walk, what k_hc12_search runs) differs from the plain one (Hc12Lane)."""
        self.L.emu_hc12_search_check.restype = C.c_int
        self.L.emu_hc12_search_check.argtypes = [u8p, C.c_int, C.c_void_p]
        return int(self.L.emu_hc12_search_check(_ptr(src) if src.size else C.cast(None, u8p), src.size, None))

    def compress_hc_lazy(self, src, cap, level, max_segs=1, min_seg=65536):
        """Levels 3..9 as the kernels run them: segments walked independently, stitched, record emit (lz4hc_lazy_device.inl)."""
        self.L.emu_compress_hc_lazy.restype = C.c_int
        self.L.emu_compress_hc_lazy.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        dst = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
        r = int(self.L.emu_compress_hc_lazy(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap, level, max_segs, min_seg))
        return r, dst[:max(r, 0)]

    def compress_hc_lazy_ext(self, src, cap, level, seg, max_segs=1, min_seg=65536):
        """Levels 3..12 behind an external segment (a linked block, a block > 4 KiB under an attached dictionary) as the kernels
        run them: lists over segment + block, the walk in segments, stitched, record emit."""
        if level == 2:
            return self.compress_hc_mid_ext(src, cap, seg)
        self.L.emu_compress_hc_lazy_ext.restype = C.c_int
        self.L.emu_compress_hc_lazy_ext.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        dst = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
        nul = C.cast(None, u8p)
        r = int(self.L.emu_compress_hc_lazy_ext(_ptr(src) if src.size else nul, src.size, _ptr(dst), cap, level,
                                                _ptr(seg) if seg.size else nul, seg.size, max_segs, min_seg))
        return r, dst[:max(r, 0)]

    def compress_hc_mid_ext(self, src, cap, seg):
        """Level 2 behind an external segment as the kernels run it (hc_mid_parse<true>: tables primed over the segment)."""
        self.L.emu_compress_hc_mid_ext.restype = C.c_int
        self.L.emu_compress_hc_mid_ext.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int]
        dst = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
        nul = C.cast(None, u8p)
        r = int(self.L.emu_compress_hc_mid_ext(_ptr(src) if src.size else nul, src.size, _ptr(dst), cap, _ptr(seg) if seg.size else nul, seg.size))
        return r, dst[:max(r, 0)]

    def compress_hc_mid(self, src, cap):
        """Level 2 as the kernels run it: batches of a literal run over the two tables, records, emit (lz4hc_lazy_device.inl)."""
        self.L.emu_compress_hc_mid.restype = C.c_int
        self.L.emu_compress_hc_mid.argtypes = [u8p, C.c_int, u8p, C.c_int]
        dst = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
        r = int(self.L.emu_compress_hc_mid(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def compress_hc_pre(self, src, cap, level):
        """HC levels 3..11 with the chain built up front (what the kernels run for independent blocks without dictionary)."""
        self.L.emu_compress_hc_pre.restype = C.c_int
        self.L.emu_compress_hc_pre.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int]
        dst = np.empty(max(cap, 1) + 64, dtype=np.uint8)
        r = int(self.L.emu_compress_hc_pre(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap, level))
        return r, dst[:max(r, 0)]

    def compress_hc_lists(self, src, cap, level):
        """HC levels on the chain AND the per-hash lists built up front (levels 4..12: up to 63 candidates per round)."""
        self.L.emu_compress_hc_lists.restype = C.c_int
        self.L.emu_compress_hc_lists.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int]
        dst = np.empty(max(cap, 1) + 64, dtype=np.uint8)
        r = int(self.L.emu_compress_hc_lists(_ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap, level))
        return r, dst[:max(r, 0)]
