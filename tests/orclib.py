"""ctypes bindings for the CHECKERS: oracle/_build/liboracle.so (our CPU restatement) and, when it has
been built, oracle/_ref/liblz4ref.so (the reference's vendored liblz4 v1.10.0, compiled unmodified).
Test infrastructure only -- nothing under plz4_amd/ imports this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblz4ref.so")

u8p = C.POINTER(C.c_uint8)


def _ptr(a):
    if a is None:
        return C.cast(None, u8p)
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(a, dtype=np.uint8)
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


def build_oracle():
    src = os.path.join(ROOT, "oracle", "plz4_oracle.c")
    if (not os.path.exists(ORC_SO)) or os.path.getmtime(ORC_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


class OrcStream(C.Structure):
    pass


OrcStream._fields_ = [
    ("table", C.c_uint32 * 4096),
    ("dictionary", C.c_void_p),
    ("dictCtx", C.c_void_p),
    ("currentOffset", C.c_uint32),
    ("tableType", C.c_uint32),
    ("dictSize", C.c_uint32),
]


class Oracle:
    def __init__(self):
        build_oracle()
        L = self.L = C.CDLL(ORC_SO)
        L.orc_xxh32.restype = C.c_uint32
        L.orc_xxh32.argtypes = [u8p, C.c_size_t]
        L.orc_compress_bound.restype = C.c_int
        L.orc_compress_bound.argtypes = [C.c_int]
        L.orc_compress_fast.restype = C.c_int
        L.orc_compress_fast.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.orc_decompress_safe.restype = C.c_int
        L.orc_decompress_safe.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.orc_decompress_safe_dict.restype = C.c_int
        L.orc_decompress_safe_dict.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int]
        L.orc_block_record.restype = C.c_int
        L.orc_block_record.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_frame_header.restype = C.c_int
        L.orc_frame_header.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_uint32]
        L.orc_frame_encode.restype = C.c_int64
        L.orc_frame_encode.argtypes = [u8p, C.c_int64, C.c_int, C.c_int, C.c_int, u8p, C.c_int64]
        L.orc_frame_decode.restype = C.c_int64
        L.orc_frame_decode.argtypes = [u8p, C.c_int64, u8p, C.c_int64]
        L.orc_stream_init.argtypes = [C.POINTER(OrcStream)]
        L.orc_stream_reset_fast.argtypes = [C.POINTER(OrcStream)]
        L.orc_stream_load_dict.restype = C.c_int
        L.orc_stream_load_dict.argtypes = [C.POINTER(OrcStream), u8p, C.c_int, C.c_int]
        L.orc_stream_attach.argtypes = [C.POINTER(OrcStream), C.POINTER(OrcStream)]
        L.orc_stream_compress.restype = C.c_int
        L.orc_stream_compress.argtypes = [C.POINTER(OrcStream), u8p, C.c_int, u8p, C.c_int]

    def xxh32(self, data) -> int:
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        return int(self.L.orc_xxh32(_ptr(a) if a.size else C.cast(None, u8p), a.size))

    def bound(self, n):
        return int(self.L.orc_compress_bound(n))

    def compress_fast(self, src: np.ndarray, cap: int):
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        r = int(self.L.orc_compress_fast(_ptr(src), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def decompress_safe(self, src: np.ndarray, cap: int):
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = int(self.L.orc_decompress_safe(_ptr(src), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def decompress_safe_dict(self, src, cap, dct):
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = int(self.L.orc_decompress_safe_dict(_ptr(src), src.size, _ptr(dst), cap, _ptr(dct), dct.size))
        return r, dst[:max(r, 0)]

    # ---- streams (config 5): the three ways plz4 primes a level-1 stream (clz4.go:96-120, :160-179, :224-248)
    def dict_ctx(self, dct: np.ndarray) -> "OrcStream":
        """clz4.NewDictCtx: resetStream_fast + LZ4_loadDictSlow."""
        d = OrcStream(); self.L.orc_stream_init(C.byref(d)); self.L.orc_stream_reset_fast(C.byref(d))
        self._keep = getattr(self, "_keep", []); self._keep.append(dct)
        self.L.orc_stream_load_dict(C.byref(d), _ptr(dct) if dct.size else C.cast(None, u8p), dct.size, 1)
        return d

    def compress_indie_dict(self, src: np.ndarray, cap: int, dctx: "OrcStream"):
        """StreamIndieCtx.Compress: resetStream_fast + attach_dictionary + compress_fast_continue on a fresh stream."""
        s = OrcStream(); self.L.orc_stream_init(C.byref(s)); self.L.orc_stream_reset_fast(C.byref(s))
        self.L.orc_stream_attach(C.byref(s), C.byref(dctx))
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        r = int(self.L.orc_stream_compress(C.byref(s), _ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def compress_linked(self, src: np.ndarray, cap: int, prev_tail, dctx=None):
        """StreamLinkedCtx.Compress for one block: block 0 (prev_tail None) runs on the fresh stream (+ attached dict ctx),
        later blocks LZ4_loadDict(prev_tail) first.  Equivalent to the worker-owned stream because loadDict resets it."""
        s = OrcStream(); self.L.orc_stream_init(C.byref(s)); self.L.orc_stream_reset_fast(C.byref(s))
        if dctx is not None:
            self.L.orc_stream_attach(C.byref(s), C.byref(dctx))
        if prev_tail is not None:
            self.L.orc_stream_load_dict(C.byref(s), _ptr(prev_tail) if prev_tail.size else C.cast(None, u8p), prev_tail.size, 0)
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        r = int(self.L.orc_stream_compress(C.byref(s), _ptr(src) if src.size else C.cast(None, u8p), src.size, _ptr(dst), cap))
        return r, dst[:max(r, 0)]

    def block_record(self, src: np.ndarray, bsz: int, block_checksum: bool):
        rec = np.empty(bsz + 8, dtype=np.uint8)
        r = int(self.L.orc_block_record(_ptr(src), src.size, bsz, int(block_checksum), _ptr(rec)))
        return rec[:r]

    def frame_header(self, bs_idx, linked=False, block_checksum=False, content_checksum=False,
                     content_size=None, dict_id=None) -> bytes:
        out = np.zeros(19, dtype=np.uint8)
        n = self.L.orc_frame_header(_ptr(out), bs_idx, int(linked), int(block_checksum), int(content_checksum),
                                    int(content_size is not None), int(content_size or 0),
                                    int(dict_id is not None), int(dict_id or 0))
        return out[:n].tobytes()

    def frame_encode(self, src: np.ndarray, bs_idx: int, block_checksum: bool, content_checksum: bool):
        bsz = {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20}[bs_idx]
        nblk = (src.size + bsz - 1) // bsz
        cap = 19 + src.size + nblk * 8 + bsz + 16
        out = np.empty(cap, dtype=np.uint8)
        r = int(self.L.orc_frame_encode(_ptr(src) if src.size else C.cast(None, u8p), src.size, bs_idx,
                                        int(block_checksum), int(content_checksum), _ptr(out), cap))
        assert r >= 0
        return out[:r]

    def frame_decode(self, frame: np.ndarray, cap: int):
        out = np.empty(max(cap, 1), dtype=np.uint8)
        r = int(self.L.orc_frame_decode(_ptr(frame), frame.size, _ptr(out), cap))
        return r, out[:max(r, 0)]


# an empty block still gets a valid address: with NULL the optimal parser of the reference (levels 10-12) dereferences it
_ONE = np.zeros(16, dtype=np.uint8)


class Ref:
    """The real thing: liblz4 v1.10.0 from /root/reference/internal/pkg/clz4 (oracle/Makefile `ref`)."""

    def __init__(self):
        L = self.L = C.CDLL(REF_SO)
        L.LZ4_versionString.restype = C.c_char_p
        assert L.LZ4_versionString() == b"1.10.0"
        L.LZ4_compress_fast.restype = C.c_int
        L.LZ4_compress_fast.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.LZ4_decompress_safe.restype = C.c_int
        L.LZ4_decompress_safe.argtypes = [u8p, u8p, C.c_int, C.c_int]
        L.LZ4_decompress_safe_usingDict.restype = C.c_int
        L.LZ4_decompress_safe_usingDict.argtypes = [u8p, u8p, C.c_int, C.c_int, u8p, C.c_int]
        L.LZ4_compress_HC.restype = C.c_int
        L.LZ4_compress_HC.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.LZ4_compressBound.restype = C.c_int
        L.LZ4_compressBound.argtypes = [C.c_int]
        L.LZ4_sizeofState.restype = C.c_int
        for f in ("LZ4_resetStream_fast",):
            getattr(L, f).argtypes = [C.c_void_p]
        L.LZ4_initStream.restype = C.c_void_p
        L.LZ4_initStream.argtypes = [C.c_void_p, C.c_size_t]
        L.LZ4_loadDict.restype = C.c_int
        L.LZ4_loadDict.argtypes = [C.c_void_p, u8p, C.c_int]
        L.LZ4_loadDictSlow.restype = C.c_int
        L.LZ4_loadDictSlow.argtypes = [C.c_void_p, u8p, C.c_int]
        L.LZ4_attach_dictionary.argtypes = [C.c_void_p, C.c_void_p]
        L.LZ4_compress_fast_continue.restype = C.c_int
        L.LZ4_compress_fast_continue.argtypes = [C.c_void_p, u8p, u8p, C.c_int, C.c_int, C.c_int]

        L.LZ4_sizeofStateHC.restype = C.c_int
        L.LZ4_initStreamHC.restype = C.c_void_p
        L.LZ4_initStreamHC.argtypes = [C.c_void_p, C.c_size_t]
        L.LZ4_resetStreamHC_fast.argtypes = [C.c_void_p, C.c_int]
        L.LZ4_loadDictHC.restype = C.c_int
        L.LZ4_loadDictHC.argtypes = [C.c_void_p, u8p, C.c_int]
        L.LZ4_attach_HC_dictionary.argtypes = [C.c_void_p, C.c_void_p]
        L.LZ4_compress_HC_continue.restype = C.c_int
        L.LZ4_compress_HC_continue.argtypes = [C.c_void_p, u8p, u8p, C.c_int, C.c_int]

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    # ---- the Go stream types of clz4.go over the real liblz4 (HC): a Go zero value is zeroed memory
    def _zeroed_hc(self):
        buf = C.create_string_buffer(self.L.LZ4_sizeofStateHC() + 64)
        return buf, (C.addressof(buf) + 15) & ~15

    def new_dict_ctx_hc(self, dct: np.ndarray, level: int):
        """clz4.NewDictCtxHC (clz4.go:127-147).  Returns (keepalive, address)."""
        data = np.ascontiguousarray(dct).copy()
        buf, addr = self._zeroed_hc()
        self.L.LZ4_resetStreamHC_fast(addr, level)
        self.L.LZ4_loadDictHC(addr, _ptr(data) if data.size else C.cast(None, u8p), data.size)
        return (buf, data), addr

    def stream_ctx_hc(self, level: int, dict_addr):
        """clz4.StreamCtxHC (clz4.go:181-209): returns compress(src, cap) on ONE reused stream, as a worker would."""
        buf, addr = self._zeroed_hc()

        def compress(src: np.ndarray, cap: int):
            self.L.LZ4_resetStreamHC_fast(addr, level)
            self.L.LZ4_attach_HC_dictionary(addr, dict_addr)
            dst = np.empty(max(cap, 1), dtype=np.uint8)
            r = int(self.L.LZ4_compress_HC_continue(addr, _ptr(src if src.size else _ONE), _ptr(dst), src.size, cap))
            return r, dst[:max(r, 0)]
        compress._keep = buf
        return compress

    def stream_linked_ctx_hc(self, level: int, dict_addr=None):
        """clz4.StreamLinkedCtxHC (clz4.go:250-283): returns compress(src, cap, tail_or_None)."""
        buf, addr = self._zeroed_hc()
        self.L.LZ4_resetStreamHC_fast(addr, level)
        if dict_addr is not None:
            self.L.LZ4_attach_HC_dictionary(addr, dict_addr)

        def compress(src: np.ndarray, cap: int, tail):
            if tail is not None:
                t = np.ascontiguousarray(tail)
                self.L.LZ4_loadDictHC(addr, _ptr(t) if t.size else C.cast(None, u8p), t.size)
                compress._tail = t
            dst = np.empty(max(cap, 1), dtype=np.uint8)
            r = int(self.L.LZ4_compress_HC_continue(addr, _ptr(src if src.size else _ONE), _ptr(dst), src.size, cap))
            return r, dst[:max(r, 0)]
        compress._keep = buf
        return compress

    def new_stream(self):
        buf = C.create_string_buffer(self.L.LZ4_sizeofState() + 64)
        addr = (C.addressof(buf) + 15) & ~15
        self.L.LZ4_initStream(addr, self.L.LZ4_sizeofState())
        return buf, addr

    def compress_fast(self, src: np.ndarray, cap: int):
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        r = int(self.L.LZ4_compress_fast(_ptr(src), _ptr(dst), src.size, cap, 1))
        return r, dst[:max(r, 0)]

    def compress_hc(self, src: np.ndarray, cap: int, level: int):
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        r = int(self.L.LZ4_compress_HC(_ptr(src), _ptr(dst), src.size, cap, level))
        return r, dst[:max(r, 0)]

    def decompress_safe(self, src: np.ndarray, cap: int):
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = int(self.L.LZ4_decompress_safe(_ptr(src), _ptr(dst), src.size, cap))
        return r, dst[:max(r, 0)]

    def decompress_safe_dict(self, src, cap, dct):
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        r = int(self.L.LZ4_decompress_safe_usingDict(_ptr(src), _ptr(dst), src.size, cap, _ptr(dct), dct.size))
        return r, dst[:max(r, 0)]
