"""ctypes binding of the C ABI (include/plz4hip.h) implemented by plz4_amd/libplz4hip.so.

There is NO fallback: if the HIP library is missing or cannot be loaded this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplz4hip.so")

ABI_VERSION = 2
SYMBOLS = [
    "plz4hip_abi_version", "plz4hip_device_count", "plz4hip_ctx_create", "plz4hip_ctx_destroy",
    "plz4hip_last_error", "plz4hip_compress_bound", "plz4hip_compress_batch", "plz4hip_decompress_batch",
    "plz4hip_xxh32_batch", "plz4hip_encode_records", "plz4hip_decode_records", "plz4hip_dev_stage_stride",
    "plz4hip_dev_encode_records", "plz4hip_dev_compact_records", "plz4hip_dev_scatter_records",
    "plz4hip_dev_decode_records", "plz4hip_dev_duplex_records", "plz4hip_dev_encode_body", "plz4hip_dev_duplex_body", "plz4hip_dev_compress", "plz4hip_dev_decompress", "plz4hip_ctx_trim",
    "plz4hip_dev_resident_waves", "plz4hip_dict_create", "plz4hip_dict_destroy", "plz4hip_compress_batch_dict", "plz4hip_decode_records_chains",
    "plz4hip_decompress_batch_dict", "plz4hip_encode_records_ex", "plz4hip_decode_records_ex",
    "plz4hip_xxh32_stream_create", "plz4hip_xxh32_stream_destroy", "plz4hip_xxh32_stream_reset", "plz4hip_xxh32_stream_update",
    "plz4hip_dev_xxh32_stream_update", "plz4hip_xxh32_stream_sum", "plz4hip_ctx_set_content_hash",
    "plz4hip_mgpu_create", "plz4hip_mgpu_destroy", "plz4hip_mgpu_count", "plz4hip_mgpu_ctx", "plz4hip_mgpu_last_error",
    "plz4hip_mgpu_compress_batch", "plz4hip_mgpu_decompress_batch", "plz4hip_mgpu_encode_records", "plz4hip_mgpu_decode_records",
    "plz4hip_mgpu_dev_encode_frame", "plz4hip_mgpu_dev_decode_frame",
]

E_NAMES = {0: "OK", -1: "E_ARG", -2: "E_DEVICE", -3: "E_NOMEM", -4: "E_UNSUPPORTED"}
BLK_OK, BLK_HASH_MISMATCH, BLK_SIZE_OVERFLOW, BLK_CORRUPT = 0, 1, 2, 3


class EngineError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("plz4hip: %s (%s)" % (E_NAMES.get(code, code), text))
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("plz4_amd: %s is missing -- build it with `python -m plz4_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7).  If torch
    # is importable, load it FIRST so that our NEEDED libamdhip64.so.7 binds to the copy torch uses; loaded the
    # other way round the process ends up with two runtimes and torch sees no GPU.
    if os.environ.get("PLZ4_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32p, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    pp = C.POINTER(C.c_void_p)
    L.plz4hip_abi_version.restype = C.c_int
    L.plz4hip_device_count.restype = C.c_int
    L.plz4hip_ctx_create.restype = C.c_int
    L.plz4hip_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.plz4hip_ctx_destroy.argtypes = [vp]
    L.plz4hip_last_error.restype = C.c_char_p
    L.plz4hip_last_error.argtypes = [vp]
    L.plz4hip_compress_bound.restype = C.c_int
    L.plz4hip_compress_bound.argtypes = [C.c_int]
    L.plz4hip_compress_batch.restype = C.c_int
    L.plz4hip_compress_batch.argtypes = [vp, C.c_int, pp, i32p, pp, i32p, C.c_int, i32p]
    L.plz4hip_decompress_batch.restype = C.c_int
    L.plz4hip_decompress_batch.argtypes = [vp, C.c_int, pp, i32p, pp, i32p, i32p]
    L.plz4hip_xxh32_batch.restype = C.c_int
    L.plz4hip_xxh32_batch.argtypes = [vp, C.c_int, pp, i32p, C.POINTER(C.c_uint32)]
    L.plz4hip_encode_records.restype = C.c_int
    L.plz4hip_encode_records.argtypes = [vp, C.c_int, pp, i32p, C.c_int, C.c_int, C.c_int, pp, i32p]
    L.plz4hip_decode_records.restype = C.c_int
    L.plz4hip_decode_records.argtypes = [vp, C.c_int, pp, i32p, C.c_int, C.c_int, pp, i32p, i32p]
    L.plz4hip_dev_stage_stride.restype = C.c_int64
    L.plz4hip_dev_stage_stride.argtypes = [C.c_int]
    L.plz4hip_dev_encode_records.restype = C.c_int
    L.plz4hip_dev_encode_records.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.plz4hip_dev_compact_records.restype = C.c_int
    L.plz4hip_dev_compact_records.argtypes = [vp, vp, C.c_int64, vp, C.c_int, vp, vp, C.c_int64, vp]
    L.plz4hip_dev_scatter_records.restype = C.c_int
    L.plz4hip_dev_scatter_records.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int64, vp]
    L.plz4hip_dev_decode_records.restype = C.c_int
    L.plz4hip_dev_decode_records.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, C.c_int, vp, vp, vp]
    L.plz4hip_dev_duplex_records.restype = C.c_int
    L.plz4hip_dev_duplex_records.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp,
                                             vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, C.c_int, vp, vp, vp]
    L.plz4hip_dev_encode_body.restype = C.c_int
    L.plz4hip_dev_encode_body.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, vp, vp]
    L.plz4hip_dev_duplex_body.restype = C.c_int
    L.plz4hip_dev_duplex_body.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int64, vp, vp,
                                          vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, C.c_int, vp, vp, vp]
    L.plz4hip_dev_compress.restype = C.c_int
    L.plz4hip_dev_compress.argtypes = [vp, C.c_int, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp]
    L.plz4hip_dev_decompress.restype = C.c_int
    L.plz4hip_dev_decompress.argtypes = [vp, C.c_int, vp, C.c_int64, vp, vp, C.c_int64, vp, vp, vp]
    L.plz4hip_dev_resident_waves.restype = C.c_int
    L.plz4hip_dev_resident_waves.argtypes = [vp, C.c_int]
    L.plz4hip_dict_create.restype = C.c_int
    L.plz4hip_dict_create.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.plz4hip_dict_destroy.argtypes = [vp, vp]
    L.plz4hip_compress_batch_dict.restype = C.c_int
    L.plz4hip_compress_batch_dict.argtypes = [vp, C.c_int, pp, i32p, pp, i32p, C.c_int, vp, i32p]
    L.plz4hip_decompress_batch_dict.restype = C.c_int
    L.plz4hip_decompress_batch_dict.argtypes = [vp, C.c_int, pp, i32p, pp, i32p, vp, i32p]
    L.plz4hip_decode_records_chains.restype = C.c_int
    L.plz4hip_decode_records_chains.argtypes = [vp, C.c_int, i32p, pp, i32p, C.c_int, C.c_int, vp, i32p, pp, i32p, i32p]
    L.plz4hip_encode_records_ex.restype = C.c_int
    L.plz4hip_encode_records_ex.argtypes = [vp, C.c_int, pp, i32p, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, pp, i32p]
    L.plz4hip_decode_records_ex.restype = C.c_int
    L.plz4hip_decode_records_ex.argtypes = [vp, C.c_int, pp, i32p, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_int), pp, i32p, i32p]
    L.plz4hip_xxh32_stream_create.restype = C.c_int
    L.plz4hip_xxh32_stream_create.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.plz4hip_xxh32_stream_destroy.restype = None
    L.plz4hip_xxh32_stream_destroy.argtypes = [vp, vp]
    L.plz4hip_xxh32_stream_reset.restype = C.c_int
    L.plz4hip_xxh32_stream_reset.argtypes = [vp, vp]
    L.plz4hip_xxh32_stream_update.restype = C.c_int
    L.plz4hip_xxh32_stream_update.argtypes = [vp, vp, vp, C.c_int64]
    L.plz4hip_dev_xxh32_stream_update.restype = C.c_int
    L.plz4hip_dev_xxh32_stream_update.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.plz4hip_xxh32_stream_sum.restype = C.c_int
    L.plz4hip_xxh32_stream_sum.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    L.plz4hip_ctx_set_content_hash.restype = C.c_int
    L.plz4hip_ctx_set_content_hash.argtypes = [vp, vp]
    L.plz4hip_ctx_trim.restype = C.c_int
    L.plz4hip_ctx_trim.argtypes = [vp]
    L.plz4hip_mgpu_create.restype = C.c_int
    L.plz4hip_mgpu_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.plz4hip_mgpu_destroy.restype = None
    L.plz4hip_mgpu_destroy.argtypes = [vp]
    L.plz4hip_mgpu_count.restype = C.c_int
    L.plz4hip_mgpu_count.argtypes = [vp]
    L.plz4hip_mgpu_ctx.restype = C.c_void_p
    L.plz4hip_mgpu_ctx.argtypes = [vp, C.c_int]
    L.plz4hip_mgpu_last_error.restype = C.c_char_p
    L.plz4hip_mgpu_last_error.argtypes = [vp]
    L.plz4hip_mgpu_compress_batch.restype = C.c_int
    L.plz4hip_mgpu_compress_batch.argtypes = [vp, C.c_int, pp, i32p, pp, i32p, C.c_int, i32p]
    L.plz4hip_mgpu_decompress_batch.restype = C.c_int
    L.plz4hip_mgpu_decompress_batch.argtypes = [vp, C.c_int, pp, i32p, pp, i32p, i32p]
    L.plz4hip_mgpu_encode_records.restype = C.c_int
    L.plz4hip_mgpu_encode_records.argtypes = [vp, C.c_int, pp, i32p, C.c_int, C.c_int, C.c_int, pp, i32p]
    L.plz4hip_mgpu_decode_records.restype = C.c_int
    L.plz4hip_mgpu_decode_records.argtypes = [vp, C.c_int, pp, i32p, C.c_int, C.c_int, pp, i32p, i32p]
    L.plz4hip_mgpu_dev_encode_frame.restype = C.c_int
    L.plz4hip_mgpu_dev_encode_frame.argtypes = [vp, C.c_int, pp, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int64,
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.plz4hip_mgpu_dev_decode_frame.restype = C.c_int
    L.plz4hip_mgpu_dev_decode_frame.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int64), C.c_int, C.c_int, pp, C.c_int64, C.c_int,
                                                i32p, i32p]
    if L.plz4hip_abi_version() != ABI_VERSION:
        raise ImportError("plz4_amd: libplz4hip.so ABI %d != binding %d" % (L.plz4hip_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _ptr_array(bufs):
    arr = (C.c_void_p * len(bufs))()
    for i, b in enumerate(bufs):
        arr[i] = b.ctypes.data if (b is not None and b.size) else None
    return arr


def _i32(vals):
    return np.ascontiguousarray(np.asarray(vals, dtype=np.int32))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Engine:
    """One plz4hip_ctx.  Mirrors the reference's per-worker Compressor/Decompressor instantiation
    (internal/pkg/async/writer.go:237, compress/compress.go:50-57) in batch form."""

    def __init__(self, device: int = 0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.plz4hip_ctx_create(device, C.byref(h))
        if rc != 0:
            raise EngineError(rc, "plz4hip_ctx_create(device=%d) failed -- is an MI355X visible?" % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.plz4hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, (self.L.plz4hip_last_error(self.h) or b"").decode())

    @staticmethod
    def compress_bound(n: int) -> int:
        return int(load().plz4hip_compress_bound(n))

    def resident_waves(self, decode: bool) -> int:
        return int(self.L.plz4hip_dev_resident_waves(self.h, int(decode)))

    # ---- A. batched Compressor / Decompressor over host buffers
    def compress_batch(self, srcs, caps, level: int = 1):
        n = len(srcs)
        lens = _i32([s.size for s in srcs]); capa = _i32(caps)
        dsts = [np.empty(max(int(c), 1), dtype=np.uint8) for c in caps]
        res = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plz4hip_compress_batch(self.h, n, _ptr_array(srcs), _i32p(lens), _ptr_array(dsts), _i32p(capa),
                                                level, _i32p(res)))
        return res, [d[:max(int(r), 0)] for d, r in zip(dsts, res)]

    def decompress_batch(self, srcs, caps):
        n = len(srcs)
        lens = _i32([s.size for s in srcs]); capa = _i32(caps)
        dsts = [np.zeros(max(int(c), 1), dtype=np.uint8) for c in caps]
        res = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plz4hip_decompress_batch(self.h, n, _ptr_array(srcs), _i32p(lens), _ptr_array(dsts), _i32p(capa),
                                                  _i32p(res)))
        return res, [d[:max(int(r), 0)] for d, r in zip(dsts, res)]

    def xxh32_batch(self, bufs):
        n = len(bufs)
        lens = _i32([b.size for b in bufs])
        out = np.zeros(n, dtype=np.uint32)
        self._chk(self.L.plz4hip_xxh32_batch(self.h, n, _ptr_array(bufs), _i32p(lens), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    # ---- B. frame records over host buffers
    def encode_records(self, srcs, bsz: int, block_checksum: bool, level: int = 1):  # level 1 or 10..12
        n = len(srcs)
        lens = _i32([s.size for s in srcs])
        recs = [np.empty(bsz + 8, dtype=np.uint8) for _ in range(n)]
        rl = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plz4hip_encode_records(self.h, n, _ptr_array(srcs), _i32p(lens), bsz, level, int(block_checksum),
                                                _ptr_array(recs), _i32p(rl)))
        return [r[:int(k)] for r, k in zip(recs, rl)]

    def decode_records(self, recs, bsz: int, block_checksum: bool):
        n = len(recs)
        lens = _i32([r.size for r in recs])
        dsts = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(n)]
        res = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plz4hip_decode_records(self.h, n, _ptr_array(recs), _i32p(lens), bsz, int(block_checksum),
                                                _ptr_array(dsts), _i32p(res), _i32p(st)))
        return res, st, [d[:max(int(r), 0)] for d, r in zip(dsts, res)]

    # ---- B'. dictionaries / linked blocks
    def dict_create(self, dct: np.ndarray):
        h = C.c_void_p()
        self._chk(self.L.plz4hip_dict_create(self.h, dct.ctypes.data if dct.size else None, dct.size, C.byref(h)))
        return h

    def dict_destroy(self, d):
        self.L.plz4hip_dict_destroy(self.h, d)

    def compress_batch_dict(self, srcs, caps, d, level: int = 1):
        n = len(srcs); lens = _i32([s.size for s in srcs]); capa = _i32(caps)
        dsts = [np.empty(max(int(c), 1), dtype=np.uint8) for c in caps]; res = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plz4hip_compress_batch_dict(self.h, n, _ptr_array(srcs), _i32p(lens), _ptr_array(dsts), _i32p(capa), level, d, _i32p(res)))
        return res, [x[:max(int(r), 0)] for x, r in zip(dsts, res)]

    def decompress_batch_dict(self, srcs, caps, d):
        n = len(srcs); lens = _i32([s.size for s in srcs]); capa = _i32(caps)
        dsts = [np.zeros(max(int(c), 1), dtype=np.uint8) for c in caps]; res = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plz4hip_decompress_batch_dict(self.h, n, _ptr_array(srcs), _i32p(lens), _ptr_array(dsts), _i32p(capa), d, _i32p(res)))
        return res, [x[:max(int(r), 0)] for x, r in zip(dsts, res)]

    def encode_records_ex(self, srcs, bsz, block_checksum, linked=False, d=None, prev_tail=None, level: int = 1):
        n = len(srcs); lens = _i32([s.size for s in srcs])
        recs = [np.empty(bsz + 8, dtype=np.uint8) for _ in range(n)]; rl = np.zeros(n, dtype=np.int32)
        pt = prev_tail.ctypes.data if (prev_tail is not None and prev_tail.size) else (np.zeros(1, np.uint8).ctypes.data if prev_tail is not None else None)
        self._keep_pt = prev_tail
        self._chk(self.L.plz4hip_encode_records_ex(self.h, n, _ptr_array(srcs), _i32p(lens), bsz, level, int(block_checksum), int(linked), d,
                                                   pt, 0 if prev_tail is None else prev_tail.size, _ptr_array(recs), _i32p(rl)))
        return [r[:int(k)] for r, k in zip(recs, rl)]

    def decode_records_ex(self, recs, bsz, block_checksum, linked=False, d=None, window=None, window_len=0):
        n = len(recs); lens = _i32([r.size for r in recs])
        dsts = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(n)]
        res = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
        wl = C.c_int(window_len)
        self._chk(self.L.plz4hip_decode_records_ex(self.h, n, _ptr_array(recs), _i32p(lens), bsz, int(block_checksum), int(linked), d,
                                                   window.ctypes.data if window is not None else None, C.byref(wl),
                                                   _ptr_array(dsts), _i32p(res), _i32p(st)))
        return res, st, [x[:max(int(r), 0)] for x, r in zip(dsts, res)], wl.value

    def decode_records_chains(self, chains, bsz, block_checksum, windows=None, window_lens=None):
        """chains: list of lists of records (one linked frame each).  windows: uint8[nChains, 65536] (in/out) or None for empty
        windows.  Returns (res, st, outs) per chain, and the window lengths."""
        nch = len(chains)
        first = np.zeros(nch + 1, dtype=np.int32)
        for k, ch in enumerate(chains):
            first[k + 1] = first[k] + len(ch)
        recs = [r for ch in chains for r in ch]
        n = len(recs); lens = _i32([r.size for r in recs])
        dsts = [np.zeros(bsz + 8, dtype=np.uint8) for _ in range(n)]
        res = np.zeros(max(n, 1), dtype=np.int32); st = np.zeros(max(n, 1), dtype=np.int32)
        if windows is None:
            windows = np.zeros((max(nch, 1), 65536), dtype=np.uint8)
        wl = np.zeros(max(nch, 1), dtype=np.int32) if window_lens is None else np.ascontiguousarray(window_lens, dtype=np.int32)
        self._chk(self.L.plz4hip_decode_records_chains(self.h, nch, _i32p(first), _ptr_array(recs) if n else None, _i32p(lens) if n else None,
                                                       bsz, int(block_checksum), windows.ctypes.data, _i32p(wl),
                                                       _ptr_array(dsts) if n else None, _i32p(res), _i32p(st)))
        out = []
        for k in range(nch):
            a, b = int(first[k]), int(first[k + 1])
            out.append((res[a:b].copy(), st[a:b].copy(), [d[:max(int(r), 0)] for d, r in zip(dsts[a:b], res[a:b])]))
        return out, wl

    # ---- C. device-resident pipeline (raw device pointers; torch tensors supply .data_ptr())
    def dev_encode_records(self, src_ptr, src_bytes, bsz, block_checksum, stage_ptr, reclen_ptr, stream=0, level: int = 1):
        self._chk(self.L.plz4hip_dev_encode_records(self.h, src_ptr, src_bytes, bsz, level, int(block_checksum), stage_ptr,
                                                    reclen_ptr, stream))

    def dev_compact_records(self, stage_ptr, stage_stride, reclen_ptr, nblocks, recoff_ptr, body_ptr, body_cap, stream=0):
        self._chk(self.L.plz4hip_dev_compact_records(self.h, stage_ptr, stage_stride, reclen_ptr, nblocks, recoff_ptr,
                                                     body_ptr, body_cap, stream))

    def dev_scatter_records(self, src_ptr, srcoff_ptr, len_ptr, dstoff_ptr, n, max_len, dst_ptr, dst_cap, stream=0):
        self._chk(self.L.plz4hip_dev_scatter_records(self.h, src_ptr, srcoff_ptr, len_ptr, dstoff_ptr, n, max_len, dst_ptr,
                                                     dst_cap, stream))

    def dev_decode_records(self, body_ptr, recoff_ptr, nblocks, bsz, block_checksum, dst_ptr, dst_stride, dst_cap,
                           result_ptr, status_ptr, stream=0):
        self._chk(self.L.plz4hip_dev_decode_records(self.h, body_ptr, recoff_ptr, nblocks, bsz, int(block_checksum), dst_ptr,
                                                    dst_stride, dst_cap, result_ptr, status_ptr, stream))

    def dev_duplex_records(self, src_ptr, src_bytes, bsz, block_checksum, stage_ptr, reclen_ptr,
                           body_ptr, recoff_ptr, ndec, dec_bsz, dec_block_checksum, dst_ptr, dst_stride, dst_cap,
                           result_ptr, status_ptr, stream=0):
        """level-1 encode of one batch and record decode of another in one call (k_l1_duplex)"""
        self._chk(self.L.plz4hip_dev_duplex_records(self.h, src_ptr, src_bytes, bsz, int(block_checksum), stage_ptr, reclen_ptr,
                                                    body_ptr, recoff_ptr, ndec, dec_bsz, int(dec_block_checksum), dst_ptr,
                                                    dst_stride, dst_cap, result_ptr, status_ptr, stream))

    def dev_encode_body(self, src_ptr, src_bytes, bsz, block_checksum, body_ptr, body_cap, recoff_ptr, reclen_ptr, stream=0, level=1):
        """records straight into the frame body: record i at body + recOff[i], recOff[n] = the body's length"""
        self._chk(self.L.plz4hip_dev_encode_body(self.h, src_ptr, src_bytes, bsz, level, int(block_checksum), body_ptr, body_cap,
                                                 recoff_ptr, reclen_ptr, stream))

    def dev_duplex_body(self, src_ptr, src_bytes, bsz, block_checksum, body_ptr, body_cap, recoff_ptr, reclen_ptr,
                        dec_body_ptr, dec_recoff_ptr, ndec, dec_bsz, dec_block_checksum, dst_ptr, dst_stride, dst_cap,
                        result_ptr, status_ptr, stream=0):
        """dev_encode_body of one batch and the record decode of another in one call (k_l1_duplex)"""
        self._chk(self.L.plz4hip_dev_duplex_body(self.h, src_ptr, src_bytes, bsz, int(block_checksum), body_ptr, body_cap, recoff_ptr, reclen_ptr,
                                                 dec_body_ptr, dec_recoff_ptr, ndec, dec_bsz, int(dec_block_checksum), dst_ptr,
                                                 dst_stride, dst_cap, result_ptr, status_ptr, stream))

    def stage_stride(self, bsz: int) -> int:
        return int(self.L.plz4hip_dev_stage_stride(bsz))

    # ---- B''. streaming content checksum (xxh32.XXHZero) on the device
    def hash_create(self):
        h = C.c_void_p()
        self._chk(self.L.plz4hip_xxh32_stream_create(self.h, C.byref(h)))
        return h

    def hash_destroy(self, h):
        self.L.plz4hip_xxh32_stream_destroy(self.h, h)

    def hash_reset(self, h):
        self._chk(self.L.plz4hip_xxh32_stream_reset(self.h, h))

    def hash_update(self, h, buf: np.ndarray):
        buf = np.ascontiguousarray(buf)
        self._chk(self.L.plz4hip_xxh32_stream_update(self.h, h, buf.ctypes.data if buf.size else None, buf.size))

    def dev_hash_update(self, h, ptr, n, stream=0):
        self._chk(self.L.plz4hip_dev_xxh32_stream_update(self.h, h, ptr, n, stream))

    def hash_sum(self, h) -> int:
        out = C.c_uint32()
        self._chk(self.L.plz4hip_xxh32_stream_sum(self.h, h, C.byref(out)))
        return int(out.value)

    def set_content_hash(self, h):
        self._chk(self.L.plz4hip_ctx_set_content_hash(self.h, h))

    def trim(self):
        self._chk(self.L.plz4hip_ctx_trim(self.h))


class MultiEngine:
    """plz4hip_mgpu: block i of a call runs on device i mod G (section D of include/plz4hip.h)."""

    def __init__(self, devices):
        self.L = load()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.L.plz4hip_mgpu_create(arr, len(devices), C.byref(h))
        if rc != 0:
            raise EngineError(rc, "plz4hip_mgpu_create")
        self.h = h
        self.g = len(devices)

    def close(self):
        if getattr(self, "h", None):
            self.L.plz4hip_mgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, (self.L.plz4hip_mgpu_last_error(self.h) or b"").decode())

    def encode_records(self, srcs, bsz, block_checksum, level=1):
        n = len(srcs)
        recs = [np.empty(bsz + 8, dtype=np.uint8) for _ in range(n)]
        lens = _i32([s.size for s in srcs]); rl = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.L.plz4hip_mgpu_encode_records(self.h, n, _ptr_array(srcs), _i32p(lens), bsz, level, int(block_checksum),
                                                     _ptr_array(recs), _i32p(rl)))
        return [r[:int(k)] for r, k in zip(recs, rl[:n])]

    def decode_records(self, recs, bsz, block_checksum):
        n = len(recs)
        dsts = [np.empty(bsz + 8, dtype=np.uint8) for _ in range(n)]
        lens = _i32([r.size for r in recs]); res = np.zeros(max(n, 1), dtype=np.int32); st = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.L.plz4hip_mgpu_decode_records(self.h, n, _ptr_array(recs), _i32p(lens), bsz, int(block_checksum),
                                                     _ptr_array(dsts), _i32p(res), _i32p(st)))
        return res[:n], st[:n], [d[:max(int(r), 0)] for d, r in zip(dsts, res[:n])]

    def compress_batch(self, srcs, caps, level=1):
        n = len(srcs)
        dsts = [np.empty(max(c, 1), dtype=np.uint8) for c in caps]
        lens = _i32([s.size for s in srcs]); cp = _i32(caps); res = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.L.plz4hip_mgpu_compress_batch(self.h, n, _ptr_array(srcs), _i32p(lens), _ptr_array(dsts), _i32p(cp), level, _i32p(res)))
        return res[:n], [d[:max(int(r), 0)] for d, r in zip(dsts, res[:n])]

    def dev_encode_frame(self, shard_ptrs, shard_bytes, nblocks, bsz, block_checksum, owner, body_ptr, body_cap, level=1):
        sp = (C.c_void_p * self.g)(*shard_ptrs); sb = (C.c_int64 * self.g)(*shard_bytes)
        off = (C.c_int64 * (nblocks + 1))(); total = C.c_int64()
        self._chk(self.L.plz4hip_mgpu_dev_encode_frame(self.h, nblocks, sp, sb, bsz, level, int(block_checksum), owner, body_ptr, body_cap,
                                                       off, C.byref(total)))
        return np.array(off[:], dtype=np.int64), int(total.value)

    def dev_decode_frame(self, nblocks, owner, body_ptr, rec_off, bsz, block_checksum, shard_dst_ptrs, dst_stride, dst_cap):
        off = (C.c_int64 * (nblocks + 1))(*[int(x) for x in rec_off])
        dp = (C.c_void_p * self.g)(*shard_dst_ptrs)
        res = np.zeros(max(nblocks, 1), dtype=np.int32); st = np.zeros(max(nblocks, 1), dtype=np.int32)
        self._chk(self.L.plz4hip_mgpu_dev_decode_frame(self.h, nblocks, owner, body_ptr, off, bsz, int(block_checksum), dp, dst_stride, dst_cap,
                                                       _i32p(res), _i32p(st)))
        return res[:nblocks], st[:nblocks]
