"""Python view of the C++ host layer (plz4_amd/csrc/host): NewWriter / NewReader / CompressBlock / DecompressBlock with
the reference's option names, over a block engine.  The engine is the HIP one (`hip_engine`) in production; tests may
plug another one in through the same vtable the C++ layer exposes (`vtable_engine`)."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _native

# zerr.go error constants in plz4_host.hpp order
(OK, EndMark, ErrClosed, ErrCorrupted, ErrHeaderHash, ErrBlockHash, ErrContentHash, ErrHeaderRead, ErrHeaderWrite, ErrMagic,
 ErrVersion, ErrDescriptorRead, ErrBlockSizeRead, ErrBlockRead, ErrBlockSizeOverflow, ErrCompress, ErrDecompress,
 ErrReserveBitSet, ErrBlockDescriptor, ErrContentHashRead, ErrContentSize, ErrReadOffset, ErrReadOffsetLinked, ErrSkip,
 ErrNibble, ErrUnsupported, ErrIO, ErrEOF, ErrEngine) = range(29)

BlockIdx64KB, BlockIdx256KB, BlockIdx1MB, BlockIdx4MB = 4, 5, 6, 7


class Err(int):
    """Error code; `.corrupted` mirrors errors.Is(err, plz4.ErrCorrupted) (plz4_err.go:43-45)."""
    corrupted = False

    def __new__(cls, raw):
        o = int.__new__(cls, raw & 0xFFFF)
        o.corrupted = bool(raw & 0x10000)
        return o

    def __bool__(self):
        return int(self) != OK


class _Opts(C.Structure):
    _fields_ = [("nparallel", C.c_int32), ("pending_sz", C.c_int32), ("level", C.c_int32), ("has_content_sz", C.c_int32),
                ("content_sz", C.c_uint64), ("read_offset", C.c_int64),
                ("block_checksum", C.c_int32), ("block_linked", C.c_int32), ("content_checksum", C.c_int32),
                ("skip_content_sz", C.c_int32), ("has_dict_id", C.c_int32), ("dict_id", C.c_uint32),
                ("block_size_idx", C.c_int32), ("gpu_batch", C.c_int32), ("fail_after_writes", C.c_int32),
                ("fail_read_at", C.c_int64), ("has_dictionary", C.c_int32), ("dictionary", C.c_void_p),
                ("dictionary_len", C.c_int64)]


def make_opts(parallel=1, pending_size=0, level=1, content_size=None, read_offset=0, block_checksum=False,
              block_linked=False, content_checksum=True, content_size_check=True, dictionary_id=None,
              block_size=BlockIdx4MB, gpu_batch=0, fail_after_writes=-1, fail_read_at=-1, dictionary=None):
    """Defaults == parseOpts (plz4_opts.go:238-255).  Keyword names follow the With* options."""
    o = _Opts()
    o.nparallel, o.pending_sz, o.level = parallel, pending_size, level
    o.has_content_sz, o.content_sz = int(content_size is not None), int(content_size or 0)
    o.read_offset = read_offset
    o.block_checksum, o.block_linked, o.content_checksum = int(block_checksum), int(block_linked), int(content_checksum)
    o.skip_content_sz = int(not content_size_check)
    o.has_dict_id, o.dict_id = int(dictionary_id is not None), int(dictionary_id or 0)
    o.block_size_idx, o.gpu_batch = block_size, gpu_batch
    o.fail_after_writes, o.fail_read_at = fail_after_writes, fail_read_at
    o.has_dictionary = int(dictionary is not None)
    if dictionary is not None:
        d = np.frombuffer(bytes(dictionary), dtype=np.uint8).copy() if not isinstance(dictionary, np.ndarray) else np.ascontiguousarray(dictionary)
        o._dict_keep = d                                     # keep the bytes alive as long as the options
        o.dictionary = d.ctypes.data if d.size else None
        o.dictionary_len = d.size
    return o


_L = None


def lib():
    global _L
    if _L is not None:
        return _L
    L = _native.load()
    vp, u8p, szp = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
    L.plz4h_engine_hip.restype = vp; L.plz4h_engine_hip.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.plz4h_engine_vtable.restype = vp; L.plz4h_engine_vtable.argtypes = [vp]
    L.plz4h_engine_free.argtypes = [vp]
    L.plz4h_writer_new.restype = vp; L.plz4h_writer_new.argtypes = [vp, C.POINTER(_Opts)]
    L.plz4h_writer_write.argtypes = [vp, vp, C.c_size_t, szp]
    L.plz4h_writer_flush.argtypes = [vp]; L.plz4h_writer_close.argtypes = [vp]
    L.plz4h_writer_read_from.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int64, C.POINTER(C.c_int64)]
    L.plz4h_writer_output.restype = C.c_size_t; L.plz4h_writer_output.argtypes = [vp, C.POINTER(vp)]
    L.plz4h_writer_progress.restype = C.c_size_t; L.plz4h_writer_progress.argtypes = [vp, C.POINTER(vp)]
    L.plz4h_writer_free.argtypes = [vp]
    L.plz4h_reader_new.restype = vp; L.plz4h_reader_new.argtypes = [vp, C.POINTER(_Opts), vp, C.c_size_t]
    L.plz4h_reader_read.argtypes = [vp, vp, C.c_size_t, szp]
    L.plz4h_reader_write_to.argtypes = [vp, C.POINTER(C.c_int64)]
    L.plz4h_reader_output.restype = C.c_size_t; L.plz4h_reader_output.argtypes = [vp, C.POINTER(vp)]
    L.plz4h_reader_progress.restype = C.c_size_t; L.plz4h_reader_progress.argtypes = [vp, C.POINTER(vp)]
    L.plz4h_reader_close.argtypes = [vp]; L.plz4h_reader_free.argtypes = [vp]
    L.plz4h_compress_block_bound.argtypes = [C.c_int]
    L.plz4h_compress_block.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int, szp, vp, C.c_int64]
    L.plz4h_decompress_block.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, szp, vp, C.c_int64]
    L.plz4h_write_header.argtypes = [C.POINTER(_Opts), vp]
    L.plz4h_xxh32.restype = C.c_uint32; L.plz4h_xxh32.argtypes = [vp, C.c_size_t]
    L.plz4h_error_string.restype = C.c_char_p; L.plz4h_error_string.argtypes = [C.c_int]
    _L = L
    return L


def _buf(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b)
    return a, (a.ctypes.data if a.size else None)


def _pairs(ptr, n):
    if not n:
        return []
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(n,)).copy()
    return [(int(a[i]), int(a[i + 1])) for i in range(0, n, 2)]


class Engine:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("plz4h: no block engine (is an MI355X visible / the library built?)")
        self.h = handle
        self._children = weakref.WeakSet()      # writers / readers still alive: they hold engine resources (dictionary contexts)

    def close(self):
        if self.h:
            for c in list(self._children):      # free them first: their destructors call back into the engine
                c._free()
            lib().plz4h_engine_free(self.h); self.h = None


def hip_engine(device=0) -> Engine:
    rc = C.c_int(0)
    h = lib().plz4h_engine_hip(device, C.byref(rc))
    if not h:
        raise _native.EngineError(rc.value, "plz4h_engine_hip(device=%d)" % device)
    return Engine(h)


def vtable_engine(vtable_ptr) -> Engine:
    return Engine(lib().plz4h_engine_vtable(vtable_ptr))


class Writer:
    """plz4.NewWriter(wr, opts...) with an in-memory io.Writer (`output()`), plz4_writer.go:40-53."""

    def __init__(self, engine: Engine, **kw):
        self.L = lib(); self.o = make_opts(**kw)
        self.h = self.L.plz4h_writer_new(engine.h, C.byref(self.o))
        engine._children.add(self)

    def write(self, data):
        a, p = _buf(data); n = C.c_size_t(0)
        e = self.L.plz4h_writer_write(self.h, p, a.size, C.byref(n))
        return n.value, Err(e)

    def read_from(self, data, chunk=0, fail_read_at=-1):
        a, p = _buf(data); n = C.c_int64(0)
        e = self.L.plz4h_writer_read_from(self.h, p, a.size, chunk, fail_read_at, C.byref(n))
        return n.value, Err(e)

    def flush(self):
        return Err(self.L.plz4h_writer_flush(self.h))

    def close(self):
        return Err(self.L.plz4h_writer_close(self.h))

    def output(self) -> bytes:
        p = C.c_void_p(); n = self.L.plz4h_writer_output(self.h, C.byref(p))
        return bytes((C.c_ubyte * n).from_address(p.value)) if n else b""       # (string_at stops at 2 GiB)

    def progress(self):
        p = C.c_void_p(); n = self.L.plz4h_writer_progress(self.h, C.byref(p))
        return _pairs(p, n)

    def _free(self):
        if self.h:
            self.L.plz4h_writer_free(self.h); self.h = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


class Reader:
    """plz4.NewReader(rd, opts...) over an in-memory io.Reader, plz4_reader.go:28-33."""

    def __init__(self, engine: Engine, data, **kw):
        self.L = lib(); self.o = make_opts(**kw)
        a, p = _buf(data)
        self.h = self.L.plz4h_reader_new(engine.h, C.byref(self.o), p, a.size)
        engine._children.add(self)

    def read(self, n):
        buf = np.empty(max(n, 1), dtype=np.uint8); got = C.c_size_t(0)
        e = self.L.plz4h_reader_read(self.h, buf.ctypes.data, n, C.byref(got))
        return buf[:got.value].tobytes(), Err(e)

    def write_to(self):
        n = C.c_int64(0)
        e = self.L.plz4h_reader_write_to(self.h, C.byref(n))
        p = C.c_void_p(); k = self.L.plz4h_reader_output(self.h, C.byref(p))
        return n.value, (bytes((C.c_ubyte * k).from_address(p.value)) if k else b""), Err(e)

    def progress(self):
        p = C.c_void_p(); n = self.L.plz4h_reader_progress(self.h, C.byref(p))
        return _pairs(p, n)

    def close(self):
        return Err(self.L.plz4h_reader_close(self.h))

    def _free(self):
        if self.h:
            self.L.plz4h_reader_free(self.h); self.h = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def compress_block_bound(n: int) -> int:
    return int(lib().plz4h_compress_block_bound(n))


def _dict_args(dictionary):
    if dictionary is None:
        return None, None, -1
    d, dp = _buf(dictionary)
    return d, dp, d.size


def compress_block(engine: Engine, src, level=1, dst_cap=None, dictionary=None):
    """plz4.CompressBlock (plz4_block.go:96-119); dst_cap=None == no WithBlockDst; dictionary == WithBlockDictionary."""
    a, p = _buf(src)
    dk, dp, dl = _dict_args(dictionary)
    cap = compress_block_bound(a.size) if dst_cap is None else dst_cap
    out = np.empty(max(cap, 1), dtype=np.uint8); n = C.c_size_t(0)
    e = Err(lib().plz4h_compress_block(engine.h, p, a.size, level, out.ctypes.data, cap, int(dst_cap is not None), C.byref(n), dp, dl))
    return (out[:n.value].tobytes() if not e else None), e


def decompress_block(engine: Engine, src, dst_cap=None, max_out=1 << 27, dictionary=None):
    """plz4.DecompressBlock (plz4_block.go:125-172); dst_cap=None == no WithBlockDst (4x, doubling, 3 tries)."""
    a, p = _buf(src)
    dk, dp, dl = _dict_args(dictionary)
    cap = max_out if dst_cap is None else dst_cap
    out = np.empty(max(cap, 1), dtype=np.uint8); n = C.c_size_t(0)
    e = Err(lib().plz4h_decompress_block(engine.h, p, a.size, out.ctypes.data, cap, int(dst_cap is not None), C.byref(n), dp, dl))
    return (out[:n.value].tobytes() if not e else None), e


def write_header(**kw) -> bytes:
    o = make_opts(**kw); out = (C.c_uint8 * 19)()
    n = lib().plz4h_write_header(C.byref(o), out)
    return bytes(out[:n])
