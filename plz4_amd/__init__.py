"""plz4_amd -- MI355X (gfx950) LZ4 Frame block engine behind plz4's per-block hot path.

The product is plz4_amd/libplz4hip.so (HIP kernels + the C ABI in include/plz4hip.h).  This package is the thin
Python-side plumbing used by tests and bench.py; importing it does not load the library, constructing an
Engine does, and that fails loudly when the library or the GPU is missing.
"""
__all__ = ["synth"]
