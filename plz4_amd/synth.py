"""Deterministic synthetic inputs for parity tests and bench.py.

    T  text-like: 4096 pseudo-words (length U[2,12], letters a-z, trailing space) drawn
       Zipf(a=1.2) clipped to the vocabulary, numpy Generator(PCG64(0x504C5A34)).  Headline input.
    R  PCG64 random bytes (incompressible -> every block stored raw).
    Z  zeros (maximum match lengths, 0xFF-run length encoding).
    M  per-block mix: 50 % T / 25 % R / 25 % Z.

The reference corpora named by the reference's tests (webster, dickens, enwik8) are not available
(/root/reference/.MISSING_LARGE_BLOBS); these generators stand in for them and are part of the repo.
"""
from __future__ import annotations

import numpy as np

SEED = 0x504C5A34
VOCAB = 4096
ZIPF_A = 1.2


def _vocab(rng: np.random.Generator):
    lens = rng.integers(2, 13, size=VOCAB)                  # U[2,12] letters
    letters = rng.integers(97, 123, size=int(lens.sum()), dtype=np.uint8)
    wl = lens + 1                                           # + trailing space
    starts = np.concatenate(([0], np.cumsum(wl)[:-1]))
    flat = np.full(int(wl.sum()), 32, dtype=np.uint8)
    src = 0
    # place letters (vectorised): positions of non-space bytes
    pos = np.repeat(starts, lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
    flat[pos] = letters
    del src
    return flat, starts.astype(np.int64), wl.astype(np.int64)


def text(nbytes: int, seed: int = SEED) -> np.ndarray:
    """`nbytes` of T data as a uint8 array."""
    rng = np.random.Generator(np.random.PCG64(seed))
    flat, starts, wl = _vocab(rng)
    out = np.empty(nbytes, dtype=np.uint8)
    filled = 0
    chunk_words = 1 << 20
    while filled < nbytes:
        z = rng.zipf(ZIPF_A, size=chunk_words)
        idx = (np.minimum(z, VOCAB) - 1).astype(np.int64)   # clipped to the vocabulary
        l = wl[idx]
        ends = np.cumsum(l)
        total = int(ends[-1])
        o0 = ends - l
        gather = np.repeat(starts[idx] - o0, l) + np.arange(total, dtype=np.int64)
        piece = flat[gather]
        take = min(total, nbytes - filled)
        out[filled:filled + take] = piece[:take]
        filled += take
    return out


def random_bytes(nbytes: int, seed: int = SEED) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x52))
    return rng.integers(0, 256, size=nbytes, dtype=np.uint8)


def zeros(nbytes: int) -> np.ndarray:
    return np.zeros(nbytes, dtype=np.uint8)


def mixed(nbytes: int, bsz: int, seed: int = SEED) -> np.ndarray:
    """Per-block mix: blocks cycle T,R,T,Z (50 % T / 25 % R / 25 % Z)."""
    out = np.empty(nbytes, dtype=np.uint8)
    nblk = (nbytes + bsz - 1) // bsz
    t = text(((nblk + 1) // 2) * bsz, seed)
    r = random_bytes(((nblk + 3) // 4) * bsz, seed)
    ti = ri = 0
    for b in range(nblk):
        lo, hi = b * bsz, min((b + 1) * bsz, nbytes)
        n = hi - lo
        k = b & 3
        if k in (0, 2):
            out[lo:hi] = t[ti:ti + n]; ti += bsz
        elif k == 1:
            out[lo:hi] = r[ri:ri + n]; ri += bsz
        else:
            out[lo:hi] = 0
    return out


def make(kind: str, nbytes: int, bsz: int = 4 << 20, seed: int = SEED) -> np.ndarray:
    if kind == "T":
        return text(nbytes, seed)
    if kind == "R":
        return random_bytes(nbytes, seed)
    if kind == "Z":
        return zeros(nbytes)
    if kind == "M":
        return mixed(nbytes, bsz, seed)
    raise ValueError(kind)
