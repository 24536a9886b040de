"""Seeded inputs shared by the oracle tests and the GPU parity tests."""
from __future__ import annotations

import numpy as np

from plz4_amd import synth


def structured(n: int, seed: int) -> np.ndarray:
    """Random mixture of literal noise, short/long copies at assorted distances and byte runs --
    built to hit every branch of the L1 parser (long literal runs, 0xFF length chains, offset-1 RLE,
    far/near matches, matches reaching the end-of-block limits)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    alpha = int(rng.integers(2, 257))
    while pos < n:
        kind = rng.integers(0, 10)
        if kind < 3 or pos < 8:
            ln = int(min(n - pos, rng.integers(1, 40 if kind else 700)))
            out[pos:pos + ln] = rng.integers(0, alpha, size=ln, dtype=np.uint8)
        elif kind < 8:
            dist = int(rng.integers(1, min(pos, 70000) + 1))
            ln = int(min(n - pos, rng.integers(4, 30 if kind < 6 else 5000)))
            for i in range(0, ln, max(dist, 1)):          # overlapped copy semantics
                c = min(dist, ln - i)
                out[pos + i:pos + i + c] = out[pos + i - dist:pos + i - dist + c]
        else:
            ln = int(min(n - pos, rng.integers(1, 3000)))
            out[pos:pos + ln] = rng.integers(0, 256)
        pos += ln
    return out


def small_cases():
    """Edge sizes around the parser's thresholds."""
    cases = []
    for n in list(range(0, 40)) + [63, 64, 65, 255, 256, 270, 271, 300, 4095, 4096, 4097]:
        cases.append(("T%d" % n, synth.text(max(n, 1), seed=n + 1)[:n]))
        cases.append(("Z%d" % n, np.zeros(n, dtype=np.uint8)))
        cases.append(("R%d" % n, synth.random_bytes(max(n, 1), seed=n + 7)[:n]))
        cases.append(("S%d" % n, structured(max(n, 1), seed=n + 3)[:n]))
    return cases


def block_cases_64k():
    """Sizes straddling the byU16/byU32 switch at 65547 (lz4.c:710,1389)."""
    cases = []
    for n in (65535, 65536, 65546, 65547, 65548, 65536 + 12, 131072, 200000):
        cases.append(("T%d" % n, synth.text(n, seed=n)))
        cases.append(("S%d" % n, structured(n, seed=n)))
        cases.append(("Z%d" % n, np.zeros(n, dtype=np.uint8)))
    return cases


def twin_cases():
    """Inputs that force same-slot hash collisions inside one 64-position batch (the 'twin' paths of the grid
    encoder): short periods, tiny alphabets, and text with periodic islands."""
    rng = np.random.Generator(np.random.PCG64(77))
    cases = []
    for period in (1, 2, 3, 5, 7, 13, 31, 63, 64, 65, 127):
        unit = rng.integers(0, 256, size=period, dtype=np.uint8)
        cases.append(("P%d" % period, np.tile(unit, 200000 // period + 1)[:200000]))
    for alpha in (2, 3, 4, 8):
        cases.append(("A%d" % alpha, rng.integers(0, alpha, size=150000, dtype=np.uint8)))
    t = synth.text(300000, seed=5).copy()
    for k in range(40):
        off = int(rng.integers(1000, 290000)); per = int(rng.integers(1, 40)); ln = int(rng.integers(20, 3000))
        unit = t[off:off + per].copy()
        t[off:off + ln] = np.tile(unit, ln // per + 1)[:ln]
    cases.append(("Tislands", t))
    return cases
