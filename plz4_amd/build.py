"""Build the gfx950 shared library in-tree: plz4_amd/libplz4hip.so (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplz4hip.so")


def sources():
    out = []
    for dp, _, fs in os.walk(CSRC):
        for f in sorted(fs):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(dp, f))
    return out


def deps():
    out = [os.path.join(ROOT, "include", "plz4hip.h")]
    for dp, _, fs in os.walk(CSRC):
        out += [os.path.join(dp, f) for f in fs]
    return out


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -fno-slp-vectorize: the SLP pass pairs up the dwords of freshly loaded windows into 64-bit values, which puts register
    # copies (and therefore a wait for the load) directly behind loads whose latency the kernels mean to overlap
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
