"""Test helper: an oracle-backed block engine plugged into the C++ host layer through its vtable plug-in point, so the
host logic (framing, flush schedule, in-order emission, error latching) is testable without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from orclib import ROOT

SRC = os.path.join(ROOT, "tests", "hostlib", "oracle_engine.c")
SO = os.path.join(ROOT, "tests", "hostlib", "_build", "liboracle_engine.so")


class _VT(C.Structure):
    _fields_ = [("user", C.c_void_p), ("f", C.c_void_p * 10)]


_keep = []


def oracle_engine():
    from plz4_amd import host
    orc_c = os.path.join(ROOT, "oracle", "plz4_oracle.c")
    newest = max(os.path.getmtime(SRC), os.path.getmtime(orc_c))
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", SO, SRC, orc_c])
    L = C.CDLL(SO)
    vt = _VT()
    L.oracle_engine_vtable(C.byref(vt))
    _keep.append((L, vt))
    return host.vtable_engine(C.byref(vt))
