"""Diagnostics: scripts/host_rate.py with another build of the library (PLZ4HIP_LIB=path)."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plz4_amd import _native
if os.environ.get("PLZ4HIP_LIB"):
    _native.LIB_PATH = os.environ["PLZ4HIP_LIB"]
sys.argv = [os.path.join(ROOT, "scripts", "host_rate.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
