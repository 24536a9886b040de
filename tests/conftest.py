import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from orclib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (liblz4 v1.10.0 from /root/reference, built by `make -C oracle ref`)."""
    from orclib import Ref, REF_SO
    if not os.path.exists(REF_SO):
        if os.path.exists("/root/reference/internal/pkg/clz4/lz4.c"):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/liblz4ref.so not built and /root/reference absent")
    return Ref()
