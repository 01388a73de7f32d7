import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
OGG_FILES = ["1test", "2test", "3test", "issue6test"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/, test infrastructure) behind ctypes; built with gcc on first use."""
    from tests import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def ogg_bytes():
    return {n: open(os.path.join(GOLDEN, n + ".ogg"), "rb").read() for n in OGG_FILES}


@pytest.fixture(scope="session")
def gpu_ctx():
    import nvorbis_amd as nv
    ctx = nv.Context(0)
    yield ctx
    ctx.close()
