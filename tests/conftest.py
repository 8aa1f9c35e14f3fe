import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ptx():
    import torch  # noqa: F401  (before libptx_hip.so: one HIP runtime per process)
    import ptx_import
    return ptx_import.load()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu_ctx(ptx):
    ctx = ptx.Context(0)
    yield ctx
    ctx.close()
