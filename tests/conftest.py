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
    from backends import Backend
    b = Backend("orc")
    b.set_threads(min(8, os.cpu_count() or 1))
    return b


@pytest.fixture(scope="session")
def ref():
    from backends import Backend, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libcgrt_ref.so not built (needs /root/reference)")
    return Backend("ref")


@pytest.fixture(scope="session")
def gpu_ready():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU")
    return True
