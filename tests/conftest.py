import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# PyTorch's ROCm wheel bundles its own HIP runtime under the same soname as /opt/rocm's, which the product library
# links.  Whichever is loaded first serves both; torch only finds its GPUs with its own copy, so it goes first
# (bench.py imports torch before the engine for the same reason).
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
