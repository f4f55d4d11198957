import os as _os

# More hardware queues than HIP's default of 4 (which the GPU box also exports): the library runs up to nine streams per
# frame.  Must be in the environment before the HIP runtime starts, i.e. before torch touches the GPU.
if int(_os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with oracle/Makefile (gcc only)."""
    from oracle import oracle
    oracle.lib()
    return oracle
