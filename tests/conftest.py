import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Shapes the library was not built with are specialised at setup (csrc/jit.cpp: a one-off hipcc run per shape).  The suite pins
# which kernel takes which shape, and uses dozens of odd shapes to reach the run-time-shape kernels: off here, on in the tests
# of the specialisation itself (tests/test_jit.py).
os.environ.setdefault("TINYMPC_HIP_NO_JIT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    """The CPU checkers (test infrastructure).  Builds the C restatement if needed."""
    from oracle import cpu_oracle
    if not os.path.isfile(cpu_oracle.PORT_LIB):
        cpu_oracle.build(port=True, ref=False)
    return cpu_oracle


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built by __graft_entry__.build(), rebuilt here when it is missing or stale (built from other
    sources than the ones present).  GPU tests fail loudly without it."""
    import tinympc_julia_amd as t
    t.ensure_built()
    return t.load_library()
