import os
import sys

import pytest

# torch before anything loads liboalsfx_hip.so: torch ships its own HIP runtime, and the tests that hand torch device buffers to the
# library need both on the same one -- whichever is loaded first serves both, and torch does not find its GPUs through /opt/rocm's.
# (The CPU suite itself needs no torch: without one only the tests that ask for it skip.)
try:
    import torch  # noqa: F401,E402
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The product library and the CPU oracle must exist; build them if the tree is fresh."""
    from oalsfxpp_amd import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build_all()
    from oracle import oracle as orc
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        orc.build(ref=os.path.isdir("/root/reference"))
    yield
