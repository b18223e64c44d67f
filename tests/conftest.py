import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "tdoa-geolocation_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch ships its own HIP runtime; whichever runtime a process loads first is the one it keeps.  Tests
# that hand torch tensors to the library (capture_attach_device) need torch's runtime to be that one,
# exactly as in bench.py where torch is imported before the library is loaded.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for everything else
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
