import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (only present in the build container)")


@pytest.fixture(scope="session")
def hxlib():
    """The in-tree HIP extension.  GPU tests must fail loudly when it is missing -- never fall back."""
    from isaac_amd import capi
    lib = capi.lib()
    assert lib.hx_device_count() > 0, "no HIP device visible"
    return lib
