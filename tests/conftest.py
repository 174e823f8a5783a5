import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    """One mvs_ctx for the whole GPU test session (the C ABI is the only way in)."""
    from mvslam_amd import capi

    c = capi.Context(0)
    yield c
    c.close()
