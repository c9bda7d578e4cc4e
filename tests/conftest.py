import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU; run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from cvids_amd import capi
    capi.build_library()
    return capi.load_library()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
