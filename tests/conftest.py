import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/stx_oracle.cpp."""
    from oracle import oracle as O

    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """Process-wide device context; fails loudly when the HIP library or the GPU is missing."""
    import stitching_amd as S

    return S.get_context()
