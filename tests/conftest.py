import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """One native context for the whole GPU session (fails loudly without a device)."""
    from edrgp_amd.backend import HipGP
    h = HipGP(0)
    yield h
    h.close()
