import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def engine():
    """One HIP engine for the whole GPU session (a single process on the card)."""
    import rtvqa_amd
    eng = rtvqa_amd.Engine(0)
    yield eng
    eng.close()
