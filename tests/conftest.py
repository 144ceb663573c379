import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def _ensure_built():
    """The .so files are git-ignored build products: build them once if a fresh checkout lacks them
    (hipcc cross-compiles gfx950 without a GPU; gcc builds the oracle)."""
    import shutil
    import subprocess
    csrc = os.path.join(REPO, "real-time-video-quality-analysis_amd", "csrc")
    if not (os.path.exists(os.path.join(csrc, "libvqa_hip.so")) and os.path.exists(os.path.join(csrc, "lab", "libvqa_hip_lab.so"))):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if os.path.exists(hipcc):
            subprocess.check_call(["make", "-C", csrc, "-j", "8", "HIPCC=" + hipcc, "all", "lab"])
    if not os.path.exists(os.path.join(REPO, "oracle", "libvqa_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])


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
