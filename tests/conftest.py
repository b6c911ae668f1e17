import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds on CPU")


def pytest_sessionstart(session):
    """The shared libraries are build artefacts (git-ignored): make sure they exist and are current
    before any test loads them — what __graft_entry__.build() does.  hipcc cross-compiles without a GPU."""
    import subprocess
    from panagram_amd import build
    build.build(force=False, verbose=False)
    mk = os.path.join(ROOT, "oracle", "Makefile")
    if os.path.exists(mk):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def ctx():
    """A panagram_amd Context on GPU 0.  GPU tests FAIL (not skip) when the HIP
    library is missing or no device is visible: there is no fallback to hide behind."""
    from panagram_amd import engine
    c = engine.Context(0)
    yield c
    c.close()
