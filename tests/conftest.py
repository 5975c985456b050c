import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _restore_precision():
    """Every test starts in the default 'bf16' mode and leaves it that way: a test that ends (or fails) inside a
    precision loop must not hand the opt-in 8-bit stash, or fp32, to whatever runs next in the session."""
    try:
        import hypernerf_torch_amd as HN
    except Exception:           # collection on a box without the package's dependencies
        yield
        return
    before = HN.get_precision() if hasattr(HN, "get_precision") else "bf16"
    yield
    if hasattr(HN, "set_precision"):
        HN.set_precision(before)
