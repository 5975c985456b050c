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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Every skipped test with its reason, as the LAST lines of the run (a `-q` run prints 'N skipped' and nothing else;
    the two-rank RCCL test self-enables on a box with two GPUs and must not disappear silently on a one-GPU lease)."""
    skipped = terminalreporter.stats.get("skipped", [])
    if not skipped:
        return
    try:
        import torch
        n_dev = torch.cuda.device_count()
    except Exception:
        n_dev = 0
    terminalreporter.write_line("")
    for rep in skipped:
        reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) and len(rep.longrepr) == 3 else str(rep.longrepr)
        name = rep.nodeid.split("::")[-1]
        extra = f" [{n_dev} device(s) visible]" if "rccl" in name or "GPU" in reason else ""
        terminalreporter.write_line(f"SKIPPED {rep.nodeid}: {reason}{extra}")
    if any("test_two_rank_rccl_step" in rep.nodeid for rep in skipped):
        terminalreporter.write_line(f"N > 1 over RCCL was NOT exercised in this run: two-rank RCCL test skipped, {n_dev} device(s) visible "
                                    "(covered instead by the 2-rank gloo tests and the one-rank RCCL graph)")
