import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "isca-2025-lia_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lia_oracle
    lia_oracle.lib()
    return lia_oracle


@pytest.fixture(autouse=True)
def _no_state_outside_the_test(tmp_path, monkeypatch):
    """the cooperative controller's converged counts (scheduler.CoopStore) and the box calibration go where LIA_STATE_DIR points:
    every test gets its own empty directory, so no test depends on an earlier run and none writes into the user's home"""
    monkeypatch.setenv("LIA_STATE_DIR", str(tmp_path / "lia_state"))
