import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _minimiser_groups_on_small_inputs(monkeypatch):
    """The library builds the minimiser groups (the fast kernel's second access path) only for big read sets probed by one context
    (dev_build_index); the tests' inputs are small, and the groups are half of the kernel's look-up code: forced on here unless the
    environment or the test sets the switch itself (test_without_minimiser_groups_matches_oracle covers the other half)."""
    if "SAGE2OV_MINIMIZER_INDEX" not in os.environ:
        monkeypatch.setenv("SAGE2OV_MINIMIZER_INDEX", "1")
