import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The fast probe kernel has two access paths to the index: the uniform table and the minimiser groups.  Which one a context uses is the LIBRARY's decision
# (dev_build_index: by the share of reads without a predecessor in the locality order -- never for the small inputs of a test, and not at BASELINE
# configs[1] / configs[2] either).  SAGE2OV_MINIMIZER_INDEX=0/1 is a test-only override of that decision.  Nothing here is autouse (round 3's conftest forced
# the groups on for every test, so the full-size digest tests never ran the route the library ships): a module that wants the groups' code exercised on its
# small inputs asks for `minimiser_groups_on`, and the tests that must hold on every route take `access_route`.
@pytest.fixture
def minimiser_groups_on(monkeypatch):
    if "SAGE2OV_MINIMIZER_INDEX" not in os.environ:
        monkeypatch.setenv("SAGE2OV_MINIMIZER_INDEX", "1")


@pytest.fixture(params=["library_default", "groups_on", "groups_off"])
def access_route(request, monkeypatch, minimiser_groups_on):      # (after the module-level opt-in, whichever way pytest orders them)
    """the route of the fast kernel's look-ups: what the library picks by itself, the minimiser groups forced on, forced off"""
    monkeypatch.delenv("SAGE2OV_MINIMIZER_INDEX", raising=False)
    if request.param != "library_default":
        monkeypatch.setenv("SAGE2OV_MINIMIZER_INDEX", "1" if request.param == "groups_on" else "0")
    return request.param
