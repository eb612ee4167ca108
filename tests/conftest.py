import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger cases; GPU tests marked slow run only under -m \"gpu and slow\" (or SSAMD_RUN_SLOW=1)")


def pytest_collection_modifyitems(config, items):
    """`pytest -m gpu` is what the driver runs at round end, under a time limit: the heaviest GPU cases (the 200^3 / 160^3
    RCCL self tests, the 64^3 and 100^3 eight-peer runs, the 16 500-row dense fronts) are marked slow as well and are
    deselected unless the mark expression names `slow` (`-m "gpu and slow"`) or SSAMD_RUN_SLOW=1.  One 200^3 property
    test and one eight-peer oracle case stay in the default set."""
    expr = config.getoption("markexpr", "") or ""
    if "slow" in expr or os.environ.get("SSAMD_RUN_SLOW", "0") not in ("", "0"):
        return
    keep, drop = [], []
    for it in items:
        (drop if (it.get_closest_marker("gpu") and it.get_closest_marker("slow")) else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
