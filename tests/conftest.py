import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The library is built in-tree (git-ignored); build it if this checkout has none, or a stale one
    (hipcc cross-compiles gfx950 without a GPU).  A failed build is reported by the tests that load it."""
    try:
        from manner_amd.build import build_library
        build_library(verbose=False)
    except Exception as e:          # noqa: BLE001
        print(f"[conftest] libmanner_hip.so build skipped/failed: {e}")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
