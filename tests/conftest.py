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


_MEASURED = {}


@pytest.fixture
def measured(request):
    """record(key=value, ...): keeps the MEASURED side of a kernel-vs-kernel tolerance next to its bound.  The values
    of a session are written to gpurun_out/measured_tolerances.json (scratch that travels back from the GPU box; the copy that is
    judged lives under profiles/)."""
    def record(**values):
        def plain(v):
            try:
                return float(v)                 # python and numpy scalars alike
            except (TypeError, ValueError):
                return str(v)
        _MEASURED.setdefault(request.node.name, {}).update({k: plain(v) for k, v in values.items()})
    return record


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "measured_tolerances.json")
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_MEASURED)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
