import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The tools re-execute themselves once per visible GPU (robosat_amd.launch); tests that call a tool's main() in-process
# must stay one process whatever box they run on.  The launcher tests set ROBOSAT_GPUS themselves.
os.environ.setdefault("ROBOSAT_GPUS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """The co-residency screen runs LAST.  Its positive control depends on how the box schedules two HIP streams; should it come out
    INCONCLUSIVE on some box, that must cost the screen's own tests and not -- under the driver's `-x` -- every parity test that would
    have been collected behind it."""
    last = [it for it in items if "test_gpu_race_screen" in it.nodeid]
    if last:
        items[:] = [it for it in items if "test_gpu_race_screen" not in it.nodeid] + last
