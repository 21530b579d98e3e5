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
