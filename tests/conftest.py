import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# torch's eager convolutions (the fp32 references of the GPU tests, the DDP step) go through MIOpen: its default find mode benchmarks every
# new problem for seconds; FAST takes the heuristic's pick (the product path has no MIOpen call)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
