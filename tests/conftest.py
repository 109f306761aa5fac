import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# torch first, in every session.  The wheel ships its own libamdhip64.so and librccl.so; whichever copy of a library is mapped
# first serves the whole process (same SONAME), and several test modules import torch while they are collected.  A partial run
# (one test file) that opened RCCL through libpanacus_hip.so -- ROCm's copy -- and imported torch only later ended up with TWO
# copies of RCCL in the process and died in a double free when it exited.  With torch mapped before anything else the state is
# the one bench.py runs in, whatever subset of the tests is selected.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
