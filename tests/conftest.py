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


# A process that dies inside the native libraries leaves the C stack of the faulting thread behind (hostlib installs the
# handler when it loads the host library): pytest's fd capture swallows whatever the runtime printed on its way down.
if "PANACUS_AMD_CRASH_LOG" not in os.environ:
    _crash_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(_crash_dir, exist_ok=True)
    except OSError:
        _crash_dir = "/tmp"
    os.environ["PANACUS_AMD_CRASH_LOG"] = os.path.join(_crash_dir, "pytest_crash.txt")


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


# PNX_TEST_RESOURCES=<file>: after every test one line with the process's open descriptors, threads and mappings (a leak that
# only a whole session accumulates shows as a trend; the file is flushed per line, so it survives a process that dies)
_RES = os.environ.get("PNX_TEST_RESOURCES")


def pytest_runtest_teardown(item, nextitem):
    if not _RES:
        return
    try:
        with open("/proc/self/maps") as f:
            n_maps = sum(1 for _ in f)
        line = "%s\tfds=%d\tthreads=%d\tmaps=%d\n" % (item.nodeid, len(os.listdir("/proc/self/fd")),
                                                      len(os.listdir("/proc/self/task")), n_maps)
        with open(_RES, "a") as f:
            f.write(line)
    except OSError:
        pass
