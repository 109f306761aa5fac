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


# The CLI tests run their commands IN this process (hostlib.run_cli -> pnh_run_cli): the way a host binds the library, hundreds of
# commands with a GPU context each in one pytest process.  At the end of round 5 three of eight whole sessions died of SIGABRT
# inside such tests and the suite ran every command as the `panacus-amd` binary instead; round 6 hunted the abort (guard-page
# device allocator, AddressSanitizer / UBSan / ThreadSanitizer builds of both libraries under soak harnesses, 15 whole in-process
# sessions -- 5 of them on round 5's own tree -- without one abort: DESIGN.md section 2) and made in-process the default again.
# PNX_TEST_CLI_OWN_PROCESS=1 runs every command as the binary in a process of its own (a command that dies then fails ONE test,
# with its stderr in the report); a process that dies in native code leaves its C stack and the tail of the captured stderr in
# PANACUS_AMD_CRASH_LOG (above).
def _run_cli_in_a_process(args):
    import subprocess

    exe = os.path.join(ROOT, "panacus_amd", "panacus-amd")
    # (argv[0] as the in-process runner passes it: the tables' header line echoes the command)
    p = subprocess.run(["panacus-amd"] + [str(a) for a in args], executable=exe, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    if p.returncode < 0:
        raise RuntimeError("panacus-amd %s died with signal %d\n%s" % (" ".join(map(str, args)), -p.returncode,
                                                                       p.stderr.decode(errors="replace")[-4000:]))
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.fixture(scope="session", autouse=True)
def _cli_commands_in_their_own_process():
    if not os.environ.get("PNX_TEST_CLI_OWN_PROCESS"):
        yield
        return
    exe = os.path.join(ROOT, "panacus_amd", "panacus-amd")
    if not os.path.exists(exe):
        yield
        return
    from panacus_amd import hostlib

    saved = hostlib.run_cli
    hostlib.run_cli = _run_cli_in_a_process
    try:
        yield
    finally:
        hostlib.run_cli = saved
