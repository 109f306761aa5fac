"""The libraries bound in-process the way a host does it, without Python in the loop: tools/soak_pnx.cpp (the device ABI: context
after context on several threads, every histogram against a serial count) and tools/soak_cli.cpp (pnh_run_cli: the commands of the
CLI tests in a loop on several threads, every table against the first iteration's).  Short runs here; tools/soak_gpu.sh runs them
for thousands of contexts and commands, plain and under AddressSanitizer / UBSan / ThreadSanitizer builds (DESIGN.md section 2:
the hunt of round 5's in-process abort)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "panacus_amd")


def _run(exe, args, timeout=600):
    path = os.path.join(PKG, exe)
    if not os.path.exists(path):
        pytest.skip(f"{path} is not built (python -m panacus_amd._build)")
    return subprocess.run([path] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.gpu
def test_contexts_on_four_threads_through_the_abi_alone():
    """240 contexts per thread on 4 threads, up to 3 alive per thread: init -> set_csr -> set_order -> hist (-> second sweep, ordered
    growth, intersections) -> free; every histogram equals the serial count of abacus.rs:719-787"""
    r = _run("soak_pnx", [240, 4, 3])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"960 contexts, 0 failures" in r.stderr


@pytest.mark.gpu
def test_cli_commands_in_process_on_three_threads(tmp_path, golden_dir):
    """3 threads x 3 iterations x 47 commands (BED subset / exclude lists, node / bp / edge, ordered growth, similarity, table,
    the .pcsr cache, growth from a hist TSV), each command with a GPU context of its own in ONE process: every table repeats"""
    r = _run("soak_cli", [tmp_path / "w", 3, 3, golden_dir])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b", 0 failures" in r.stderr.split(b"soak_cli:")[-1]
