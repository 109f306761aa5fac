"""The host's worker pool under oversubscription: a worker that wakes up late for a finished job must not run (or count)
a task of the next one.  Before round 4's fix 1 job in ~13,000 ran a task twice with 48 threads on 8 cores -- in the CLI: a
table row missing, a crash, once in ~50,000 commands of the fuzz soak."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_task_runs_exactly_once(tmp_path):
    exe = str(tmp_path / "stress")
    host = os.path.join(ROOT, "panacus_amd", "host")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", host, os.path.join(ROOT, "tests", "thread_pool_stress.cpp"),
                    os.path.join(host, "thread_pool.cpp"), "-o", exe], check=True)
    for threads in ("6", "48"):
        r = subprocess.run([exe, "150000"], env=dict(os.environ, PANACUS_AMD_THREADS=threads), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (threads, r.stdout, r.stderr)
