"""Builds the native parts of panacus_amd in-tree with hipcc / g++ (no JIT cache).

  libpanacus_hip.so   HIP kernels + the C ABI of include/panacus_amd.h   (gfx950)
  libpanacus_hip_steps.so   the step routes of round 2 (kernels_cover.hip, kernels_runs.hip): a CROSS-CHECK module the tests
                      hold the product against; the product library opens it only when PNX_CFG_COVER_VARIANT asks for one
  libpanacus_host.so  C++ host layer (GFA front end, closed-form growth, TSV writers)
  panacus-amd         CLI (hist | growth | histgrowth | ordered-histgrowth)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build
container; the .so files travel to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOSTSRC = os.path.join(HERE, "host")
LIB_HIP = os.path.join(HERE, "libpanacus_hip.so")
LIB_STEPS = os.path.join(HERE, "libpanacus_hip_steps.so")
LIB_HOST = os.path.join(HERE, "libpanacus_host.so")
CLI = os.path.join(HERE, "panacus-amd")

HIP_SOURCES = ["pnx_api.hip", "pnx_comm.hip", "pass_pipeline.hip", "upload_scan.hip", "kernels_hist.hip", "kernels_rows.hip", "kernels_band.hip", "kernels_gfa.hip", "kernels_relabel.hip", "kernels_cut.hip", "kernels_growth.hip", "kernels_pairs.hip", "kernels_pairs_mfma.hip", "kernels_closed_form.hip", "pansyn.hip"]
STEP_SOURCES = ["kernels_cover.hip", "kernels_runs.hip"]  # the cross-check module
HOST_SOURCES = ["thread_pool.cpp", "growth_closed_form.cpp", "gfa_graph.cpp", "tables.cpp", "synth_gfa.cpp", "linkage.cpp", "mini_yaml.cpp", "report.cpp", "commands.cpp", "host_api.cpp"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: panacus_amd needs the ROCm toolchain (no CPU fallback exists)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_hip(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, "pnx_context.hpp"), os.path.join(CSRC, "step_chunks.hpp"), os.path.join(ROOT, "include", "panacus_amd.h"),
            os.path.join(CSRC, "tile_counters.hpp"), os.path.join(CSRC, "name_table.hpp"), os.path.join(CSRC, "exp2_exact.hpp"), os.path.join(CSRC, "exp2_table.inc"),
            os.path.join(CSRC, "log2_exact.hpp"), os.path.join(CSRC, "log2_table.inc")]
    objs, step_objs = [], []
    procs = []
    for src in HIP_SOURCES + STEP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        (step_objs if src in STEP_SOURCES else objs).append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o,
                   "-Wno-unused-result"]
            if src == "kernels_closed_form.hip":
                cmd.append("-ffp-contract=off")  # the restated exp2 must not be fused into FMAs
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("build failed: " + " ".join(cmd))
        if verbose and out.strip():
            print(out)
    if force or procs or _newer(LIB_HIP, objs):
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs + ["-ldl"])
    if force or procs or _newer(LIB_STEPS, step_objs + [LIB_HIP]):
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_STEPS] + step_objs +
             ["-L" + HERE, "-lpanacus_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB_HIP


def build_host(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(HOSTSRC, s) for s in HOST_SOURCES]
    hdrs = [os.path.join(HOSTSRC, h) for h in os.listdir(HOSTSRC) if h.endswith(".hpp")] if os.path.isdir(HOSTSRC) else []
    hdrs.append(os.path.join(ROOT, "include", "panacus_amd.h"))
    if not all(os.path.exists(s) for s in srcs):
        return ""
    if force or _newer(LIB_HOST, srcs + hdrs + [LIB_HIP]):
        # -ffp-contract=off: the closed-form growth must keep the reference's f64 operation order
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fno-fast-math",
               "-Wall", "-o", LIB_HOST] + srcs + ["-L" + HERE, "-lpanacus_hip", "-Wl,-rpath,$ORIGIN", "-lz", "-lm"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    main = os.path.join(HOSTSRC, "cli_main.cpp")
    if os.path.exists(main) and (force or _newer(CLI, [main, LIB_HOST, LIB_HIP] + hdrs)):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-o", CLI, main, "-L" + HERE, "-lpanacus_host", "-lpanacus_hip",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + HERE, "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    return LIB_HOST


def build_soak(force: bool = False, verbose: bool = False):
    """The two soak harnesses (tools/soak_pnx.cpp: the device ABI alone; tools/soak_cli.cpp: the in-process CLI), no Python in
    them: contexts and commands in loops on several threads -- how a host binds the libraries in-process.  tests/test_gpu_soak.py
    runs both briefly in every GPU session; tools/soak_gpu.sh runs them long, plain and under sanitizers."""
    tools = os.path.join(ROOT, "tools")
    for name, libs in (("soak_pnx", ["-lpanacus_hip"]), ("soak_cli", ["-lpanacus_host", "-lpanacus_hip"])):
        src, exe = os.path.join(tools, name + ".cpp"), os.path.join(HERE, name)
        if not os.path.exists(src):
            continue
        if force or _newer(exe, [src, LIB_HIP, LIB_HOST, os.path.join(ROOT, "include", "panacus_amd.h")]):
            cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", exe, src, "-L" + HERE] + libs + \
                  ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + HERE, "-Wl,-rpath-link,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd))
            _run(cmd)


def build_all(force: bool = False, verbose: bool = False):
    build_hip(force, verbose)
    build_host(force, verbose)
    build_soak(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", LIB_HIP, os.path.exists(LIB_HIP), LIB_HOST, os.path.exists(LIB_HOST))
