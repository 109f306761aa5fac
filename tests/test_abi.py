"""The C-ABI library loads on a CPU-only box and exports every symbol include/panacus_amd.h
declares; without a GPU the entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from panacus_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "panacus_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pnx_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    L = capi.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/panacus_amd.h but not exported"
    assert sorted(capi.ABI_SYMBOLS) == names


def test_header_is_plain_c(tmp_path):
    # the header must be bindable from C (and thus from Rust extern "C"): compile it as C11
    src = tmp_path / "t.c"
    src.write_text('#include "panacus_amd.h"\nint main(void){ pnx_info_t i; (void)i; return PNX_OK; }\n')
    import subprocess
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PnxError) as e:
        capi.Context(0)
    assert e.value.code == capi.PNX_ENODEV


def test_product_never_touches_the_oracle():
    """panacus_amd/ must not import, link or reference the oracle (test infrastructure)."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "panacus_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b|liboracle|panacus_oracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def _plain_c_host(tmp_path):
    """A C host (the shape of the Rust FFI) links against libpanacus_hip.so; without a GPU
    pnx_init returns PNX_ENODEV and a message, it never falls back to a CPU path."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include "panacus_amd.h"
int main(void) {
    pnx_ctx *ctx = NULL;
    int rc = pnx_init(&ctx, 0);
    printf("%d|%s|%s\n", rc, pnx_version(), rc ? pnx_last_error(NULL) : "ok");
    if (rc == PNX_OK) {
        uint32_t items[3] = {1, 2, 2};
        uint64_t off[2] = {0, 3};
        uint32_t pi[1] = {0}, gi[1] = {0}, cnt[3];
        uint64_t hist[2];
        if (pnx_set_csr(ctx, items, off, 1, 2, NULL, NULL) || pnx_set_order(ctx, pi, gi, 1, 1) ||
            pnx_hist(ctx, cnt, hist)) { printf("ERR %s\n", pnx_last_error(ctx)); return 2; }
        printf("hist %llu %llu cnt %u %u\n", (unsigned long long)hist[0], (unsigned long long)hist[1], cnt[1], cnt[2]);
        pnx_free(ctx);
    }
    return 0;
}
''')
    exe = tmp_path / "host"
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lpanacus_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.decode()
    return int(out.split("|")[0]), out


def test_plain_c_program_links_and_fails_loudly(tmp_path):
    """without a GPU: PNX_ENODEV and a message -- never a CPU path (on a GPU box the same host computes, see the gpu test)"""
    import torch
    rc, out = _plain_c_host(tmp_path)
    if torch.cuda.is_available():
        assert rc == 0 and "hist 0 2 cnt 1 1" in out
    else:
        assert rc == capi.PNX_ENODEV and "no CPU fallback" in out


@pytest.mark.gpu
def test_plain_c_program_computes_on_the_gpu(tmp_path):
    """the same C host on a GPU box: upload, order, histogram through nothing but the C ABI"""
    rc, out = _plain_c_host(tmp_path)
    assert rc == 0 and "hist 0 2 cnt 1 1" in out, out



def test_step_routes_are_a_module_beside_the_product_library():
    """round 2's step routes (kernels_cover.hip, kernels_runs.hip) are cross-check code: built into libpanacus_hip_steps.so,
    which the product library opens only when PNX_CFG_COVER_VARIANT asks for a step route -- its own symbols hold none of them"""
    import os
    import subprocess
    from panacus_amd import _build
    _build.build_hip()
    prod = subprocess.run(["nm", "-D", "--defined-only", _build.LIB_HIP], capture_output=True, text=True, check=True).stdout
    mod = subprocess.run(["nm", "-D", "--defined-only", _build.LIB_STEPS], capture_output=True, text=True, check=True).stdout
    for sym in ("k_tile_cover", "k_tile_index", "k_runs_count", "k_pack12", "prepare_steps", "build_run_index"):
        assert sym not in prod, sym
        assert sym in mod, sym
    assert "pnx_step_routes_table" in mod and "pnx_step_routes_table" not in prod
    assert "k_rows_cover" in prod and "k_band_cover" in prod
    # (15.5 MB while the step routes were inside; 12.0 MB without them, 12.6 MB with k_quorum_fused and k_cf_eval_small, 15.0 MB
    # with rocPRIM's radix sort of u32 keys: the sorted copies of shuffled paths at upload, upload_scan.hip; 17.2 MB with the
    # two-ranks-per-step instances of k_growth_fused -- the symbols above are what keeps the step routes out)
    assert os.path.getsize(_build.LIB_HIP) < 18 << 20
