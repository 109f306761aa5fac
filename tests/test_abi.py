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
