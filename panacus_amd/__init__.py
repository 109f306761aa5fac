"""panacus_amd -- MI355X-native coverage-histogram / pangenome-growth engine.

The hot path of marschall-lab/panacus (hist / growth / histgrowth / ordered-histgrowth),
rebuilt as hand-written HIP kernels for gfx950 behind the C ABI of include/panacus_amd.h.
The Python layer is a thin ctypes binding plus the host-side pieces that sit above the ABI.
There is no CPU fallback: without libpanacus_hip.so or without a GPU every call raises.
"""
from . import capi  # noqa: F401
from .capi import Context, PnxError  # noqa: F401
from .thresholds import Threshold, ThresholdContainer, coverage_abs, quorum_table  # noqa: F401

__version__ = "0.5.0"
