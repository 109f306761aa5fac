"""Coverage / quorum thresholds with the reference's semantics.

Mirrors ``Threshold`` (src/util.rs:328-364) and ``ThresholdContainer::parse_params``
(src/graph_broker/hist.rs:266-322): coverage values are absolute integers, quorum values
are relative floats in [0, 1]; comma lists; a list of length 1 is broadcast.
The integer tables handed to the kernels are computed here in f64 exactly like
``AbacusByGroup::calc_growth`` does (src/graph_broker/abacus.rs:997-998, 1009).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

ABSOLUTE, RELATIVE = 0, 1


@dataclass(frozen=True)
class Threshold:
    kind: int
    value: float

    def to_absolute(self, n: int) -> int:
        # src/util.rs:350-355
        if self.kind == ABSOLUTE:
            return int(self.value)
        return int(math.ceil(float(n) * self.value))

    def to_relative(self, n: int) -> float:
        # src/util.rs:357-362
        if self.kind == RELATIVE:
            return float(self.value)
        return float(int(self.value)) / float(n)

    def get_string(self) -> str:
        # src/util.rs:343-348 -- Rust `{}` formatting of usize / f64
        if self.kind == ABSOLUTE:
            return str(int(self.value))
        return format_f64(self.value)


def format_f64(x: float) -> str:
    """Rust ``{}`` for f64: shortest round-trip, never scientific, integers without '.0'."""
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    r = repr(float(x))
    if "e" in r or "E" in r:
        from decimal import Decimal
        r = format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


class ThresholdContainer:
    """hist.rs:261-322"""

    def __init__(self, coverage, quorum):
        self.coverage = list(coverage)
        self.quorum = list(quorum)

    @staticmethod
    def parse_params(quorum: str, coverage: str) -> "ThresholdContainer":
        q = [_parse_rel(s, i, quorum) for i, s in enumerate(quorum.split(","))] if quorum else []
        if not q:
            raise ValueError("quorum threshold setting requires at least one element, but none is given")
        c = [_parse_abs(s, i, coverage) for i, s in enumerate(coverage.split(","))] if coverage else []
        if not c:
            raise ValueError("coverage threshold setting requires at least one element, but none is given")
        if len(q) != len(c):
            if len(q) == 1:
                q = q * len(c)
            elif len(c) == 1:
                c = c * len(q)
            else:
                raise ValueError("number of coverage and quorum threshold must match, or either one must have a single value")
        return ThresholdContainer(c, q)

    def __len__(self):
        return len(self.coverage)


def _parse_rel(s, i, whole):
    try:
        v = float(s.strip())
    except ValueError:
        raise ValueError(f'threshold "{whole}" ({i + 1}. element in list) is required to be float, but isn\'t.')
    if not (0.0 <= v <= 1.0):
        raise ValueError(f'relative threshold "{whole}" ({i + 1}. element in list) must be within [0,1].')
    return Threshold(RELATIVE, v)


def _parse_abs(s, i, whole):
    t = s.strip()
    if not t.isdigit():
        raise ValueError(f'threshold "{whole}" ({i + 1}. element in list) is required to be integer, but isn\'t.')
    return Threshold(ABSOLUTE, int(t))


def coverage_abs(t: Threshold, n_groups: int) -> int:
    """c of AbacusByGroup::calc_growth: usize::max(1, t_coverage.to_absolute(G))  (abacus.rs:997)"""
    return max(1, t.to_absolute(n_groups))


def quorum_table(t: Threshold, n_groups: int) -> np.ndarray:
    """Tq[rank] = ceil((rank + 1.0) * q), q = max(0, t_quorum.to_relative(G))  (abacus.rs:998, 1009)"""
    if n_groups == 0:
        return np.zeros(0, dtype=np.uint32)
    q = max(0.0, t.to_relative(n_groups))
    r = np.arange(n_groups, dtype=np.float64)
    return np.ceil((r + 1.0) * q).astype(np.uint32)
