"""pansyn-v1 helpers that live on the host: random group orders for permuted growth.

The graph itself is generated in HBM (csrc/pansyn.hip); the R random orders of BASELINE
config 4 are Fisher-Yates shuffles driven by the same counter-based hash (stream 7).
"""
from __future__ import annotations

import numpy as np

_M = (1 << 64) - 1


def splitmix64(x: int) -> int:
    z = (x + 0x9E3779B97F4A7C15) & _M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
    return z ^ (z >> 31)


def key(seed: int, stream: int) -> int:
    return splitmix64((seed ^ ((0xA0761D6478BD642F * (stream + 1)) & _M)) & _M)


def random_orders(seed: int, n_orders: int, n_groups: int) -> np.ndarray:
    """(n_orders, n_groups) u32; row r = Fisher-Yates shuffle of 0..G-1 with h(seed, 7, r, i)."""
    out = np.empty((n_orders, n_groups), dtype=np.uint32)
    k7 = key(seed, 7)
    for r in range(n_orders):
        kr = splitmix64((k7 + r) & _M)
        perm = list(range(n_groups))
        for i in range(n_groups - 1, 0, -1):
            j = splitmix64((kr + i) & _M) % (i + 1)
            perm[i], perm[j] = perm[j], perm[i]
        out[r] = perm
    return out
