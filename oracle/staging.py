"""CPU restatement of the reference's input staging (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package).

* ``transform_rasters`` -- maestro/dataset/dataset.py:224-257: with ``use_transform`` every sample draws three booleans
  (``rng.choice([True, False])``) and, in this order, flips axis 2, flips axis 3 and swaps axes 2 and 3 of every raster of
  the sample (inputs AND raster targets share the three draws), then makes them contiguous.
"""
from __future__ import annotations

import numpy as np


def draw_flags(rng: np.random.Generator) -> int:
    """The three draws of one sample in the reference's order -> bit0 flip axis 2, bit1 flip axis 3, bit2 swap axes."""
    flags = 0
    for bit in range(3):
        if rng.choice([True, False]):
            flags |= 1 << bit
    return flags


def transform_rasters(rasters: dict[str, np.ndarray], flags: int) -> dict[str, np.ndarray]:
    """``rasters[name]``: one sample, ``[D, C, H, W]``."""
    out = {}
    for name, arr in rasters.items():
        if flags & 1:
            arr = np.flip(arr, axis=2)
        if flags & 2:
            arr = np.flip(arr, axis=3)
        if flags & 4:
            arr = np.swapaxes(arr, 2, 3)
        out[name] = np.ascontiguousarray(arr)
    return out
