"""Host-side (init-time) helpers: positional tables, their pooling to a modality grid, structural-mask draws.

These run once at construction or once per step on the HOST (index/RNG logic); the per-token arithmetic of the
hot path lives in the HIP kernels.  References: ``maestro/layers/utils.py:103-125,176-198`` (tables) and
``maestro/ssl/mae.py:178-226`` (structural masks).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F  # noqa: N812
from torch import Tensor


def posemb_sincos_2d(h: int, w: int, dim: int, date_dim: int, temperature: float = 10000.0) -> Tensor:
    """``[h, w, dim]`` table: sin/cos of x then y frequencies, zeros in the last ``date_dim`` channels."""
    if dim % 4 or date_dim % 4:
        raise ValueError(f"Invalid embedding dimensions {dim}, {date_dim}. Expected multiples of 4")
    n = (dim - date_dim) // 4
    omega = 1.0 / (temperature ** (torch.arange(n) / (n - 1)))
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    ya, xa = ys[..., None] * omega, xs[..., None] * omega
    return torch.cat([xa.sin(), xa.cos(), ya.sin(), ya.cos(), torch.zeros(h, w, date_dim)], dim=-1).float()


def pool_pos_table(table: Tensor, grid: int) -> Tensor:
    """Constant per-modality positional rows ``[grid*grid, dim]`` (block mean, bilinear pre-resize if needed).

    The reference recomputes this every step; it only depends on the geometry, so it is built once.
    """
    G = table.shape[0]  # noqa: N806
    if G % grid:
        G = grid * round(G / float(grid))  # noqa: N806
        table = F.interpolate(table.permute(2, 0, 1)[None], (G, G), mode="bilinear")[0].permute(1, 2, 0)
    r = G // grid
    return table.reshape(grid, r, grid, r, -1).mean(dim=(1, 3)).reshape(grid * grid, -1).contiguous()


def draw_struct_masks(groups, mods, generator=None) -> dict[str, Tensor]:
    """Structural Bernoulli masks per group, ``{group: bool [Beff, L]}``, drawn on the HOST generator.

    ``groups``: list of ``GroupSpec``; ``mods``: dict of ``ModSpec`` (see maestro_amd/ssl/mae.py).  Draw order per
    rejection-loop iteration = modalities in ``dataset.inputs`` order, each: mod, bands, dates, loc (only the active
    ones) -- exactly the reference's order so that a fixed seed gives bit-identical masks on the CPU generator.
    """
    # The random numbers come from torch (the reference's generator and draw order); the boolean bookkeeping around them runs
    # in numpy: these arrays are a few KiB, and torch's reductions over them go through the intra-op thread pool -- with the
    # default pool of a many-core host inside a container that may schedule a fraction of those cores, ONE ``.all(dim=1)`` was
    # measured at 6 ms (12 ms per step: more than half of the GPU's step time, spent on the host before the first launch).
    pending = {g.name: np.ones(g.Beff, dtype=bool) for g in groups}
    out = {g.name: np.ones((g.Beff, g.L), dtype=bool) for g in groups}
    while any(p.any() for p in pending.values()):
        draw = {}
        for m in mods.values():
            if m.gi:                     # band-groups 1.. of a modality: drawn together with its group 0
                continue
            B, D, L, G = m.Beff, m.D, m.L, m.G  # noqa: N806
            mk = np.zeros((B, G, D, L), dtype=bool)
            if m.p_mod:
                mk = mk | (torch.rand((B, 1, 1, 1), generator=generator) < m.p_mod).numpy()
            if m.p_bands:
                mk = mk | (torch.rand((B, G, 1, 1), generator=generator) < m.p_bands).numpy()
            if m.p_dates:
                mk = mk | (torch.rand((B, 1, D, 1), generator=generator) < m.p_dates).numpy()
            if m.p_loc:
                mk = mk | (torch.rand((B, 1, 1, L), generator=generator) < m.p_loc).numpy()
            for gi in range(G):          # the specs of one modality are named <modality>#<g> when there are several
                draw[m.name if G == 1 else f"{m.src}#{gi}"] = mk[:, gi].reshape(B, D * L)
        for g in groups:
            new = np.concatenate([draw[m.name] for m in g.mods], axis=1)
            take = pending[g.name]
            out[g.name] = np.where(take[:, None], new, out[g.name])
            pending[g.name] = out[g.name].all(axis=1)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}
