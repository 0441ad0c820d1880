"""Oracle restatement of the reference's embedding / encoding layers (PyTorch CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference lines it follows.
The arithmetic is written out explicitly (unfold + matmul, explicit statistics) rather than through
``nn.Conv2d`` / ``nn.GroupNorm`` forward calls, so that it doubles as the specification of the data
layouts the HIP kernels use.  Parameters live in modules with the reference's names so state dicts load
both ways.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F  # noqa: N812
from torch import Tensor, nn


# --------------------------------------------------------------------------- encodings
def posemb_sincos_2d(h: int, w: int, dim: int, date_dim: int, temperature: float = 10000.0) -> Tensor:
    """2-D sin/cos table ``[h, w, dim]``; last ``date_dim`` channels are zero.

    Reference: ``maestro/layers/utils.py:176-198``.
    Channel blocks: ``[sin(x w_i), cos(x w_i), sin(y w_i), cos(y w_i), 0 x date_dim]``, ``w_i = T^(-i/(n-1))``,
    ``n = (dim - date_dim) / 4``.
    """
    if dim % 4 or date_dim % 4:
        raise ValueError(f"Invalid embedding dimensions {dim}, {date_dim}. Expected multiples of 4")
    n = (dim - date_dim) // 4
    omega = 1.0 / (temperature ** (torch.arange(n) / (n - 1)))
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    ya = yy[:, :, None] * omega[None, None, :]
    xa = xx[:, :, None] * omega[None, None, :]
    return torch.cat([xa.sin(), xa.cos(), ya.sin(), ya.cos(), torch.zeros(h, w, date_dim)], dim=-1).float()


def pool_pos_encoding(table: Tensor, grid: int) -> Tensor:
    """Pool a ``[G, G, dim]`` table to a modality grid -> ``[grid*grid, dim]``.

    Reference: ``maestro/layers/utils.py:103-125`` applied to the pos-enc buffer
    (``maestro/ssl/mim.py:242-245``): if ``G % grid`` the table is first bilinearly resized to
    ``grid * round(G / grid)``; then each token takes the mean of its ``(G'/grid)^2`` block.
    """
    G = table.shape[0]  # noqa: N806
    enc = table
    if G % grid:
        resize = grid * round(G / float(grid))
        enc = F.interpolate(enc.permute(2, 0, 1)[None], (resize, resize), mode="bilinear")[0].permute(1, 2, 0)
        G = resize  # noqa: N806
    if G < grid:  # reference's ``max(grid_size, shape)`` expand branch only triggers for size-1 tables
        raise ValueError("positional table smaller than modality grid")
    r = G // grid
    enc = enc.reshape(grid, r, grid, r, -1).mean(dim=(1, 3))
    return enc.reshape(grid * grid, -1)


def date_features(dates: Tensor, ref_date: Tensor, fac_date_enc: float) -> Tensor:
    """Per-(b, d) date features ``[B, D, 8]`` = ``[diff x4, sin doy, cos doy, sin hour, cos hour] * fac``.

    Reference: ``maestro/layers/utils.py:128-167``.  Kept in the reference's operation order and dtypes:
    ``year`` stays int16, ``doy/365.25`` and ``hour/24`` are fp32, ``(year + doy) - (year_ref + doy_ref)`` is
    evaluated in fp32 (``diff`` is quantised to ~1.2e-4 years).
    """
    year, doy, hour = dates[:, :, 0], dates[:, :, 1] / 365.25, dates[:, :, 2] / 24.0
    year_ref, doy_ref = ref_date[:, :, 0], ref_date[:, :, 1] / 365.25
    diff = (year + doy) - (year_ref + doy_ref)
    doy = 2 * math.pi * doy
    hour = 2 * math.pi * hour
    feats = torch.stack([diff, diff, diff, diff, doy.sin(), doy.cos(), hour.sin(), hour.cos()], dim=-1)
    return feats * fac_date_enc


def encode_dates(dates: Tensor, ref_date: Tensor, dim: int, date_dim: int, fac_date_enc: float,
                 grid_size: int, len_bands: int) -> Tensor:
    """Date encoding ``[B, G*D, grid^2, dim]``: zeros in the first ``dim-date_dim`` channels.

    Reference: ``maestro/layers/utils.py:128-173`` (broadcast over tokens, tiled over band-groups
    band-group-major on the date axis).
    """
    if date_dim != 8:
        raise NotImplementedError("reference always uses date_dim=8 (4 diff copies + 4 sin/cos)")
    feats = date_features(dates, ref_date, fac_date_enc)  # [B, D, 8]
    B, D, _ = feats.shape  # noqa: N806
    enc = torch.cat([torch.zeros(B, D, dim - date_dim), feats], dim=-1)
    enc = enc[:, :, None, :].expand(B, D, grid_size * grid_size, dim)
    if len_bands > 1:
        enc = enc[:, None].expand(B, len_bands, D, grid_size * grid_size, dim).flatten(1, 2)
    return enc


# --------------------------------------------------------------------------- group / ungroup
def group_mods(x: dict[str, Tensor], fusion_mode: str, groups: list[tuple]) -> dict[str, Tensor]:
    """``[B, D, L, C]`` per modality -> per-group sequences.  Reference: ``maestro/layers/utils.py:12-47``."""
    if fusion_mode in ("shared", "monotemp"):
        return {m: t.flatten(0, 1) for m, t in x.items()}
    flat = {m: t.flatten(1, 2) for m, t in x.items()}
    if fusion_mode == "mod":
        return flat
    out: dict[str, list] = {}
    for name_mod, name_group in groups:
        out.setdefault(name_group, []).append(flat[name_mod])
    return {g: torch.cat(ts, dim=1) for g, ts in out.items()}


def ungroup_mods(xg: dict[str, Tensor], fusion_mode: str, groups: list[tuple], num_dates: dict[str, int],
                 grid_size: dict[str, int]) -> dict[str, Tensor]:
    """Inverse of :func:`group_mods`.  Reference: ``maestro/layers/utils.py:50-100``."""
    if fusion_mode in ("shared", "monotemp"):
        return {m: t.unflatten(0, (-1, num_dates[m])) for m, t in xg.items()}
    if fusion_mode == "mod":
        per_mod = dict(xg)
    else:
        members: dict[str, list] = {}
        for name_mod, name_group in groups:
            members.setdefault(name_group, []).append(name_mod)
        per_mod = {}
        for g, t in xg.items():
            sizes = [num_dates[m] * grid_size[m] ** 2 for m in members[g]]
            for m, part in zip(members[g], torch.split(t, sizes, dim=1)):
                per_mod[m] = part
    return {m: t.unflatten(1, (num_dates[m], -1)) for m, t in per_mod.items()}


# --------------------------------------------------------------------------- patchify / pixelify
def _num_bands(bands) -> list[int]:
    return [bands] if isinstance(bands, int) else [len(b) for b in bands]


def im2col_patches(x: Tensor, patch: int) -> Tensor:
    """``[N, C, H, W]`` -> ``[N, L, C*P*P]`` with K index ``c*P*P + p1*P + p2`` (= Conv2d weight flattening)."""
    N, C, H, W = x.shape  # noqa: N806
    g = H // patch
    x = x.reshape(N, C, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(N, g * g, C * patch * patch)


class PatchifyBands(nn.Module):
    """Conv(k=s=P)+bias then GroupNorm(1, E) over the whole (E x tokens) image; per-channel affine.

    Reference: ``maestro/layers/embed.py:37-66`` (stats: biased variance, eps 1e-5 = nn.GroupNorm defaults).
    """

    def __init__(self, in_chans: int, embed_dim: int, patch_size: int) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.conv = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.GroupNorm(1, embed_dim)

    def forward(self, x: Tensor) -> Tensor:
        B, D, C, H, W = x.shape  # noqa: N806
        cols = im2col_patches(x.reshape(B * D, C, H, W), self.patch_size)  # [BD, L, K]
        y = cols @ self.conv.weight.reshape(self.conv.out_channels, -1).t() + self.conv.bias  # [BD, L, E]
        mu = y.mean(dim=(1, 2), keepdim=True)
        var = ((y - mu) ** 2).mean(dim=(1, 2), keepdim=True)
        y = (y - mu) / torch.sqrt(var + self.norm.eps) * self.norm.weight + self.norm.bias
        return y.reshape(B, D, -1, y.shape[-1])


class Patchify(nn.Module):
    """Per-band-group patch embedding, band-groups concatenated on the date axis.  Ref ``embed.py:8-34``."""

    def __init__(self, bands, embed_dim: int, patch_size: int) -> None:
        super().__init__()
        self.num_bands = _num_bands(bands)
        self.patchify_bands = nn.ModuleList([PatchifyBands(c, embed_dim, patch_size) for c in self.num_bands])

    def forward(self, x: Tensor) -> Tensor:
        parts = torch.split(x, self.num_bands, dim=2)
        return torch.cat([pb(p) for pb, p in zip(self.patchify_bands, parts)], dim=1)


class PixelifyBands(nn.Module):
    """1x1 conv to ``C*P*P`` then depth-to-space; out channel index ``(p1*P + p2)*C + c``.

    Reference: ``maestro/layers/embed.py:122-160``.
    """

    def __init__(self, embed_dim: int, out_chans: int, patch_size: int) -> None:
        super().__init__()
        self.patch_size, self.out_chans = patch_size, out_chans
        self.conv = nn.Conv2d(embed_dim, out_chans * patch_size**2, kernel_size=1)

    def patches(self, x: Tensor) -> Tensor:
        """``[B, D, L, Dd]`` -> patch-layout reconstruction ``[B, D, L, P*P*C]``."""
        return x @ self.conv.weight.reshape(self.conv.out_channels, -1).t() + self.conv.bias

    def forward(self, x: Tensor) -> Tensor:
        B, D, L, _ = x.shape  # noqa: N806
        g, P, C = round(L**0.5), self.patch_size, self.out_chans  # noqa: N806
        y = self.patches(x).reshape(B, D, g, g, P, P, C)
        return y.permute(0, 1, 6, 2, 4, 3, 5).reshape(B, D, C, g * P, g * P)


class Pixelify(nn.Module):
    """Per-band-group pixelify + token mask repeated to pixel resolution.  Ref ``embed.py:69-119``."""

    def __init__(self, embed_dim: int, bands, patch_size: int) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.num_bands = _num_bands(bands)
        self.pixelify_bands = nn.ModuleList([PixelifyBands(embed_dim, c, patch_size) for c in self.num_bands])

    def forward(self, x: Tensor, mask: Tensor) -> tuple[Tensor, Tensor]:
        G = len(self.num_bands)  # noqa: N806
        B, GD, L, _ = x.shape  # noqa: N806
        D, g, P = GD // G, round(L**0.5), self.patch_size  # noqa: N806
        xs = x.reshape(B, G, D, L, -1)
        ms = mask.reshape(B, G, D, g, 1, g, 1).expand(B, G, D, g, P, g, P).reshape(B, G, D, 1, g * P, g * P)
        rec = [pb(xs[:, i]) for i, pb in enumerate(self.pixelify_bands)]
        msk = [ms[:, i].expand(B, D, c, g * P, g * P) for i, c in enumerate(self.num_bands)]
        return torch.cat(rec, dim=2), torch.cat(msk, dim=2)
