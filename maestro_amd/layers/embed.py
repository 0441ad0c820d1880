"""Parameter holders for patch embedding / pixelify with the reference's module tree (``maestro/layers/embed.py``).

``patchify_bands.<i>.{conv,norm}`` and ``pixelify_bands.<i>.conv`` keep the reference's parameter shapes
(``[E, C, P, P]`` and ``[C*P*P, Dd, 1, 1]``); the arithmetic is in the HIP engine.
"""

from __future__ import annotations

from torch import nn

from maestro_amd.layers.vit import _EngineOnly


def num_bands(bands) -> list[int]:
    return [bands] if isinstance(bands, int) else [len(b) for b in bands]


class PatchifyBands(_EngineOnly):
    def __init__(self, in_chans: int, embed_dim: int, patch_size: int) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.conv = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.GroupNorm(1, embed_dim)


class Patchify(_EngineOnly):
    def __init__(self, bands, embed_dim: int, patch_size: int) -> None:
        super().__init__()
        self.num_bands = num_bands(bands)
        self.patchify_bands = nn.ModuleList([PatchifyBands(c, embed_dim, patch_size) for c in self.num_bands])


class PixelifyBands(_EngineOnly):
    def __init__(self, embed_dim: int, out_chans: int, patch_size: int) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.conv = nn.Conv2d(embed_dim, out_chans * patch_size**2, kernel_size=1)


class Pixelify(_EngineOnly):
    def __init__(self, embed_dim: int, bands, patch_size: int) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.num_bands = num_bands(bands)
        self.pixelify_bands = nn.ModuleList([PixelifyBands(embed_dim, c, patch_size) for c in self.num_bands])
