"""Parameter holders for the probe / finetune heads with the reference's module tree (``maestro/layers/head.py``):
``heads.<target>.{reduce.{norm,norm_fc,to_kv,query}, linear | conv}``.  The arithmetic is in
``maestro_amd.engine_sup.SupervisedEngine`` (HIP kernels of ``csrc/heads.hip`` + the shared LayerNorm / GEMM kernels)."""

from __future__ import annotations

import torch
from torch import nn

from maestro_amd.layers.embed import PixelifyBands
from maestro_amd.layers.vit import _EngineOnly


class AttentiveReduce(_EngineOnly):
    """head.py:28-62: LayerNorm -> to_kv -> one learned query per head attends over the token axis -> LayerNorm."""

    def __init__(self, dim: int, heads: int = 8) -> None:
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.norm, self.norm_fc = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_kv = nn.Linear(dim, dim * 2, bias=False)
        self.query = nn.Parameter(torch.randn(dim))


def _reduce(type_head: str, dim: int, heads: int):
    if type_head == "attentive":
        return AttentiveReduce(dim, heads)
    if type_head == "linear":   # head.py:77-78: torch.mean over the token axis, no parameters
        return None
    raise ValueError(f"Invalid head type {type_head}")


class ClassificationHead(_EngineOnly):
    def __init__(self, type_head: str, dim: int, num_classes: int, heads: int = 8) -> None:
        super().__init__()
        self.type_head, self.num_classes = type_head, num_classes
        red = _reduce(type_head, dim, heads)
        if red is not None:
            self.reduce = red
        self.linear = nn.Linear(dim, num_classes)


class PixelifyHead(PixelifyBands):
    def __init__(self, type_head: str, dim: int, out_chans: int, patch_size: int, heads: int = 8) -> None:
        super().__init__(dim, out_chans, patch_size)
        self.type_head, self.num_classes = type_head, out_chans
        red = _reduce(type_head, dim, heads)
        if red is not None:
            self.reduce = red
