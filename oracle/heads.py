"""CPU oracle of the probe / finetune heads (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``maestro/layers/head.py``:
* ``AttentiveReduce``      head.py:28-62   one learned query per head attends over the token axis
* ``ClassificationHead``   head.py:65-94   reduce (mean | attentive) over all tokens, then Linear
* ``PixelifyHead``         head.py:97-130  reduce over the date axis per location, then PixelifyBands (1x1 conv + depth-to-space)
and the loss of ``maestro/train/base.py:98-151`` (``loss_pred``).
Parameter names equal the reference's, so state dicts interchange key for key.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn import functional as F  # noqa: N812

from oracle.layers import PixelifyBands


class AttentiveReduce(nn.Module):
    def __init__(self, dim: int, heads: int = 8) -> None:
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.norm, self.norm_fc = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_kv = nn.Linear(dim, 2 * dim, bias=False)
        self.query = nn.Parameter(torch.randn(dim))

    def forward(self, x: Tensor) -> Tensor:                      # x [B, T, dim] -> [B, dim]
        B, T, dim = x.shape  # noqa: N806
        h, d = self.heads, dim // self.heads
        kv = self.to_kv(self.norm(x))
        k, v = kv[..., :dim].reshape(B, T, h, d), kv[..., dim:].reshape(B, T, h, d)
        dots = torch.einsum("hd,bthd->bht", self.query.reshape(h, d), k) * self.scale
        attn = dots.softmax(dim=-1)
        out = torch.einsum("bht,bthd->bhd", attn, v).reshape(B, dim)
        return self.norm_fc(out)


def _make_reduce(type_head: str, dim: int, heads: int):
    if type_head == "attentive":
        return AttentiveReduce(dim, heads)
    if type_head == "linear":                                    # head.py:77-78: plain mean over the token axis
        return None
    raise ValueError(f"Invalid head type {type_head}")


class ClassificationHead(nn.Module):
    def __init__(self, type_head: str, dim: int, num_classes: int, heads: int = 8) -> None:
        super().__init__()
        red = _make_reduce(type_head, dim, heads)
        if red is not None:
            self.reduce = red
        self.linear = nn.Linear(dim, num_classes)

    def forward(self, x: Tensor, ssl_phase: str) -> Tensor:      # x [B, T, dim] -> [B, num_classes]
        if ssl_phase == "probe":
            x = x.detach()
        x = self.reduce(x) if hasattr(self, "reduce") else x.mean(dim=1)
        return self.linear(x)


class PixelifyHead(PixelifyBands):
    def __init__(self, type_head: str, dim: int, out_chans: int, patch_size: int, heads: int = 8) -> None:
        super().__init__(dim, out_chans, patch_size)
        red = _make_reduce(type_head, dim, heads)
        if red is not None:
            self.reduce = red

    def forward(self, x: Tensor, ssl_phase: str) -> Tensor:      # x [B, D, L, dim] -> [B, 1, C, g*P, g*P]
        if ssl_phase == "probe":
            x = x.detach()
        B, D, L, dim = x.shape  # noqa: N806
        seq = x.permute(0, 2, 1, 3).reshape(B * L, D, dim)       # one sequence over the dates per location
        red = self.reduce(seq) if hasattr(self, "reduce") else seq.mean(dim=1)
        return super().forward(red.reshape(B, 1, L, dim))


def compute_loss_pred(dataset, batch: dict, logits: dict) -> Tensor:
    """``maestro/train/base.py:98-151`` without the metric updates: cross entropy (segment / classif) or BCE-with-logits
    (multilabel) over the rows whose target is not ``missing_val``; summed over the targets."""
    total = None
    for name, target in dataset.targets.items():
        lg, tg = logits[name], batch[name]
        if target.type_target == "segment":
            lg = lg[:, 0].permute(0, 2, 3, 1).reshape(-1, lg.shape[2])        # 'b 1 c h w -> (b h w) c'
            tg = tg.reshape(-1).long()                                        # 'b 1 1 h w -> (b h w)'
        elif target.type_target == "multilabel_classif":
            tg = tg.float()
        else:
            tg = tg.long()
        keep = (tg != target.missing_val).all(dim=1) if tg.ndim > 1 else tg != target.missing_val
        idx = keep.nonzero().squeeze(1)
        if len(idx) == 0:
            continue
        fn = F.binary_cross_entropy_with_logits if target.type_target == "multilabel_classif" else F.cross_entropy
        term = fn(lg.index_select(0, idx), tg.index_select(0, idx))
        total = term if total is None else total + term
    if total is None:
        total = 0 * list(logits.values()).pop().mean()
    return total
