"""Restatement of ``vit_pytorch.vit.Transformer`` (vit-pytorch 1.10.1; pin: reference ``poetry.lock:6339``).

The package is NOT in ``/root/reference`` and not installed, so this is a restatement of its published
definition, anchored on the reference's call sites (``maestro/ssl/mae.py:135-141,157-163,168-174``):

    Transformer(dim, depth, heads, dim_head, mlp_dim)
      layers[l][0] = Attention:  norm=LayerNorm(dim); to_qkv=Linear(dim, 3*heads*dim_head, bias=False);
                                 softmax(q k^T * dim_head**-0.5) v; to_out=Sequential(Linear(inner, dim), Dropout)
      layers[l][1] = FeedForward: net=Sequential(LayerNorm(dim), Linear(dim, mlp), GELU(erf), Dropout,
                                                 Linear(mlp, dim), Dropout)
      forward: for attn, ff in layers: x = attn(x) + x; x = ff(x) + x;  return norm(x)   (final LayerNorm)

State-dict sub-keys therefore are ``layers.<l>.0.{norm,to_qkv,to_out.0}``, ``layers.<l>.1.net.{0,1,4}``, ``norm``.
Dropout is 0 everywhere at the call sites, so it is omitted from the arithmetic (modules kept for key parity).
"parity unpinned" by the reference (its tests never run a forward); pinned here (tests/test_oracle_vit.py) by
``transformer_fp64``, an independent loop-level float64 restatement of the same equations, and by code that was NOT written
in this repo: ``torch.nn.TransformerEncoder(norm_first=True, activation="gelu")`` -- the torch library's implementation of the
same published pre-LN block -- reproduces this module's outputs and gradients to 1e-10 under the key map in that test.
"""

from __future__ import annotations

import math

import torch
from torch import Tensor, nn


class Attention(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int) -> None:
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head**-0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        project_out = not (heads == 1 and dim_head == dim)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0)) if project_out else nn.Identity()

    def forward(self, x: Tensor) -> Tensor:
        b, n, _ = x.shape
        h, d = self.heads, self.dim_head
        qkv = self.to_qkv(self.norm(x)).view(b, n, 3, h, d)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))  # [b h n d]
        attn = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        out = torch.matmul(attn, v).transpose(1, 2).reshape(b, n, h * d)
        return self.to_out(out)


class FeedForward(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.net = nn.Sequential(
            nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0),
            nn.Linear(hidden, dim), nn.Dropout(0.0),
        )

    def forward(self, x: Tensor) -> Tensor:
        return self.net(x)


class Transformer(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int, dropout: float = 0.0) -> None:
        super().__init__()
        if dropout:
            raise NotImplementedError("dropout is 0 at every MAESTRO call site")
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList(
            [nn.ModuleList([Attention(dim, heads, dim_head), FeedForward(dim, int(mlp_dim))]) for _ in range(depth)]
        )

    def forward(self, x: Tensor) -> Tensor:
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return self.norm(x)


# ----------------------------------------------------------------------------------------------
# Independent fp64 restatement (explicit sums, no nn.Module / F.* calls) used to pin the block above.
# ----------------------------------------------------------------------------------------------
def _ln64(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.sum(-1, keepdim=True) / x.shape[-1]
    var = ((x - mu) ** 2).sum(-1, keepdim=True) / x.shape[-1]
    return (x - mu) / torch.sqrt(var + eps) * w + b


def transformer_fp64(x: Tensor, sd: dict[str, Tensor], heads: int, dim_head: int) -> Tensor:
    """Evaluate the Transformer from a state dict in float64 with explicit formulas."""
    sd = {k: v.double() for k, v in sd.items()}
    x = x.double()
    depth = 1 + max([int(k.split(".")[1]) for k in sd if k.startswith("layers.")], default=-1)
    b, n, _ = x.shape
    inner = heads * dim_head
    for l in range(depth):
        p = f"layers.{l}.0."
        y = _ln64(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
        qkv = torch.einsum("bnc,oc->bno", y, sd[p + "to_qkv.weight"])
        q, k, v = qkv[..., :inner], qkv[..., inner : 2 * inner], qkv[..., 2 * inner :]
        out = torch.zeros(b, n, inner, dtype=torch.float64)
        for hh in range(heads):
            sl = slice(hh * dim_head, (hh + 1) * dim_head)
            s = torch.einsum("bid,bjd->bij", q[..., sl], k[..., sl]) / math.sqrt(dim_head)
            e = torch.exp(s - s.max(-1, keepdim=True).values)
            out[..., sl] = torch.einsum("bij,bjd->bid", e / e.sum(-1, keepdim=True), v[..., sl])
        if (p + "to_out.0.weight") in sd:
            out = torch.einsum("bni,oi->bno", out, sd[p + "to_out.0.weight"]) + sd[p + "to_out.0.bias"]
        x = out + x
        p = f"layers.{l}.1.net."
        y = _ln64(x, sd[p + "0.weight"], sd[p + "0.bias"])
        hdn = torch.einsum("bnc,oc->bno", y, sd[p + "1.weight"]) + sd[p + "1.bias"]
        hdn = 0.5 * hdn * (1.0 + torch.erf(hdn / math.sqrt(2.0)))
        x = torch.einsum("bnh,oh->bno", hdn, sd[p + "4.weight"]) + sd[p + "4.bias"] + x
    return _ln64(x, sd["norm.weight"], sd["norm.bias"])
