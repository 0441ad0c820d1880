"""Parameter holders for the ViT blocks with the state-dict layout of ``vit_pytorch.vit.Transformer`` 1.10.1.

The reference builds these at ``maestro/ssl/mae.py:135-141,157-163,168-174``.  Here the modules only own the
parameters (so checkpoints interchange key-for-key); the arithmetic runs in the HIP engine
(``maestro_amd/engine.py``: LayerNorm / MFMA GEMM / fused attention kernels).  Calling ``forward`` is an error.
"""

from __future__ import annotations

from torch import nn


class _EngineOnly(nn.Module):
    def forward(self, *a, **k):  # noqa: D102
        raise RuntimeError(f"{type(self).__name__} holds parameters only; run it through maestro_amd.engine.MAEEngine")


class Attention(_EngineOnly):
    def __init__(self, dim: int, heads: int, dim_head: int) -> None:
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head**-0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(0.0))


class FeedForward(_EngineOnly):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0),
                                 nn.Linear(hidden, dim), nn.Dropout(0.0))


class Transformer(_EngineOnly):
    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int) -> None:
        super().__init__()
        self.dim, self.depth, self.heads, self.dim_head, self.mlp_dim = dim, depth, heads, dim_head, int(mlp_dim)
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList(
            [nn.ModuleList([Attention(dim, heads, dim_head), FeedForward(dim, int(mlp_dim))]) for _ in range(depth)])
