"""Pins the restated vit_pytorch Transformer (oracle/vit.py) to an independent fp64 loop-level restatement.

The reference holds no vector for this third-party block ("parity unpinned" by the reference).
"""

import pytest
import torch

from oracle.vit import Transformer, transformer_fp64


@pytest.mark.parametrize("dim,depth,heads,dim_head,mlp,n", [(48, 2, 3, 16, 96, 7), (64, 1, 16, 32, 192, 33)])
def test_transformer_matches_fp64(dim, depth, heads, dim_head, mlp, n):
    torch.manual_seed(0)
    t = Transformer(dim, depth, heads, dim_head, mlp)
    with torch.no_grad():
        for p in t.parameters():
            p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(2, n, dim)
    want = transformer_fp64(x, t.state_dict(), heads, dim_head)
    got = t(x).double()
    assert (got - want).abs().max().item() < 5e-5
    keys = set(t.state_dict())
    assert {"norm.weight", "layers.0.0.norm.weight", "layers.0.0.to_qkv.weight", "layers.0.0.to_out.0.bias",
            "layers.0.1.net.0.weight", "layers.0.1.net.1.bias", "layers.0.1.net.4.weight"} <= keys
    assert "layers.0.0.to_qkv.bias" not in keys
