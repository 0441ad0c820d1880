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


@pytest.mark.parametrize("dim,depth,heads,mlp,n", [(48, 2, 3, 96, 7), (192, 2, 3, 768, 17), (512, 1, 16, 3072, 9)])
def test_transformer_matches_torch_prenorm_encoder(dim, depth, heads, mlp, n):
    """A pin that is NOT written in this repo: ``torch.nn.TransformerEncoder`` with ``norm_first=True`` and exact-erf GELU is
    library code for the same published pre-LN block (LayerNorm -> multi-head attention with q scaled by dim_head ** -0.5 and
    heads split as contiguous column blocks -> output projection -> residual; LayerNorm -> Linear -> GELU -> Linear -> residual;
    final LayerNorm).  Under the key map below it must reproduce the restated ``vit_pytorch`` Transformer wherever
    ``heads * dim_head == dim`` -- every Transformer of the reference's model factories (768 = 12 x 64, 1024 = 16 x 64,
    512 = 16 x 32, 192 = 3 x 64, 384 = 6 x 64; ``maestro/ssl/mae.py:135-174,309-378``).  The only freedom left is the
    q/k/v order inside ``to_qkv.weight`` (``chunk(3)`` = torch's ``in_proj_weight`` order) and the missing qkv bias (zero)."""
    dim_head = dim // heads
    torch.manual_seed(1)
    t = Transformer(dim, depth, heads, dim_head, mlp).double()
    with torch.no_grad():
        for p in t.parameters():
            p.add_(0.1 * torch.randn_like(p))
    layer = torch.nn.TransformerEncoderLayer(dim, heads, dim_feedforward=mlp, dropout=0.0, activation="gelu", batch_first=True,
                                             norm_first=True)
    enc = torch.nn.TransformerEncoder(layer, depth, norm=torch.nn.LayerNorm(dim), enable_nested_tensor=False).double()
    sd = t.state_dict()
    mapped = {"norm.weight": sd["norm.weight"], "norm.bias": sd["norm.bias"]}
    for i in range(depth):
        src, dst = f"layers.{i}.", f"layers.{i}."
        mapped.update({
            dst + "norm1.weight": sd[src + "0.norm.weight"], dst + "norm1.bias": sd[src + "0.norm.bias"],
            dst + "self_attn.in_proj_weight": sd[src + "0.to_qkv.weight"],
            dst + "self_attn.in_proj_bias": torch.zeros(3 * dim, dtype=torch.float64),
            dst + "self_attn.out_proj.weight": sd[src + "0.to_out.0.weight"], dst + "self_attn.out_proj.bias": sd[src + "0.to_out.0.bias"],
            dst + "norm2.weight": sd[src + "1.net.0.weight"], dst + "norm2.bias": sd[src + "1.net.0.bias"],
            dst + "linear1.weight": sd[src + "1.net.1.weight"], dst + "linear1.bias": sd[src + "1.net.1.bias"],
            dst + "linear2.weight": sd[src + "1.net.4.weight"], dst + "linear2.bias": sd[src + "1.net.4.bias"]})
    enc.load_state_dict(mapped, strict=True)
    enc.train()   # (eval mode may take torch's fused inference path; the training path is the plain composition)
    x = torch.randn(2, n, dim, dtype=torch.float64)
    got, want = t(x), enc(x)
    assert (got - want).abs().max().item() < 1e-10
    # gradients too: the backward of the block is what the HIP path is compared with
    gx, wx = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    t(gx).square().sum().backward()
    enc(wx).square().sum().backward()
    assert (gx.grad - wx.grad).abs().max().item() < 1e-8
    assert (t.layers[0][0].to_qkv.weight.grad - enc.layers[0].self_attn.in_proj_weight.grad).abs().max().item() < 1e-8
