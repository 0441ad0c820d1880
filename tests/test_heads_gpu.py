"""GPU parity of the probe / finetune kernels (csrc/heads.hip) against fp32 PyTorch restatements of the reference ops:
F.interpolate bilinear (mim.py:357-366), AttentiveReduce / mean (head.py:28-62, 77-78), the classification linear,
F.cross_entropy / F.binary_cross_entropy_with_logits on the rows kept by base.py:119-137."""
import pytest
import torch
import torch.nn.functional as F  # noqa: N812

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("h,H", [(5, 32), (4, 4), (8, 4), (3, 10)])
def test_token_resize_fwd_bwd(h, H):  # noqa: N803
    from maestro_amd import hip
    dev = _dev()
    B, D, E, pre, post = 2, 3, 64, 7, 5            # the modality's tokens sit between other rows of the sequence
    g = torch.Generator().manual_seed(h * 100 + H)
    x = torch.randn(B, pre + D * h * h + post, E, generator=g)
    out = torch.full((B, 2 + D * H * H, E), float("nan"), device=dev)
    hip.token_resize(x.to(dev), x.shape[1], pre, out, out.shape[1], 2, B, D, h, H, E)
    xr = x[:, pre: pre + D * h * h].reshape(B * D, h, h, E).permute(0, 3, 1, 2).clone().requires_grad_(True)
    want = F.interpolate(xr, (H, H), mode="bilinear")
    got = out[:, 2:].cpu().reshape(B * D, H, H, E).permute(0, 3, 1, 2)
    assert (got - want.detach()).abs().max() < 1e-5
    if h == H:
        assert torch.equal(got, xr.detach())       # same grid: exact copy
    dout = torch.randn(B, 2 + D * H * H, E, generator=g)
    want.backward(dout[:, 2:].reshape(B * D, H, H, E).permute(0, 3, 1, 2))
    din = torch.full_like(x, 1.0).to(dev)
    hip.token_resize_bwd(dout.to(dev), dout.shape[1], 2, din, x.shape[1], pre, B, D, h, H, E, accumulate=True)
    ref = torch.ones_like(x)
    ref[:, pre: pre + D * h * h] += xr.grad.permute(0, 2, 3, 1).reshape(B, D * h * h, E)
    assert (din.cpu() - ref).abs().max() < 1e-4


@pytest.mark.parametrize("dim", [192, 768])
@pytest.mark.parametrize("nb,T,Lr", [(3, 17, 16), (4, 333, 1), (37, 5, 1)])
def test_attentive_reduce_fwd_bwd(dim, nb, T, Lr):  # noqa: N803
    from maestro_amd import hip
    dev = _dev()
    heads, dh = 8, dim // 8
    g = torch.Generator().manual_seed(dim + T)
    kv16 = (torch.randn(nb * T * Lr, 2 * dim, generator=g) * 1.5).bfloat16()
    query = torch.randn(dim, generator=g)
    kvf = kv16.float().reshape(nb, T, Lr, 2 * dim).permute(0, 2, 1, 3).reshape(nb * Lr, T, 2 * dim).requires_grad_(True)
    q = query.clone().requires_grad_(True)
    k, v = kvf[..., :dim].reshape(-1, T, heads, dh), kvf[..., dim:].reshape(-1, T, heads, dh)
    attn = (torch.einsum("hd,sthd->sht", q.reshape(heads, dh), k) * dh ** -0.5).softmax(-1)
    want = torch.einsum("sht,sthd->shd", attn, v).reshape(-1, dim)
    out = torch.empty(nb * Lr, dim, device=dev)
    lse = torch.empty(nb * Lr, heads, device=dev)
    hip.attn_reduce_fwd(kv16.to(dev), query.to(dev), out, lse, nb, T, Lr, dim)
    assert (out.cpu() - want.detach()).abs().max() < 2e-4 * max(1.0, want.abs().max().item())
    dout = torch.randn(nb * Lr, dim, generator=g)
    want.backward(dout)
    dkv = torch.empty_like(kv16, device=dev)
    part = torch.full((hip.attn_reduce_partial_rows(nb * Lr), dim), float("nan"), device=dev)
    hip.attn_reduce_bwd(kv16.to(dev), query.to(dev), out, lse, dout.to(dev), dkv, part, nb, T, Lr, dim)
    dq = part.sum(0).cpu()
    assert (dq - q.grad).abs().max() < 2e-3 * q.grad.abs().max()
    ref = kvf.grad.reshape(nb, Lr, T, 2 * dim).permute(0, 2, 1, 3).reshape(nb * T * Lr, 2 * dim)
    err = (dkv.float().cpu() - ref).abs().max().item()
    assert err < 1e-2 * ref.abs().max().item() + 1e-6, err      # bf16 output rounding


def test_mean_reduce_and_head_linear():
    from maestro_amd import hip
    dev = _dev()
    nb, T, Lr, dim, C = 3, 7, 5, 192, 15
    g = torch.Generator().manual_seed(3)
    x = torch.randn(nb * T * Lr, dim, generator=g)
    out = torch.empty(nb * Lr, dim, device=dev)
    hip.mean_reduce_fwd(x.to(dev), out, nb, T, Lr, dim)
    want = x.reshape(nb, T, Lr, dim).mean(1).reshape(nb * Lr, dim)
    assert (out.cpu() - want).abs().max() < 1e-6
    dout = torch.randn(nb * Lr, dim, generator=g)
    dx = torch.empty(nb * T * Lr, dim, device=dev)
    hip.mean_reduce_bwd(dout.to(dev), dx, nb, T, Lr, dim)
    ref = (dout.reshape(nb, 1, Lr, dim) / T).expand(nb, T, Lr, dim).reshape(-1, dim)
    assert (dx.cpu() - ref).abs().max() < 1e-7
    # classification linear with an odd class count
    B = 6
    xb = torch.randn(B, dim, generator=g, requires_grad=True)
    W = torch.randn(C, dim, generator=g, requires_grad=True)  # noqa: N806
    bias = torch.randn(C, generator=g, requires_grad=True)
    logits = torch.empty(B, C, device=dev)
    hip.head_linear_fwd(xb.detach().to(dev), W.detach().to(dev), bias.detach().to(dev), logits, B, C, dim)
    wl = xb @ W.t() + bias
    assert (logits.cpu() - wl.detach()).abs().max() < 1e-4
    dl = torch.randn(B, C, generator=g)
    wl.backward(dl)
    dxg, dW, db = torch.empty(B, dim, device=dev), torch.ones(C, dim, device=dev), torch.ones(C, device=dev)  # noqa: N806
    hip.head_linear_bwd(xb.detach().to(dev), W.detach().to(dev), dl.to(dev), dxg, dW, db, B, C, dim)
    assert (dxg.cpu() - xb.grad).abs().max() < 1e-4 and (dW.cpu() - 1 - W.grad).abs().max() < 1e-4
    assert (db.cpu() - 1 - bias.grad).abs().max() < 1e-5


@pytest.mark.parametrize("tdtype", [torch.int64, torch.int32, torch.uint8])
@pytest.mark.parametrize("B,g,P,C,missing", [(2, 4, 8, 15, -1), (3, 2, 4, 19, 19), (5, 1, 1, 7, -1)])
def test_cross_entropy_patch_layout(B, g, P, C, missing, tdtype):  # noqa: N803
    from maestro_amd import hip
    dev = _dev()
    if tdtype == torch.uint8 and missing < 0:
        pytest.skip("unsigned targets cannot hold a negative missing value")
    S = g * P  # noqa: N806
    gen = torch.Generator().manual_seed(B * 10 + C)
    patch = torch.randn(B * g * g, P * P * C, generator=gen) * 2
    target = torch.randint(0, C, (B, S, S), generator=gen)
    target[torch.rand(B, S, S, generator=gen) < 0.2] = missing
    # image-layout logits [B, C, S, S] from the patch layout ('(p1 p2 c)' columns)
    img = patch.reshape(B, g, g, P, P, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, S, S).clone().requires_grad_(True)
    lg = img.permute(0, 2, 3, 1).reshape(-1, C)
    tg = target.reshape(-1)
    idx = (tg != missing).nonzero().squeeze(1)
    want = F.cross_entropy(lg.index_select(0, idx), tg.index_select(0, idx))
    want.backward()
    cnt, acc = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
    t_dev = target.to(tdtype).to(dev)
    hip.count_valid(t_dev, missing, cnt)
    assert cnt.item() == len(idx)
    for dt in (torch.float32, torch.bfloat16):
        acc.zero_()
        d = torch.full((B * g * g, P * P * C), float("nan"), device=dev, dtype=dt)
        hip.ce_loss(patch.to(dev), t_dev, missing, cnt, acc, d, B, g, P, C)
        assert abs(acc.item() - want.item()) < 1e-4 * abs(want.item())
        ref = img.grad.reshape(B, C, g, P, g, P).permute(0, 2, 4, 3, 5, 1).reshape(B * g * g, P * P * C)
        tol = 1e-6 if dt == torch.float32 else 1e-2 * ref.abs().max().item()
        assert (d.float().cpu() - ref).abs().max() <= tol
    # nothing valid: loss 0, zero gradient (base.py:130-131 skips the target)
    none = torch.full_like(t_dev, missing)
    cnt.zero_(); acc.zero_()
    hip.count_valid(none, missing, cnt)
    d = torch.full((B * g * g, P * P * C), float("nan"), device=dev)
    hip.ce_loss(patch.to(dev), none, missing, cnt, acc, d, B, g, P, C)
    assert cnt.item() == 0 and acc.item() == 0.0 and float(d.abs().max()) == 0.0


def test_bce_with_missing_rows():
    from maestro_amd import hip
    dev = _dev()
    B, C = 9, 15
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(B, C, generator=gen) * 3).requires_grad_(True)
    t = (torch.rand(B, C, generator=gen) < 0.3).float()
    t[2, 4] = -1.0
    t[7, 0] = -1.0
    keep = (t != -1).all(dim=1).nonzero().squeeze(1)
    want = F.binary_cross_entropy_with_logits(x.index_select(0, keep), t.index_select(0, keep))
    want.backward()
    acc, d = torch.zeros(1, device=dev), torch.empty(B, C, device=dev)
    hip.bce_loss(x.detach().to(dev), t.to(dev), -1, acc, d, B, C)
    assert abs(acc.item() - want.item()) < 1e-5 and (d.cpu() - x.grad).abs().max() < 1e-6
