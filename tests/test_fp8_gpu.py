"""fp8 path (BASELINE configs[4]): quantisers against torch's OCP float8 casts (bit-exact), the scaled-MFMA GEMM on exact
integer data (operand lane map, K order, tiling: bit-exact vs an fp32 matmul), its epilogues against the bf16 kernel fed with
the SAME (already quantised) values, and the scale-update rule."""

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _q(x, fmt):
    return x.to(torch.float8_e4m3fn if fmt == 0 else torch.float8_e5m2)


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quantiser_matches_torch_float8(dev, fmt, dtype):
    from maestro_amd import hip
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(257, 64, generator=g) * torch.exp(3 * torch.randn(257, 1, generator=g))).to(dtype)   # wide dynamic range
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-9, 447.0, 449.0, -1e6, 0.0019, 0.06], dtype=dtype)
    sc = hip.Fp8Scales(2, dev)
    sc.scale[1] = 0.25
    src = x.to(dev)
    dst, dst_t = torch.zeros(x.shape, dtype=torch.uint8, device=dev), torch.zeros(64, 257, dtype=torch.uint8, device=dev)
    qb = hip.QuantBatch([dict(src=src, dst=dst, dst_t=dst_t, slot=1, format=fmt)], sc, dev)
    qb.launch(2)
    torch.cuda.synchronize()
    lim = hip.FP8_MAX[fmt]
    want = _q((x.float() * 0.25).clamp(-lim, lim), fmt).view(torch.uint8)
    assert torch.equal(dst.cpu(), want), "cast differs from torch's OCP float8 conversion (round to nearest even, saturating)"
    assert torch.equal(dst_t.cpu(), want.t().contiguous())
    assert float(sc.amax[1]) == float(x.float().abs().max()) and float(sc.amax[0]) == 0.0
    sc.update(fmt=fmt, margin=1)
    amax = float(x.float().abs().max())
    import math
    assert float(sc.scale[1]) == 2.0 ** (math.floor(math.log2(lim / amax)) - 1) and float(sc.amax[1]) == 0.0
    assert float(sc.scale[0]) == 1.0 and float(sc.descale[1]) == 1.0 / float(sc.scale[1])


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 384), (1000, 136, 256), (64, 768, 768)])
@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_fp8_integer_exact(dev, M, N, K, a_fmt):
    """Small integers are exact in e4m3 / e5m2 and every partial sum is exact in fp32: any mistake in the operand lane map,
    the swizzles, the K order or the tiling changes bits."""
    from maestro_amd import hip
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-4, 5, (M, K), generator=g).float()
    b = torch.randint(-4, 5, (N, K), generator=g).float()
    # asymmetric content so that a transposed / permuted fragment cannot hide
    a[:, ::7] += 1.0
    A8, B8 = _q(a, a_fmt).view(torch.uint8).to(dev), _q(b, 0).view(torch.uint8).to(dev)  # noqa: N806
    C = torch.full((M, N), float("nan"), device=dev)  # noqa: N806
    da, db = torch.tensor([0.5], device=dev), torch.tensor([4.0], device=dev)
    hip.gemm_fp8(M, N, K, A8, K, B8, K, C, N, da, db, flags=hip.OUT_F32, a_format=a_fmt)
    torch.cuda.synchronize()
    assert torch.equal(C.cpu(), 2.0 * (a @ b.t()))


def test_gemm_fp8_epilogues_match_bf16_kernel(dev):
    """With operands that are exactly representable in fp8 the fp8 GEMM and the bf16 GEMM see the same numbers; every
    product (<= 8 significant bits) and the fp32 accumulation order per K block differ only in grouping, so outputs agree to
    fp32 rounding; the fused epilogues (bias + GELU + saved GELU', fp32 residual, fp8 copy of the output) must follow."""
    from maestro_amd import hip
    g = torch.Generator().manual_seed(11)
    M, N, K = 640, 1024, 256  # noqa: N806
    a = _q(torch.randn(M, K, generator=g), 0).float()
    b = _q(torch.randn(N, K, generator=g) * 0.1, 0).float()
    A8, B8 = _q(a, 0).view(torch.uint8).to(dev), _q(b, 0).view(torch.uint8).to(dev)  # noqa: N806
    A16, B16 = a.bfloat16().to(dev), b.bfloat16().to(dev)  # noqa: N806
    one = torch.ones(1, device=dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    # fc1-style: bias + GELU, saves GELU', bf16 out + fp8 copy
    C8, C16 = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2))  # noqa: N806
    X8, X16 = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2))  # noqa: N806
    c8, s8, amax = torch.zeros(M, N, dtype=torch.uint8, device=dev), torch.tensor([8.0], device=dev), torch.zeros(1, device=dev)
    fl = hip.BIAS | hip.GELU | hip.AUX_DGELU
    hip.gemm_fp8(M, N, K, A8, K, B8, K, C8, N, one, one, flags=fl, bias=bias, aux_out=X8, ldaux=N, c8=c8, ldc8=N, c8_scale=s8, c8_amax=amax)
    hip.gemm(hip.GEMM_NT, M, N, K, A16, K, B16, K, C16, N, fl, bias=bias, aux_out=X16, ldaux=N)
    torch.cuda.synchronize()
    assert (C8.float() - C16.float()).abs().max() <= 2e-2 * C16.float().abs().max() and (X8.float() - X16.float()).abs().max() < 2e-2
    assert (C8.float() - C16.float()).abs().mean() < 1e-4          # identical but for a few bf16 rounding flips
    want8 = _q((C8.float() * 8.0).clamp(-448, 448), 0).view(torch.uint8)
    assert torch.equal(c8.cpu(), want8.cpu()) and abs(float(amax) - float(C8.float().abs().max())) < 1e-6
    # proj / fc2-style: fp32 out + bias + residual
    D8, D16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)  # noqa: N806
    fl = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
    hip.gemm_fp8(M, N, K, A8, K, B8, K, D8, N, one, one, flags=fl, bias=bias, res=res, ldr=N)
    hip.gemm(hip.GEMM_NT, M, N, K, A16, K, B16, K, D16, N, fl, bias=bias, res=res, ldr=N)
    torch.cuda.synchronize()
    assert (D8 - D16).abs().max() < 1e-4
