"""EXPERIMENTAL kernel, gated: the 128 x 128 NT tile on v_mfma_f32_32x32x16_bf16 (csrc/gemm_m32.hip, MH_TILE_M32_128) was written
at the end of round 3 without GPU time left -- compiled and ISA-checked, never run.  These tests are what it has to pass before
anything dispatches to it; they run only with MAESTRO_TEST_EXPERIMENTAL=1 (so that an unverified kernel cannot turn the suite
red), first thing next round:

    MAESTRO_TEST_EXPERIMENTAL=1 python -m pytest tests/test_gemm_m32_gpu.py -x -q && python scripts/bench_m32.py
"""

import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("MAESTRO_TEST_EXPERIMENTAL") != "1",
                                 reason="experimental kernel, not yet run on hardware (MAESTRO_TEST_EXPERIMENTAL=1 enables)")]

SHAPES = [(128, 128, 128), (256, 384, 192), (300, 136, 512), (1000, 256, 1024), (8192, 3072, 768)]   # ragged M / N, the fc1 shape


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _operands(M, N, K, dev, integer):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    if integer:      # exactly representable: the fp32-accumulated result is bit exact whatever the MFMA's summation order
        a = torch.randint(-3, 4, (M, K), generator=g).float() + (torch.arange(M)[:, None] % 3 == 0).float()
        w = torch.randint(-2, 3, (N, K), generator=g).float()
    else:
        a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K**0.5
    a, w = a.to(dev).bfloat16(), w.to(dev).bfloat16()
    return a, w, a.float() @ w.float().t()


@pytest.mark.parametrize("shape", SHAPES)
def test_m32_exact_integers_and_guard_rows(shape):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    A, W, want = _operands(M, N, K, dev, integer=True)
    C = torch.full((M + 3, N), 7.0, device=dev, dtype=torch.bfloat16)
    hip.gemm(0, M, N, K, A, K, W, K, C, N, 0, tile=hip.TILE_M32_128)
    torch.cuda.synchronize()
    assert torch.equal(C[:M], want.bfloat16()), f"max diff {(C[:M].float() - want).abs().max().item()}"
    assert bool((C[M:] == 7.0).all()), "rows beyond M were written"
    bias = torch.randint(-4, 5, (N,), generator=torch.Generator().manual_seed(5)).float().to(dev)
    hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.BIAS, bias=bias, tile=hip.TILE_M32_128)
    torch.cuda.synchronize()
    assert torch.equal(C[:M], (want + bias).bfloat16()) and bool((C[M:] == 7.0).all())


@pytest.mark.parametrize("shape", SHAPES[1:])
def test_m32_gelu_epilogues_match_the_16x16x32_kernel(shape):
    """Same epilogue arithmetic on fp32 sums that differ only by the MFMA's internal summation order: outputs within one bf16
    ulp / one byte code of mh_gemm_bf16's register-staged kernel, almost all identical."""
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    A, W, _ = _operands(M, N, K, dev, integer=False)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(11)).to(dev)

    def run(tile, flags, **kw):
        C = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
        hip.gemm(0, M, N, K, A, K, W, K, C, N, flags, tile=tile, **kw)
        torch.cuda.synchronize()
        return C

    def close(a, b, frac):
        d = (a.view(torch.int16).int() - b.view(torch.int16).int()).abs()
        return d.max().item() <= 1 and (d != 0).float().mean().item() < frac

    assert close(run(hip.TILE_M32_128, 0), run(hip.TILE_REG_128, 0), 5e-2)
    assert close(run(hip.TILE_M32_128, hip.BIAS | hip.GELU, bias=bias), run(hip.TILE_REG_128, hip.BIAS | hip.GELU, bias=bias), 5e-2)
    fl = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
    aux = [torch.zeros((M, N), device=dev, dtype=torch.uint8) for _ in range(2)]
    c_new = run(hip.TILE_M32_128, fl, bias=bias, aux_out=aux[0], ldaux=N)
    c_ref = run(hip.TILE_REG_128, fl, bias=bias, aux_out=aux[1], ldaux=N)
    assert close(c_new, c_ref, 5e-2)
    dcode = (aux[0].int() - aux[1].int()).abs()
    assert dcode.max().item() <= 1 and (dcode != 0).float().mean().item() < 5e-2
    pre = [torch.zeros((M, N), device=dev, dtype=torch.bfloat16) for _ in range(2)]      # the pre-activation saved as bf16
    run(hip.TILE_M32_128, hip.BIAS | hip.GELU, bias=bias, aux_out=pre[0], ldaux=N)
    run(hip.TILE_REG_128, hip.BIAS | hip.GELU, bias=bias, aux_out=pre[1], ldaux=N)
    assert close(pre[0], pre[1], 5e-2)


def test_m32_declines_what_it_does_not_serve():
    from maestro_amd import hip
    dev = _dev()
    A = torch.zeros(256, 512, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(256, 256, device=dev, dtype=torch.bfloat16)
    Cf = torch.zeros(256, 256, device=dev)
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(1, 256, 256, 512, A, 512, A, 512, C, 256, 0, tile=hip.TILE_M32_128)                     # NN
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(0, 256, 256, 96, A, 512, A, 512, C, 256, 0, tile=hip.TILE_M32_128)                      # K % 64
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(0, 256, 256, 512, A, 512, A, 512, Cf, 256, hip.OUT_F32, tile=hip.TILE_M32_128)          # fp32 output
