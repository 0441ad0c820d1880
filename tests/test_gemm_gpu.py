"""GPU parity of mh_gemm_bf16 (all three operand layouts + fused epilogues) through the C ABI."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _operands(layout, M, N, K, dev, integer):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + layout)
    if integer:  # exactly representable -> the fp32-accumulated result must be bit exact
        a = torch.randint(-3, 4, (M, K), generator=g).float()
        b = torch.randint(-2, 3, (K, N), generator=g).float()
        a += (torch.arange(M)[:, None] % 3 == 0).float()  # asymmetric
    else:
        a = torch.randn(M, K, generator=g)
        b = torch.randn(K, N, generator=g) / K**0.5
    a, b = a.to(dev).bfloat16(), b.to(dev).bfloat16()
    want = a.float() @ b.float()
    A = a.contiguous() if layout in (0, 1) else a.t().contiguous()       # [M,K] or [K,M]
    B = b.t().contiguous() if layout == 0 else b.contiguous()            # [N,K] or [K,N]
    return A, B, want


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 192), (200, 72, 104), (16, 8, 8), (1000, 136, 1000)])
def test_gemm_exact_integers(layout, shape):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    if layout == 2 and M % 8:
        M = (M + 7) // 8 * 8
    A, B, want = _operands(layout, M, N, K, dev, integer=True)
    C = torch.full((M, N), float("nan"), device=dev)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.OUT_F32)
    torch.cuda.synchronize()
    assert torch.equal(C, want), f"max diff {(C - want).abs().max().item()}"
    Cb = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], Cb, N, 0)
    torch.cuda.synchronize()
    assert torch.equal(Cb, want.bfloat16())


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_split_k_atomic(layout):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 264, 200, 5000  # K = "tokens": not a multiple of anything
    if layout != 2:
        K = 5000 // 8 * 8
    A, B, want = _operands(layout, M, N, K, dev, integer=True)
    C = torch.ones((M, N), device=dev)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.OUT_F32 | hip.ATOMIC)
    torch.cuda.synchronize()
    assert torch.equal(C, want + 1.0)


def test_gemm_epilogues():
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 300, 264, 320
    A, B, want = _operands(0, M, N, K, dev, integer=False)
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    pre = want + bias
    # bias + GELU with saved pre-activation
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    aux = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.BIAS | hip.GELU, bias=bias, aux_out=aux, ldaux=N)
    torch.cuda.synchronize()
    assert (aux.float() - pre).abs().max() < 2e-2
    assert (C.float() - torch.nn.functional.gelu(pre)).abs().max() < 2e-2
    # bias + residual, fp32 out (out-of-place residual stream)
    C = torch.empty((M, N), device=dev)
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res, ldr=N)
    torch.cuda.synchronize()
    assert (C - (pre + res)).abs().max() < 1e-4 * K**0.5
    # dgelu: C = acc * gelu'(aux)
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.DGELU, aux_in=aux, ldaux=N)
    torch.cuda.synchronize()
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert (C.float() - want * x.grad).abs().max() < 3e-2
    # the pair the engine uses: the forward saves GELU'(pre-activation) (AUX_DGELU), the backward multiplies by it (MULAUX)
    dsave = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.BIAS | hip.GELU | hip.AUX_DGELU, bias=bias, aux_out=dsave, ldaux=N)
    torch.cuda.synchronize()
    assert (C.float() - torch.nn.functional.gelu(pre)).abs().max() < 2e-2
    xp = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(xp).sum().backward()
    assert (dsave.float() - xp.grad).abs().max() < 1e-2, "saved derivative differs from GELU'(x)"
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.MULAUX, aux_in=dsave, ldaux=N)
    torch.cuda.synchronize()
    prod = want * dsave.float()
    assert ((C.float() - prod).abs().max() / prod.abs().max()).item() < 1e-2
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.AUX_DGELU, aux_out=dsave, ldaux=N)          # needs the GELU epilogue
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.MULAUX | hip.DGELU, aux_in=dsave, ldaux=N)   # exclusive


def test_gemm_bad_arguments_fail_loudly():
    from maestro_amd import hip
    dev = _dev()
    A = torch.zeros(8, 12, device=dev, dtype=torch.bfloat16)
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(0, 8, 8, 12, A, 12, A, 12, torch.zeros(8, 8, device=dev), 8, hip.OUT_F32)  # K % 8


@pytest.mark.parametrize("M,N,K", [(512, 768, 256), (1000, 1536, 192)])
def test_saved_gelu_derivative_as_bytes(M, N, K):
    """``MH_GEMM_AUX_U8``: the GELU derivative saved by the fc1 epilogue as one byte per element (code = round((d + 0.13) * 200))
    decodes to the bf16 one within half a step (0.0025) + bf16 rounding, the activation output is unchanged, and the backward's
    ``MULAUX`` epilogue reading the bytes equals the product with the decoded values."""
    from maestro_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(M)
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)  # noqa: N806
    B = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)  # noqa: N806
    bias = torch.randn(N, generator=g).to(dev)
    act16, act8 = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2))
    aux16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux8 = torch.zeros(M, N, dtype=torch.uint8, device=dev)
    fl = hip.BIAS | hip.GELU | hip.AUX_DGELU
    hip.gemm(hip.GEMM_NT, M, N, K, A, K, B, K, act16, N, fl, bias=bias, aux_out=aux16, ldaux=N)
    hip.gemm(hip.GEMM_NT, M, N, K, A, K, B, K, act8, N, fl | hip.AUX_U8, bias=bias, aux_out=aux8, ldaux=N)
    torch.cuda.synchronize()
    assert torch.equal(act16.view(torch.int16), act8.view(torch.int16))
    dec = aux8.float() / 200.0 - 0.13
    assert (dec - aux16.float()).abs().max() <= 0.0025 + 2 ** -8 * 1.13 + 1e-6
    assert float(aux16.float().min()) < -0.1 and float(aux16.float().max()) > 1.1      # the whole range of GELU' is exercised
    # backward: dh = (dY W) * GELU'
    dy = (torch.randn(M, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    W = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).to(dev)  # noqa: N806
    rows = (M + 63) // 64
    d16, d8 = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2))
    cs16, cs8 = torch.zeros(rows, N, device=dev), torch.zeros(rows, N, device=dev)
    dec16 = dec.to(torch.bfloat16)          # (exactly representable: at most 8 significant bits ... not quite: compare numerically)
    hip.gemm(hip.GEMM_NN, M, N, K, dy, K, W, N, d8, N, hip.MULAUX | hip.COLSUM | hip.AUX_U8, aux_in=aux8, ldaux=N, colsum=cs8)
    hip.gemm(hip.GEMM_NN, M, N, K, dy, K, W, N, d16, N, hip.MULAUX | hip.COLSUM, aux_in=dec16, ldaux=N, colsum=cs16)
    torch.cuda.synchronize()
    assert (d8.float() - d16.float()).abs().max() <= 2e-2 * d16.float().abs().max()
    assert (cs8 - cs16).abs().max() <= 2e-2 * cs16.abs().max()
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(hip.GEMM_NT, M, N, K, A, K, B, K, act8, N, hip.BIAS | hip.GELU | hip.AUX_U8, bias=bias, aux_out=aux8, ldaux=N)


# ---- MH_TILE_PP_128 (gemm_pp.hip): persistent workgroups, the epilogue of tile t inside the main loop of tile t + 1
_PP_SHAPES = [(5100, 2048, 512), (8192, 3072, 768), (3000, 1536, 640), (1000, 256, 512)]   # > 512 tiles, 3 tiles per workgroup,
                                                                                          # a ragged run, one tile per workgroup

@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", _PP_SHAPES)
def test_gemm_pp_exact_integers(layout, shape):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    A, B, want = _operands(layout, M, N, K, dev, integer=True)
    Cb = torch.full((M + 3, N), 7.0, device=dev, dtype=torch.bfloat16)     # three guard rows behind the output
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], Cb, N, 0, tile=hip.TILE_PP_128)
    torch.cuda.synchronize()
    assert torch.equal(Cb[:M], want.bfloat16()), f"max diff {(Cb[:M].float() - want).abs().max().item()}"
    assert bool((Cb[M:] == 7.0).all()), "rows beyond M were written"
    g = torch.Generator().manual_seed(5)
    bias = torch.randint(-4, 5, (N,), generator=g).float().to(dev)
    res = torch.randint(-9, 10, (M, N), generator=g).float().to(dev)
    C = torch.full((M + 3, N), 7.0, device=dev)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res, ldr=N,
             tile=hip.TILE_PP_128)
    torch.cuda.synchronize()
    assert torch.equal(C[:M], want + bias + res) and bool((C[M:] == 7.0).all())


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", _PP_SHAPES[:3])
def test_gemm_pp_epilogues_match_the_one_tile_kernel(layout, shape):
    """Same fp32 sums (same MFMA order), same epilogue arithmetic -> the persistent kernel's outputs are compared bit for bit with
    mh_gemm_bf16's register-staged kernel (column sums: other summation order, fp32 tolerance)."""
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    A, B, _ = _operands(layout, M, N, K, dev, integer=False)
    g = torch.Generator().manual_seed(11)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    lda, ldb = A.shape[1], B.shape[1]

    def run(tile, flags, out_dtype, **kw):
        C = torch.zeros((M, N), device=dev, dtype=out_dtype)
        hip.gemm(layout, M, N, K, A, lda, B, ldb, C, N, flags, tile=tile, **kw)
        torch.cuda.synchronize()
        return C

    for flags, dt, kw in [(0, torch.bfloat16, {}),
                          (hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, torch.float32, dict(bias=bias, res=res, ldr=N))]:
        assert torch.equal(run(hip.TILE_PP_128, flags, dt, **kw), run(hip.TILE_REG_128, flags, dt, **kw)), flags
    # fc1: bias + GELU, GELU' saved as bytes
    fl = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
    aux = [torch.zeros((M, N), device=dev, dtype=torch.uint8) for _ in range(2)]
    c_pp = run(hip.TILE_PP_128, fl, torch.bfloat16, bias=bias, aux_out=aux[0], ldaux=N)
    c_ref = run(hip.TILE_REG_128, fl, torch.bfloat16, bias=bias, aux_out=aux[1], ldaux=N)
    # (the two kernels contract the GELU arithmetic into FMAs differently: a few results land on the neighbouring bf16 / code)
    ulp = (c_pp.view(torch.int16).int() - c_ref.view(torch.int16).int()).abs()
    assert ulp.max().item() <= 1 and (ulp != 0).float().mean().item() < 2e-3, (ulp.max().item(), (ulp != 0).float().mean().item())
    dcode = (aux[0].int() - aux[1].int()).abs()
    assert dcode.max().item() <= 1 and (dcode != 0).float().mean().item() < 2e-3
    # fc2 dgrad: multiply by the saved derivative + 64-row block column sums
    fl = hip.MULAUX | hip.AUX_U8 | hip.COLSUM
    cs = [torch.full(((M + 63) // 64, N), float("nan"), device=dev) for _ in range(2)]
    c_pp = run(hip.TILE_PP_128, fl, torch.bfloat16, aux_in=aux[1], ldaux=N, colsum=cs[0])
    c_ref = run(hip.TILE_REG_128, fl, torch.bfloat16, aux_in=aux[1], ldaux=N, colsum=cs[1])
    ulp = (c_pp.view(torch.int16).int() - c_ref.view(torch.int16).int()).abs()
    assert ulp.max().item() <= 1 and (ulp != 0).float().mean().item() < 2e-3
    assert torch.isfinite(cs[0]).all() and (cs[0] - cs[1]).abs().max().item() <= 2e-5 * cs[1].abs().max().item() + 1e-6


def test_gemm_pp_declines_what_it_does_not_serve():
    from maestro_amd import hip
    dev = _dev()
    A = torch.zeros(256, 512, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(256, 256, device=dev, dtype=torch.bfloat16)
    for args in [dict(K=448), dict(N=192), dict(flags=hip.BIAS)]:       # K < 512, N % 128, an epilogue it has no form for
        K, N, flags = args.get("K", 512), args.get("N", 256), args.get("flags", 0)
        with pytest.raises(hip.HipExtensionError):
            hip.gemm(0, 256, N, K, A, 512, A, 512, C, 256, flags, bias=torch.zeros(256, device=dev), tile=hip.TILE_PP_128)


def test_shipped_library_has_no_wrong_output_tiles():
    """MH_TILE_PP_128_DIAG1..5 (ablation builds that skip parts of the kernel) exist only under -DMH_DIAG_TILES: the C ABI of the
    shipped library declines them (-2, nothing launched, the output untouched) for every layout and epilogue."""
    import os

    from maestro_amd import hip
    if "MH_DIAG_TILES" in os.environ.get("MH_BUILD_FLAGS", ""):
        pytest.skip("an ablation build")
    dev = _dev()
    A = torch.ones(256, 512, device=dev, dtype=torch.bfloat16)
    for layout in (0, 1):
        for diag in range(1, 6):
            C = torch.full((256, 512), 7.0, device=dev, dtype=torch.bfloat16)
            with pytest.raises(hip.HipExtensionError):
                hip.gemm(layout, 256, 512, 512, A, 512, A.t().contiguous() if layout else A, 512, C, 512, 0, tile=hip.TILE_PP_128 + diag)
            torch.cuda.synchronize()
            assert (C == 7.0).all()


# ---- MH_TILE_REG_64 / MH_TILE_REG_192: the register-staged kernel with 64 x 128 / 192 x 128 tiles
@pytest.mark.parametrize("tile_name", ["TILE_REG_64", "TILE_REG_192"])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", [(200, 136, 192), (1000, 768, 512), (8192, 768, 768), (330, 72, 104)])
def test_gemm_other_tile_heights(tile_name, layout, shape):
    from maestro_amd import hip
    dev = _dev()
    tile = getattr(hip, tile_name)
    M, N, K = shape
    A, B, want = _operands(layout, M, N, K, dev, integer=True)
    C = torch.full((M + 2, N), 7.0, device=dev)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.OUT_F32, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(C[:M], want) and bool((C[M:] == 7.0).all()), f"max diff {(C[:M] - want).abs().max().item()}"
    # the fused epilogues: bit for bit the 128 x 128 kernel's (same K order per output element, same epilogue code)
    A, B, _ = _operands(layout, M, N, K, dev, integer=False)
    g = torch.Generator().manual_seed(2)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    for flags, dt, kw in [(0, torch.bfloat16, {}), (hip.BIAS | hip.GELU, torch.bfloat16, dict(bias=bias)),
                          (hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, torch.float32, dict(bias=bias, res=res, ldr=N))]:
        outs = []
        for t in (tile, hip.TILE_REG_128):
            C = torch.zeros((M, N), device=dev, dtype=dt)
            hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, flags, tile=t, **kw)
            outs.append(C)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), (flags, (outs[0].float() - outs[1].float()).abs().max().item())
