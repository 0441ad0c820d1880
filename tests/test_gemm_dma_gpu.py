"""GPU parity of the LDS-DMA GEMM tile family (mh_gemm_bf16_tile, MH_TILE_DMA_*): exact on integer data for every layout, ragged edges,
split-K atomics and the fused epilogues; same oracle as tests/test_gemm_gpu.py."""

import pytest
import torch

from tests.test_gemm_gpu import _dev, _operands

pytestmark = pytest.mark.gpu
DMA_TILES = [1, 2, 3, 4, 5]   # MH_TILE_DMA_256, _256x128, _128x256, _128, _128x4


@pytest.mark.parametrize("tile", DMA_TILES)
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(256, 256, 32), (512, 768, 96), (300, 264, 320), (1000, 136, 1024), (40, 8, 64),
                                   (257, 520, 160), (8192, 768, 768)])
def test_dma_gemm_exact_integers(layout, shape, tile):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    if layout == 2 and M % 8:
        M = (M + 7) // 8 * 8
    if layout == 2:
        K = K + 37   # K-major operands: any K (rows past K read as zero through the descriptor)
    A, B, want = _operands(layout, M, N, K, dev, integer=True)
    C = torch.full((M, N), float("nan"), device=dev)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.OUT_F32, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(C, want), f"max diff {(C - want).abs().max().item()}"
    Cb = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], Cb, N, 0, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(Cb, want.bfloat16())


@pytest.mark.parametrize("tile", DMA_TILES)
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_dma_gemm_split_k_atomic(layout, tile):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 520, 264, 20000 if layout == 2 else 19968
    A, B, want = _operands(layout, M, N, K, dev, integer=True)
    C = torch.ones((M, N), device=dev)
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, hip.OUT_F32 | hip.ATOMIC, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(C, want + 1.0)


@pytest.mark.parametrize("tile", DMA_TILES)
def test_dma_gemm_epilogues_and_rejection(tile):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 600, 520, 320
    A, B, want = _operands(0, M, N, K, dev, integer=False)
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    pre = want + bias
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    aux = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.BIAS | hip.GELU, bias=bias, aux_out=aux, ldaux=N, tile=tile)
    assert (aux.float() - pre).abs().max() < 2e-2 and (C.float() - torch.nn.functional.gelu(pre)).abs().max() < 2e-2
    C32 = torch.empty((M, N), device=dev)
    hip.gemm(0, M, N, K, A, K, B, K, C32, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res, ldr=N, tile=tile)
    assert (C32 - (pre + res)).abs().max() < 1e-4 * K**0.5
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.DGELU, aux_in=aux, ldaux=N, tile=tile)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert (C.float() - want * x.grad).abs().max() < 3e-2
    A2 = torch.zeros(64, 40, device=dev, dtype=torch.bfloat16)   # K = 40: tail inside a K-minor row -> not eligible
    with pytest.raises(hip.HipExtensionError):
        hip.gemm(0, 64, 64, 40, A2, 40, A2, 40, torch.zeros(64, 64, device=dev), 64, hip.OUT_F32, tile=tile)


def test_grouped_tn_exact_and_timed():
    """One launch over many wgrad-shaped problems: exact on integer data, incl. ragged tile edges and odd K."""
    from maestro_amd import hip
    dev = _dev()
    shapes = [(768, 3072, 1000), (3072, 768, 1000), (768, 768, 333), (2304, 768, 1000), (512, 1536, 77), (40, 512, 4096),
              (264, 520, 2049)]
    probs, wants = [], []
    for i, (M, N, K) in enumerate(shapes):
        A, B, want = _operands(2, M, N, K + i, dev, integer=True)
        C = torch.full((M, N), float("nan"), device=dev)
        probs.append((A, B, C, M, N, K + i, M, N, N))
        wants.append(want)
    g = hip.GroupedTN(probs, dev)
    assert g.tiles == sum(-(-M // 256) * -(-N // 256) for M, N, _ in shapes)
    g.launch()
    torch.cuda.synchronize()
    for (A, B, C, *_), want in zip(probs, wants):
        assert torch.equal(C, want), (C - want).abs().max().item()


def test_grouped_tn_shared_output_accumulates():
    """Several problems writing one dW (an encoder shared by several groups) add up atomically; others are stored."""
    from maestro_amd import hip
    dev = _dev()
    M, N = 264, 520
    shared = torch.zeros(M, N, device=dev)
    probs, want_shared = [], torch.zeros(M, N, device=dev)
    for K in (300, 77, 1024):
        A, B, want = _operands(2, M, N, K, dev, integer=True)
        probs.append((A, B, shared, M, N, K, M, N, N))
        want_shared += want
    A, B, want_own = _operands(2, 512, 256, 200, dev, integer=True)
    own = torch.full((512, 256), float("nan"), device=dev)
    probs.append((A, B, own, 512, 256, 200, 512, 256, 256))
    hip.GroupedTN(probs, dev).launch()
    torch.cuda.synchronize()
    assert torch.equal(shared, want_shared) and torch.equal(own, want_own)


def test_tuned_tile_choice_is_recorded_and_exact():
    """With tuning on, the first call of a signature times every eligible tile and the choice is reused afterwards."""
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 1024, 768, 512
    A, B, want = _operands(0, M, N, K, dev, integer=True)
    C = torch.full((M, N), float("nan"), device=dev)
    hip.set_gemm_tuning(True)
    try:
        hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.OUT_F32)
    finally:
        hip.set_gemm_tuning(False)
    torch.cuda.synchronize()
    assert torch.equal(C, want)
    assert hip.gemm_tile_choices()[(0, M, N, K, hip.OUT_F32)] in hip.TILES
    C.fill_(float("nan"))
    hip.gemm(0, M, N, K, A, K, B, K, C, N, hip.OUT_F32)
    torch.cuda.synchronize()
    assert torch.equal(C, want)


@pytest.mark.parametrize("tile", [0] + DMA_TILES)
def test_gemm_colsum_side_output(tile):
    """MH_GEMM_COLSUM: the epilogue stores per-64-row-block column sums of the (pre-rounding) bf16 output."""
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 1000, 520, 256          # ragged in both tile dimensions
    A, B, want = _operands(1, M, N, K, dev, integer=True)
    aux = (torch.randn(M, N, device=dev) * 0.5).bfloat16()
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    acc = torch.full(((M + 63) // 64, N), float("nan"), device=dev)
    hip.gemm(1, M, N, K, A, K, B, N, C, N, hip.DGELU | hip.COLSUM, aux_in=aux, ldaux=N, colsum=acc, tile=tile)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref = want * x.grad
    assert (C.float() - ref).abs().max() <= 2e-2 * ref.abs().max()
    pad = torch.zeros(acc.shape[0] * 64, N, device=dev)
    pad[:M] = ref
    exp = pad.view(-1, 64, N).sum(1)
    assert (acc - exp).abs().max() <= 1e-3 * ref.abs().sum(0).max(), (acc - exp).abs().max().item()
