"""GPU parity of the stream-K GEMM (mh_gemm_bf16_sk, csrc/gemm_sk.hip) through the C ABI.

Integer operands make every fp32 partial sum exactly representable, so the result must be BIT EXACT whatever the work split:
tiles owned by one workgroup, tiles shared by two, tiles cut into many pieces (tiny problems on a large grid), ragged M.
"""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _operands(layout, M, N, K, dev, integer=True):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + layout)
    if integer:
        a = torch.randint(-3, 4, (M, K), generator=g).float()
        b = torch.randint(-2, 3, (K, N), generator=g).float()
        a += (torch.arange(M)[:, None] % 3 == 0).float()
    else:
        a = torch.randn(M, K, generator=g)
        b = torch.randn(K, N, generator=g) / K**0.5
    a, b = a.to(dev).bfloat16(), b.to(dev).bfloat16()
    want = a.float() @ b.float()
    A = a.contiguous()
    B = b.t().contiguous() if layout == 0 else b.contiguous()
    return A, B, want


SHAPES = [
    (192, 128, 128),      # one tile, two K steps: 256 workgroups cut it to pieces (grid is clamped to the unit count)
    (256, 256, 256),
    (1000, 384, 512),     # ragged M
    (8192, 768, 768),     # out-proj of the aerial encoder
    (3200, 768, 3072),    # fc2 of the s2 encoder: every tile shared by two or three workgroups
    (2048, 512, 1536),
    (40, 128, 192),       # M smaller than a wave tile
]
TILES = [15, 16, 17]      # MH_TILE_SK_192, MH_TILE_SK_256 (four waves, register-staged), MH_TILE_SK_DMA_256 (eight waves, LDS-DMA ring)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_sk_exact_integers(tile, layout, shape):
    from maestro_amd import hip
    dev = _dev()
    M, N, K = shape
    if tile == 17:        # 256-wide output tiles
        N = (N + 255) // 256 * 256
    A, B, want = _operands(layout, M, N, K, dev)
    g = torch.Generator().manual_seed(5)
    bias = torch.randint(-4, 5, (N,), generator=g).float().to(dev)
    res = torch.randint(-8, 9, (M, N), generator=g).float().to(dev)
    for grid in (None, 64, 100, 7):
        C = torch.full((M, N), float("nan"), device=dev)
        rc = hip.gemm_sk(tile, layout, M, N, K, A, K, B, B.shape[1], C, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res,
                         ldr=N, grid=grid)
        assert rc == 0, hip.lib().mh_last_error()
        torch.cuda.synchronize()
        assert torch.equal(C, want + bias + res), f"grid {grid}: max diff {(C - (want + bias + res)).abs().max().item()}"
        Cb = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        rc = hip.gemm_sk(tile, layout, M, N, K, A, K, B, B.shape[1], Cb, N, 0, grid=grid)
        assert rc == 0, hip.lib().mh_last_error()
        torch.cuda.synchronize()
        assert torch.equal(Cb, want.bfloat16()), f"grid {grid} (bf16)"
        ws = hip.sk_workspace(tile, grid or hip.sk_grid())
        assert hip.sk_error_flag(ws) == 0
        assert int(ws[:4092].view(torch.int32).abs().sum().item()) == 0, "flag words must be zero between launches"


@pytest.mark.parametrize("tile", TILES)
def test_sk_matches_the_library_rule_on_random_data(tile):
    """Random data: the stream-K result equals mh_gemm_bf16's up to the order in which a shared tile's K ranges are added."""
    from maestro_amd import hip
    dev = _dev()
    for layout, (M, N, K) in ((0, (8192, 768, 3072)), (1, (8192, 768, 2304)), (0, (3200, 768, 768))):
        A, B, want = _operands(layout, M, N, K, dev, integer=False)
        g = torch.Generator().manual_seed(11)
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        C0 = torch.empty((M, N), device=dev)
        C1 = torch.empty((M, N), device=dev)
        flags = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
        hip.gemm(layout, M, N, K, A, K, B, B.shape[1], C0, N, flags, bias=bias, res=res, ldr=N)
        assert hip.gemm_sk(tile, layout, M, N, K, A, K, B, B.shape[1], C1, N, flags, bias=bias, res=res, ldr=N) == 0
        torch.cuda.synchronize()
        assert (C1 - C0).abs().max().item() <= 2e-5 * K**0.5, (C1 - C0).abs().max().item()   # fp32 re-association only
        assert (C1 - (want + bias + res)).abs().max().item() < 1e-3
        # deterministic: the same launch twice gives the same bits
        C2 = torch.empty((M, N), device=dev)
        assert hip.gemm_sk(tile, layout, M, N, K, A, K, B, B.shape[1], C2, N, flags, bias=bias, res=res, ldr=N) == 0
        torch.cuda.synchronize()
        assert torch.equal(C1, C2)


@pytest.mark.parametrize("tile", [15, 17])
def test_sk_through_the_tile_table_and_graph_replay(tile):
    """hip.gemm(tile=TILE_SK_*) routes to the stream-K entry; a captured launch replays with the same result."""
    from maestro_amd import hip
    dev = _dev()
    M, N, K = 3200, 768, 3072
    A, B, want = _operands(0, M, N, K, dev)
    C = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip.gemm(0, M, N, K, A, K, B, K, C, N, 0, tile=tile)        # eager first: allocates the stream's workspace
        s.synchronize()
        assert torch.equal(C, want.bfloat16())
        C.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            hip.gemm(0, M, N, K, A, K, B, K, C, N, 0, tile=tile)
        for _ in range(3):
            C.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(C, want.bfloat16())


def test_sk_declines_what_it_does_not_serve():
    from maestro_amd import hip
    dev = _dev()
    A = torch.zeros((256, 256), device=dev, dtype=torch.bfloat16)
    C = torch.zeros((256, 256), device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(256, device=dev)
    assert hip.gemm_sk(15, 0, 256, 256, 256, A, 256, A, 256, C, 256, hip.BIAS, bias=bias) == -2      # epilogue not served
    assert hip.gemm_sk(15, 0, 256, 200, 256, A, 256, A, 256, C, 256, 0) == -2                       # N % 128
    assert hip.gemm_sk(15, 0, 256, 256, 96, A, 256, A, 256, C, 256, 0) == -2                        # K % 64
    assert hip.gemm_sk(17, 0, 256, 128, 256, A, 256, A, 256, C, 256, 0) == -2                       # the 256-wide tile: N % 256
