"""mh_gemm_grouped (persistent grouped NT / NN GEMM, csrc/gemm_persist.hip): bit-exact on integer data against an fp32
matmul, and bit-identical to the per-problem kernel (mh_gemm_bf16) for every fused epilogue -- the K order per output element
and the epilogue arithmetic are the same, only the tiling / scheduling differs.  Host scheduler: every output element is
covered exactly once for ragged shapes and any worker count."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _ints(*shape, seed, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


SHAPES = [  # (M, N, K) sets: ragged rows (strips of half tiles), column remainders, K of 2 .. 24 steps, mixed K in one launch
    [(8192 // 8, 768, 768), (3200 // 8, 768, 768)],
    [(1000, 520, 96), (264, 1032, 64), (130, 136, 160)],
    [(640, 2304, 256)],
    [(3200, 768, 128), (2048, 512, 64), (384, 384, 96), (128, 128, 64), (8, 8, 64)],
]


@pytest.mark.parametrize("workers", [8, 24, 256])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shapes", SHAPES)
def test_grouped_integer_exact(dev, shapes, layout, workers):
    from maestro_amd import hip
    probs, want = [], []
    for i, (M, N, K) in enumerate(shapes):  # noqa: N806
        a = _ints(M, K, seed=10 * i + 1)
        b = _ints(N, K, seed=10 * i + 2) if layout == 0 else _ints(K, N, seed=10 * i + 2)
        want.append(a @ (b.t() if layout == 0 else b))
        A, B = a.bfloat16().to(dev), b.bfloat16().to(dev)  # noqa: N806
        C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)  # noqa: N806
        probs.append(dict(A=A, B=B, C=C, M=M, N=N, K=K, lda=K, ldb=B.shape[1], ldc=N, flags=hip.OUT_F32))
    g = hip.GroupedGemm(layout, probs, dev, n_workers=workers)
    g.launch()
    torch.cuda.synchronize()
    for pr, w in zip(probs, want):
        assert torch.equal(pr["C"].cpu(), w), (pr["M"], pr["N"], pr["K"])
    # a second launch of the same table (graph replays do exactly this) gives the same bits
    for pr in probs:
        pr["C"].fill_(float("nan"))
    g.launch()
    torch.cuda.synchronize()
    for pr, w in zip(probs, want):
        assert torch.equal(pr["C"].cpu(), w)


@pytest.mark.parametrize("split", [1, 2, 4])
def test_grouped_epilogues_match_single_launches(dev, split):
    """The encoder layer's four forward GEMMs and the dgrad pair with their fused epilogues, two groups per launch: the
    grouped launch must reproduce mh_gemm_bf16 bit for bit (outputs, saved GELU' values, column-sum side output)."""
    from maestro_amd import hip
    g = torch.Generator().manual_seed(5)
    Ms, dim, mlp = (640, 328), 256, 1024  # noqa: N806
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5)  # noqa: E731

    def run(layout, N, K, flags, **extra):  # noqa: N803
        probs, refs = [], []
        for M in Ms:  # noqa: N806
            A = rnd(M, K).bfloat16().to(dev)  # noqa: N806
            B = (rnd(N, K) if layout == 0 else rnd(K, N)).bfloat16().to(dev)  # noqa: N806
            f32 = bool(flags & hip.OUT_F32)
            kw = dict(flags=flags)
            if flags & hip.BIAS:
                kw["bias"] = rnd(N).to(dev)
            if flags & hip.RESIDUAL:
                kw["res"], kw["ldr"] = rnd(M, N).to(dev), N
            if flags & hip.MULAUX:
                kw["aux_in"], kw["ldaux"] = rnd(M, N).bfloat16().to(dev), N
            outs = []
            for _ in range(2):
                o = dict(C=torch.full((M, N), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=dev))
                if flags & hip.AUX_DGELU:
                    o["aux_out"], kw["ldaux"] = torch.zeros(M, N, dtype=torch.bfloat16, device=dev), N
                if flags & hip.COLSUM:
                    o["colsum"] = torch.zeros((M + 63) // 64, N, device=dev)
                outs.append(o)
            hip.gemm(layout, M, N, K, A, K, B, B.shape[1], outs[0]["C"], N, bias=kw.get("bias"), res=kw.get("res"),
                     ldr=kw.get("ldr", 0), aux_in=kw.get("aux_in"), aux_out=outs[0].get("aux_out"), ldaux=kw.get("ldaux", 0),
                     colsum=outs[0].get("colsum"), flags=flags, tile=hip.TILE_REG_128)
            probs.append(dict(A=A, B=B, M=M, N=N, K=K, lda=K, ldb=B.shape[1], ldc=N, **kw, **outs[1]))
            refs.append(outs[0])
        hip.GroupedGemm(layout, probs, dev, split=split).launch()
        torch.cuda.synchronize()
        for pr, ref in zip(probs, refs):
            for key, want in ref.items():
                got = pr[key]
                assert torch.equal(got.view(torch.int16 if got.dtype == torch.bfloat16 else torch.int32),
                                   want.view(torch.int16 if want.dtype == torch.bfloat16 else torch.int32)), (layout, N, K, flags, key)

    run(0, 3 * dim, dim, 0)                                                        # qkv
    run(0, dim, dim, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL)                        # proj + residual
    run(0, mlp, dim, hip.BIAS | hip.GELU | hip.AUX_DGELU)                          # fc1 + GELU, saves GELU'
    run(0, dim, mlp, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL)                        # fc2 + residual
    run(1, mlp, dim, hip.MULAUX | hip.COLSUM)                                      # dgrad fc2 (x GELU', bias-gradient block sums)
    run(1, dim, mlp, 0)                                                            # dgrad fc1
