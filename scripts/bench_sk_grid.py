"""Round 6: the eight-wave stream-K tile (MH_TILE_SK_DMA_256) by GRID size -- how the split of tiles x K steps over the persistent
workgroups interacts with the L2: workgroups that share a tile read DISJOINT K ranges, so operand sharing between CUs only survives
where different tiles sweep the same K range at the same time (grid = tiles: one tile each; grid = 2 x tiles: two lockstep halves).
us per launch, hot / cold as scripts/bench_sk.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
SHAPES = [("fc2", 0, 8192, 768, 3072, F32), ("dfc1", 1, 8192, 768, 3072, 0), ("dqkv", 1, 8192, 768, 2304, 0), ("proj", 0, 8192, 768, 768, F32),
          ("s2 fc2", 0, 3200, 768, 3072, F32), ("jnt fc2", 0, 11392, 768, 3072, F32), ("dec fc2", 0, 32768, 512, 3072, F32),
          ("dec dfc1", 1, 32768, 512, 3072, 0), ("ds2 fc2", 0, 12800, 512, 3072, F32)]


def make(lay, M, N, K, fl, grid):  # noqa: N803
    A = torch.randn(M, K, device=dev).bfloat16()  # noqa: N806
    B = ((torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)) / K ** 0.5).bfloat16()  # noqa: N806
    out = torch.empty(M, N, dtype=torch.float32 if fl & hip.OUT_F32 else torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev) if fl & hip.BIAS else None
    res = torch.randn(M, N, device=dev) if fl & hip.RESIDUAL else None
    nbytes = sum(t.numel() * t.element_size() for t in (A, B, out, res) if t is not None)
    if grid is None:
        return (lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, bias=bias, res=res, ldr=N if res is not None else 0)), nbytes
    return (lambda: hip.gemm_sk(hip.TILE_SK_DMA_256, lay, M, N, K, A, K, B, B.shape[1], out, N, fl, bias=bias, res=res,
                                ldr=N if res is not None else 0, grid=grid)), nbytes


def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, lay, M, N, K, fl in SHAPES:
    tiles = -(-M // 256) * (N // 256)
    grids = [None] + sorted({g for g in (tiles, 2 * tiles, 3 * tiles, 4 * tiles, 128, 192, 224, 240, 256) if 64 <= g <= 256})
    out = []
    for g in grids:
        f0, b0 = make(lay, M, N, K, fl, g)
        r = max(2, int(1.5 * 2 ** 30 / b0) + 1)
        fs = [f0] + [make(lay, M, N, K, fl, g)[0] for _ in range(r - 1)]
        for f in fs:
            f()
        hot = min(timeit(fs[:1], 16) for _ in range(4))
        cold = min(timeit(fs, 2 * len(fs)) for _ in range(4))
        out.append(f"{'auto' if g is None else g}: {hot:.1f}/{cold:.1f}")
        del fs
        torch.cuda.empty_cache()
    print(f"{name:9s} {'NT' if lay == 0 else 'NN'} ({M},{N},{K}) tiles {tiles:4d} | " + "  ".join(out), flush=True)
