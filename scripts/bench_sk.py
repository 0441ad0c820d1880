"""Round 6: the stream-K GEMM (csrc/gemm_sk.hip) against the library's rule (MH_TILE_AUTO) on the step's long-K / narrow-N signatures,
each with the epilogue the step runs it with; isolated, two ways (as scripts/r05_gap.py):
  hot   ONE buffer set relaunched back to back (operands and outputs stay in the Infinity Cache)
  cold  ROTATING buffer sets (footprint >= 1.5 GiB): as inside the step, where every layer owns its activations and weights
plus a THROUGHPUT column: the same signature at 8 x M (the tile count no longer matters: what a CU-second of this kernel buys).
Interleaved rounds, min over rounds, HIP events around runs of launches.  us per launch; TFLOP/s for the throughput column."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
SHAPES = [("fc2", 0, 8192, 768, 3072, F32), ("dfc1", 1, 8192, 768, 3072, 0), ("dqkv", 1, 8192, 768, 2304, 0),
          ("proj", 0, 8192, 768, 768, F32), ("dproj", 1, 8192, 768, 768, 0),
          ("s2 fc2", 0, 3200, 768, 3072, F32), ("s2 dfc1", 1, 3200, 768, 3072, 0), ("s2 dqkv", 1, 3200, 768, 2304, 0),
          ("s2 proj", 0, 3200, 768, 768, F32),
          ("jnt fc2", 0, 11392, 768, 3072, F32), ("jnt dfc1", 1, 11392, 768, 3072, 0),
          ("dec fc2", 0, 32768, 512, 3072, F32), ("dec dfc1", 1, 32768, 512, 3072, 0), ("dec dqkv", 1, 32768, 512, 1536, 0),
          ("ds2 fc2", 0, 12800, 512, 3072, F32), ("ds2 dfc1", 1, 12800, 512, 3072, 0)]
TILES = [("auto", None), ("skd256", hip.TILE_SK_DMA_256), ("sk192", hip.TILE_SK_192), ("sk256", hip.TILE_SK_256)]
if "--dma-only" in sys.argv:
    TILES = TILES[:2]
only = [a for a in sys.argv[1:] if not a.startswith("-")]
throughput = "--no-throughput" not in sys.argv


def make(lay, M, N, K, fl, tile):  # noqa: N803
    A = torch.randn(M, K, device=dev).bfloat16()  # noqa: N806
    B = ((torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)) / K ** 0.5).bfloat16()  # noqa: N806
    out = torch.empty(M, N, dtype=torch.float32 if fl & hip.OUT_F32 else torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev) if fl & hip.BIAS else None
    res = torch.randn(M, N, device=dev) if fl & hip.RESIDUAL else None
    nbytes = sum(t.numel() * t.element_size() for t in (A, B, out, res) if t is not None)
    return (lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, bias=bias, res=res, ldr=N if res is not None else 0,
                             tile=tile)), nbytes


def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':9s} {'lay':3s} {'(M, N, K)':20s} | " + " | ".join(f"{n:>6s} hot   cold" for n, _ in TILES) +
      (" | 8 x M: " + " ".join(f"{n:>6s}" for n, _ in TILES) + " TFLOP/s" if throughput else ""), flush=True)
for name, lay, M, N, K, fl in SHAPES:
    if only and not any(o == name or (o.endswith("*") and name.startswith(o[:-1])) for o in only):
        continue
    sets = {}
    for tn, tile in TILES:
        f0, b0 = make(lay, M, N, K, fl, tile)
        r = max(2, int(1.5 * 2 ** 30 / b0) + 1)
        sets[tn] = [f0] + [make(lay, M, N, K, fl, tile)[0] for _ in range(r - 1)]
        for f in sets[tn]:
            f()
    res = {tn: ([], []) for tn, _ in TILES}
    for _ in range(5):
        for tn, _t in TILES:
            res[tn][0].append(timeit(sets[tn][:1], 16))
            res[tn][1].append(timeit(sets[tn], 2 * len(sets[tn])))
    line = f"{name:9s} {'NT' if lay == 0 else 'NN':3s} ({M:5d},{N:5d},{K:5d})  | " + " | ".join(
        f"{min(res[tn][0]):10.1f} {min(res[tn][1]):6.1f}" for tn, _ in TILES)
    del sets
    torch.cuda.empty_cache()
    if throughput:
        big = {}
        for tn, tile in TILES:
            f, _b = make(lay, 8 * M, N, K, fl, tile)
            f()
            big[tn] = f
        tf = {tn: [] for tn, _ in TILES}
        for _ in range(3):
            for tn, _t in TILES:
                tf[tn].append(timeit([big[tn]], 6))
        line += " |        " + " ".join(f"{2.0 * 8 * M * N * K / min(tf[tn]) / 1e6:6.0f}" for tn, _ in TILES)
        del big
        torch.cuda.empty_cache()
    print(line, flush=True)
