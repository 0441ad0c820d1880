"""Round 5 probe (no library change): would a 192(N) x 128(M) tiling -- 256 tiles, one per CU, on the (8192, 768, K) outputs -- beat the
128 x 128 tiling (384 tiles on 512 slots)?  The existing 192 x 128 register tile computes C^T = W X^T when the operands are swapped, which
has exactly that tile grid; main loop and balance are the real thing, only the output lands transposed.  Hot (one buffer set) and cold
(rotating sets, >= 1.5 GiB), min over 5 rounds, us."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL


def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def make(M, N, K, swapped, tile, fl):  # noqa: N803
    X = torch.randn(M, K, device=dev).bfloat16()  # noqa: N806
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()  # noqa: N806
    f32 = bool(fl & hip.OUT_F32)
    if swapped:
        out = torch.empty(N, M, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        bias = torch.randn(M, device=dev)
        res = torch.randn(N, M, device=dev) if fl & hip.RESIDUAL else None
        return lambda: hip.gemm(0, N, M, K, W, K, X, K, out, M, fl, bias=bias if fl & hip.BIAS else None, res=res,
                                ldr=M if res is not None else 0, tile=tile)
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if fl & hip.RESIDUAL else None
    return lambda: hip.gemm(0, M, N, K, X, K, W, K, out, N, fl, bias=bias if fl & hip.BIAS else None, res=res,
                            ldr=N if res is not None else 0, tile=tile)


SHAPES = ((8192, 768, 3072), (8192, 768, 768), (3200, 768, 3072), (3200, 768, 768), (11392, 768, 3072), (11392, 768, 768))
if len(sys.argv) > 1 and sys.argv[1] == 'wide':   # the wide outputs: is the 192-wide tile's lower operand traffic per FLOP worth anything where 128 x 128 already balances?
    SHAPES = ((8192, 3072, 768), (8192, 2304, 768), (32768, 3072, 512), (32768, 1536, 512), (11392, 3072, 768))
for (M, N, K) in SHAPES:
    for fl, fname in ((0, "plain"), (F32, "f32+res")):
        row = []
        cands = [("auto", False, None), ("reg128", False, hip.TILE_REG_128), ("pp128", False, hip.TILE_PP_128),
                 ("swap192 (C^T)", True, hip.TILE_REG_192)]
        if fl and hasattr(hip, "TILE_REG_N192"):   # the real N-long tiles (fp32 output): only in the experiment's build (not kept)
            cands += [("n192", False, hip.TILE_REG_N192), ("n96", False, hip.TILE_REG_N96)]
        for name, swapped, tile in cands:
            nbytes = (M * K + N * K) * 2 + M * N * (4 if fl else 2) * (2 if fl else 1)
            r = max(2, int(1.5 * 2 ** 30 / nbytes) + 1)
            fns = [make(M, N, K, swapped, tile, fl) for _ in range(r)]
            for f in fns:
                f()
            hot = min(timeit(fns[:1], 16) for _ in range(5))
            cold = min(timeit(fns, 2 * len(fns)) for _ in range(5))
            row.append(f"{name} {hot:.1f}/{cold:.1f}")
            del fns
            torch.cuda.empty_cache()
        print(f"({M},{N},{K}) {fname:8s} hot/cold us: " + " | ".join(row), flush=True)
