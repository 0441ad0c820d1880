"""Every tile on the step's small-M (s2 encoder, M = 3200) and N = 768 long-K shapes, plain bf16 output, NT and NN, against the vendor
library behind torch.matmul (reference point only).  us per launch, 20 launches per hipGraph."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
NAMES = {-1: "auto", 0: "reg128", 1: "dma256", 2: "dma256x128", 3: "dma128x256", 4: "dma128", 5: "dma128x4", 7: "pp128", 13: "reg64", 14: "reg192"}


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (M, N, K) in ((3200, 768, 768), (3200, 2304, 768), (3200, 3072, 768), (3200, 768, 3072), (3200, 768, 2304),
                  (8192, 768, 768), (8192, 768, 3072), (8192, 768, 2304), (11392, 768, 3072), (12800, 512, 3072), (12800, 3072, 512)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    Wt = W.t().contiguous()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for layout, B, ldb, name in ((0, W, K, "NT"), (1, Wt, N, "NN")):
        ref = t((lambda: torch.mm(A, W.t(), out=C)) if layout == 0 else (lambda: torch.mm(A, Wt, out=C)))
        res = {}
        for tile in NAMES:
            try:
                res[NAMES[tile]] = t(lambda: hip.gemm(layout, M, N, K, A, K, B, ldb, C, N, 0, tile=None if tile == -1 else tile))
            except hip.HipExtensionError:
                pass
        best = min((v, k) for k, v in res.items() if k != "auto")
        print(f"({M},{N},{K}) {name}: vendor {ref:6.1f} | auto {res['auto']:6.1f} | best {best[1]} {best[0]:6.1f} ({100 * (best[0] / res['auto'] - 1):+5.1f} %) | "
              + " ".join(f"{k} {v:.1f}" for k, v in res.items() if k != "auto"), flush=True)
