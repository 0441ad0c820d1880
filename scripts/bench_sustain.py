"""Does a GEMM keep its burst rate (10 launches, hot operands) when sustained (100+ ms) and with rotating cold operands?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
M, N, K = 8192, 3072, 768
NB = 24
As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NB)]
Bs = [torch.randn(N, K, device=dev).bfloat16() for _ in range(NB)]
Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
def run(tile, iters, rotate):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        j = i % NB if rotate else 0
        hip.gemm(0, M, N, K, As[j], K, Bs[j], K, Cs[j], N, 0, tile=tile)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9
for tile, name in ((0, "reg128"), (1, "d256"), (2, "d256x128")):
    run(tile, 5, False)
    print(name, " ".join(f"{lbl} {run(tile, it, rot):6.0f}TF" for lbl, it, rot in (("burst10-hot", 10, False), ("sustain3000-hot", 3000, False),
          ("burst24-cold", 24, True), ("sustain3000-cold", 3000, True), ("burst10-hot-again", 10, False))), flush=True)
