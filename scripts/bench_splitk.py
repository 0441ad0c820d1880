"""Split-K with the in-kernel fix-up (mh_gemm_bf16_ws) against the unsplit launch on the few-tile / long-K shapes of the C4 (ViT-L) and C3 steps:
us per launch (20 launches per hipGraph), plain bf16 output and the fp32 + bias + residual epilogue.  python scripts/bench_splitk.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


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


for (M, N, K) in ((576, 1024, 4096), (1152, 1024, 4096), (1792, 1024, 4096), (576, 1024, 3072), (1792, 1024, 3072), (576, 1024, 1024), (1792, 1024, 1024),
                  (576, 4096, 1024), (3200, 768, 3072), (3200, 768, 2304), (3200, 768, 768), (2304, 512, 4096), (4608, 512, 4096), (3520, 1024, 4096)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    Wt = W.t().contiguous()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Cf, res, bias = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    out = []
    for name, fn in (("NT bf16", lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, 0)),
                     ("NN bf16", lambda: hip.gemm(1, M, N, K, A, K, Wt, N, C, N, 0)),
                     ("NT f32+res", lambda: hip.gemm(0, M, N, K, A, K, W, K, Cf, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res, ldr=N))):
        os.environ["MH_GEMM_SPLITK"] = "0"
        t0 = t(fn)
        os.environ["MH_GEMM_SPLITK"] = "1"
        t1 = t(fn)
        out.append(f"{name} {t0:6.1f} -> {t1:6.1f} us ({100 * (t1 / t0 - 1):+5.1f} %, {fl / t1 / 1e6:4.0f} TF)")
    print(f"({M},{N},{K}): " + " | ".join(out), flush=True)
