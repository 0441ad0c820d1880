"""Reference point, not product code: what the vendor library behind torch.matmul (hipBLASLt / rocBLAS) reaches on the step's GEMM shapes on
this GPU, plain bf16 output, no epilogue -- to tell how far the hand-written kernels' MAIN LOOPS are from what the hardware is known to do."""
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


for (M, N, K) in ((8192, 3072, 768), (8192, 2304, 768), (8192, 768, 768), (8192, 768, 3072), (3200, 3072, 768), (3200, 768, 3072),
                  (11392, 3072, 768), (32768, 3072, 512), (32768, 512, 3072), (32768, 1536, 512), (8192, 8192, 8192)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    Wt = W.t().contiguous()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    nt_torch = t(lambda: torch.mm(A, W.t(), out=C))
    nn_torch = t(lambda: torch.mm(A, Wt, out=C))
    nt_ours = t(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, 0))
    nn_ours = t(lambda: hip.gemm(1, M, N, K, A, K, Wt, N, C, N, 0))
    print(f"({M},{N},{K}): NT torch {nt_torch:7.1f} us {fl / nt_torch / 1e6:5.0f} TF | ours {nt_ours:7.1f} us {fl / nt_ours / 1e6:5.0f} TF || "
          f"NN torch {nn_torch:7.1f} us {fl / nn_torch / 1e6:5.0f} TF | ours {nn_ours:7.1f} us {fl / nn_ours / 1e6:5.0f} TF", flush=True)
