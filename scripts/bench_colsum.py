"""Column-sum kernel on the step's bias-gradient shapes (bf16 [M, N] -> f32 [N])."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
for (M, N) in ((32768, 1024), (32768, 768), (8192, 512), (12800, 200)):
    x = torch.randn(M, N, device=dev).bfloat16(); out = torch.zeros(N, device=dev)
    for _ in range(3): hip.colsum(x, out, M, N, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.colsum(x, out, M, N, N)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"({M},{N}) {us:6.1f}us {M * N * 2 / us / 1e3:6.0f} GB/s", flush=True)
