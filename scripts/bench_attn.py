"""Attention kernels in isolation (the shapes of the C3 / C5 steps): forward and backward time, TFLOP/s on the algorithmic
FLOPs (forward 4 B H N^2 D, backward 10 B H N^2 D).  python scripts/bench_attn.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32, 1024, 16, 32), (32, 400, 16, 32), (32, 576, 16, 32), (32, 256, 12, 64), (32, 100, 12, 64), (32, 356, 12, 64),
          (32, 144, 12, 64), (32, 448, 12, 64), (8, 2048, 16, 64)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for (B, N, H, D) in SHAPES:  # noqa: N806
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, N, 3, H, D, generator=g).to(torch.bfloat16).to(dev)
    out, dout = torch.empty(B, N, H * D, dtype=torch.bfloat16, device=dev), torch.randn(B, N, H * D, generator=g).to(torch.bfloat16).to(dev)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    tf = timed(lambda: hip.attn_fwd(qkv, out, lse, B, N, H, D, D ** -0.5))
    tb = timed(lambda: hip.attn_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, D, D ** -0.5))
    fl = B * H * N * N * D
    line = f"B {B:3d} N {N:5d} H {H:2d} D {D:2d}: fwd {tf:7.1f} us {4 * fl / tf / 1e6:7.1f} TF | bwd two kernels {tb:7.1f} us {10 * fl / tb / 1e6:7.1f} TF"
    print(line, flush=True)
