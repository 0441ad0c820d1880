"""Patchify at the step's raster shapes: device time per launch (50 launches in one hipGraph) and GB/s on the algorithmic bytes
(fp32 image read + bf16 columns + fp32 target written)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
for BD, C, S, P, nbands in ((32, 4, 512, 16, (1, 3)), (32, 3, 512, 16, (3,)), (32, 4, 256, 16, (1, 3)), (32, 2, 256, 32, (2,)), (96, 10, 12, 2, (4, 4, 2))):
    img = torch.rand(BD, C, S, S, device=dev)
    g = S // P; K = C * P * P; Kpad = (K + 31) // 32 * 32
    cols = torch.empty(BD * g * g, Kpad, device=dev, dtype=torch.bfloat16)
    target = torch.empty(BD * g * g, K, device=dev)
    nb = torch.tensor(nbands, dtype=torch.int32, device=dev)
    us = t(lambda: hip.patchify(img, cols, target, BD, C, S, P, Kpad, nb, len(nbands), True, False))
    nbytes = img.numel() * 4 + cols.numel() * 2 + target.numel() * 4
    print(f"BD={BD} C={C} S={S} P={P}: {us:6.1f} us, {nbytes / 1e6:.0f} MB, {nbytes / us / 1e3:5.0f} GB/s", flush=True)
