"""mask_select at the step's shapes: device time per launch (50 launches in one hipGraph)."""
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
for B, L in ((32, 1024), (32, 400), (32, 576), (64, 1024), (32, 2000)):
    k = round(0.75 * L)
    noise = torch.rand(B, L, device=dev); struct = (torch.rand(B, L, device=dev) < 0.3).to(torch.uint8)
    vis = torch.zeros(B, L - k, dtype=torch.int32, device=dev); msk = torch.zeros(B, k, dtype=torch.int32, device=dev)
    inv = torch.zeros(B, L, dtype=torch.int32, device=dev); mask = torch.zeros(B, L, dtype=torch.uint8, device=dev)
    print(f"B={B} L={L}: {t(lambda: hip.mask_select(noise, struct, vis, msk, inv, mask, B, L, k)):6.1f} us", flush=True)
