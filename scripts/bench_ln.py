"""LayerNorm forward / backward at the step's shapes: microseconds and achieved GB/s (algorithmic bytes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, E in ((8192, 768), (3200, 768), (11392, 768), (32768, 512), (12800, 512)):
    x = torch.randn(rows, E, device=dev); dy16 = torch.randn(rows, E, device=dev).bfloat16(); dres = torch.randn(rows, E, device=dev)
    g = torch.randn(E, device=dev); b = torch.randn(E, device=dev); mean = torch.randn(rows, device=dev); rstd = torch.rand(rows, device=dev) + 0.5
    dx = torch.empty(rows, E, device=dev); y16 = torch.empty(rows, E, device=dev, dtype=torch.bfloat16)
    dg, db, dc = torch.zeros(E, device=dev), torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    out = []
    ws = torch.empty(max(1, hip.layernorm_bwd_workspace(rows, E)), device=dev)
    us = t(lambda: hip.layernorm_bwd(dy16, rows, 0, x, rows, 0, g, mean, rstd, dres, dx, None, dg, db, dc, ws, 1, rows, E))
    out.append(f"{us:6.1f}us {rows * E * 14 / us / 1e3:5.0f}GB/s")
    usf = t(lambda: hip.layernorm_fwd(x, rows, 0, g, b, y16, rows, 0, mean, rstd, 1, rows, E))
    print(f"({rows},{E}) bwd(+reduce): " + " | ".join(out) + f" || fwd {usf:5.1f}us {rows * E * 6 / usf / 1e3:5.0f}GB/s", flush=True)
