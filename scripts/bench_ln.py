"""LayerNorm forward / backward at the step's shapes: DEVICE time per launch (50 launches captured into one hipGraph, so that the
host's ~10 us per ctypes launch does not bound the measurement) and achieved GB/s on the algorithmic bytes; beside them a
plain device-to-device copy moving the same number of bytes (what the memory system gives a dependence-free stream of that size;
buffers of this size stay resident in the 256 MiB Infinity Cache, as they do in the step)."""
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
for rows, E in ((8192, 768), (3200, 768), (11392, 768), (32768, 512), (12800, 512)):
    x = torch.randn(rows, E, device=dev); dy16 = torch.randn(rows, E, device=dev).bfloat16(); dres = torch.randn(rows, E, device=dev)
    g = torch.randn(E, device=dev); b = torch.randn(E, device=dev); mean = torch.randn(rows, device=dev); rstd = torch.rand(rows, device=dev) + 0.5
    dx = torch.empty(rows, E, device=dev); dx16 = torch.empty(rows, E, device=dev, dtype=torch.bfloat16); y16 = torch.empty(rows, E, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(1, hip.layernorm_bwd_workspace(rows, E)), device=dev)
    usb = t(lambda: hip.layernorm_bwd_partial(dy16, rows, 0, x, rows, 0, g, mean, rstd, dres, dx, dx16, ws, 1, rows, E))
    usf = t(lambda: hip.layernorm_fwd(x, rows, 0, g, b, y16, rows, 0, mean, rstd, 1, rows, E))
    nb, nf = rows * E * 16, rows * E * 6
    ca, cb = torch.empty(nb // 8, device=dev), torch.empty(nb // 8, device=dev)
    fa, fb = torch.empty(nf // 8, device=dev), torch.empty(nf // 8, device=dev)
    cpb, cpf = t(lambda: cb.copy_(ca)), t(lambda: fb.copy_(fa))
    print(f"({rows},{E}) bwd {usb:6.1f} us {nb / usb / 1e3:5.0f} GB/s (copy of {nb / 1e6:.0f} MB: {cpb:5.1f} us) || fwd {usf:5.1f} us {nf / usf / 1e3:5.0f} GB/s "
          f"(copy of {nf / 1e6:.0f} MB: {cpf:5.1f} us)", flush=True)
