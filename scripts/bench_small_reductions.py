"""Device time (50 launches per hipGraph) of the small reduction kernels of the backward at the C3 step's shapes."""
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
for B, L, Dd in ((32, 1024, 512), (32, 400, 512)):
    dx = torch.randn(B, L, Dd, device=dev); mask = (torch.rand(B, L, device=dev) < 0.75).to(torch.uint8)
    slot = torch.zeros(L, dtype=torch.int32, device=dev); out = torch.zeros(Dd, device=dev)
    us = t(lambda: hip.unmask_token_grad(dx, mask, slot, out, B, L, Dd, 0, 0, L))
    nb = float(mask.sum()) * Dd * 4
    print(f"unmask_token_grad B={B} L={L}: {us:6.1f} us  {nb / us / 1e3:5.0f} GB/s", flush=True)
for M, N, f32 in ((32768, 1024, False), (32768, 768, False), (32768, 512, False), (12800, 768, False), (8192, 768, True)):
    x = torch.randn(M, N, device=dev).to(torch.float32 if f32 else torch.bfloat16); out = torch.zeros(N, device=dev)
    us = t(lambda: hip.colsum(x, out, M, N, N))
    print(f"colsum ({M},{N}) {'f32' if f32 else 'bf16'}: {us:6.1f} us  {x.numel() * x.element_size() / us / 1e3:5.0f} GB/s", flush=True)
for B, D, L, E in ((32, 1, 1024, 768), (32, 16, 25, 768)):
    y = torch.randn(B * D * L, E, device=dev); dxg = torch.randn(B, D * L, E, device=dev); gamma = torch.randn(E, device=dev)
    stats = torch.rand(B * D, 2, device=dev) + 0.5; dyc = torch.empty(B * D * L, E, device=dev, dtype=torch.bfloat16)
    dg, db, sums = torch.zeros(E, device=dev), torch.zeros(E, device=dev), torch.zeros(B * D, 2, device=dev)
    us = t(lambda: hip.embed_finish_bwd(dxg, y, stats, gamma, dyc, dg, db, sums, B, D, L, E, 0, D * L))
    print(f"embed_finish_bwd (stats + apply) B={B} D={D} L={L} E={E}: {us:6.1f} us  {y.numel() * (16 + 2) / us / 1e3:5.0f} GB/s", flush=True)
