"""Attention forward / backward with ROTATING buffer sets (qkv, out, dout, dqkv cold in HBM, as inside the step) and with one
relaunched set (hot); us per call.  python scripts/bench_attn_cold.py"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timed(fns, n):
    for f in fns[:2]: f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(n): fns[i % len(fns)]()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for (B, N, H, D) in [(32, 256, 12, 64), (32, 356, 12, 64), (32, 100, 12, 64), (32, 1024, 16, 32), (32, 400, 16, 32)]:
    nbytes = B * N * H * D * 2 * 8
    R = max(2, int(1.2 * 2 ** 30 / nbytes) + 1)
    sets = []
    g = torch.Generator().manual_seed(1)
    for _ in range(R):
        qkv = torch.randn(B, N, 3, H, D, generator=g).to(torch.bfloat16).to(dev)
        out, dout = torch.empty(B, N, H * D, dtype=torch.bfloat16, device=dev), torch.randn(B, N, H * D, generator=g).to(torch.bfloat16).to(dev)
        lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
        dqkv = torch.empty_like(qkv)
        hip.attn_fwd(qkv, out, lse, B, N, H, D, D ** -0.5)
        sets.append((qkv, out, dout, lse, delta, dqkv))
    fw = [(lambda t=t: hip.attn_fwd(t[0], t[1], t[3], B, N, H, D, D ** -0.5)) for t in sets]
    bw = [(lambda t=t: hip.attn_bwd(t[0], t[1], t[2], t[3], t[4], t[5], B, N, H, D, D ** -0.5)) for t in sets]
    r = {k: [] for k in ("fh", "fc", "bh", "bc")}
    for _ in range(4):
        r["fh"].append(timed(fw[:1], 20)); r["fc"].append(timed(fw, 2 * R)); r["bh"].append(timed(bw[:1], 20)); r["bc"].append(timed(bw, 2 * R))
    m = {k: min(v) for k, v in r.items()}
    print(f"B {B} N {N:5d} H {H} D {D}: fwd hot {m['fh']:7.1f} cold {m['fc']:7.1f} | bwd hot {m['bh']:7.1f} cold {m['bc']:7.1f} us", flush=True)
    del sets, fw, bw; torch.cuda.empty_cache()
