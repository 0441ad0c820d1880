"""A/B aid: time mh_attn_fwd / mh_attn_bwd of whatever library maestro_amd.hip loads (scripts/ab_lib.py swaps it) on the step's shapes."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timed(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for (B, N, H, D) in [(32, 1024, 16, 32), (32, 400, 16, 32), (32, 256, 12, 64), (32, 356, 12, 64), (32, 100, 12, 64)]:
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, N, 3, H, D, generator=g).to(torch.bfloat16).to(dev)
    out, dout = torch.empty(B, N, H * D, dtype=torch.bfloat16, device=dev), torch.randn(B, N, H * D, generator=g).to(torch.bfloat16).to(dev)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    tf = timed(lambda: hip.call("mh_attn_fwd", qkv, out, lse, hip._I(B), hip._I(N), hip._I(H), hip._I(D), hip._F(D ** -0.5)))
    tb = timed(lambda: hip.call("mh_attn_bwd", qkv, out, dout, lse, delta, dqkv, hip._I(B), hip._I(N), hip._I(H), hip._I(D), hip._F(D ** -0.5)))
    print(f"N {N} D {D}: fwd {tf:7.1f} us  bwd {tb:7.1f} us", flush=True)
