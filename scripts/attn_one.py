"""One attention shape, forward + backward, 10 times (for rocprofv3 --pmc passes): python scripts/attn_one.py B N H D"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
B, N, H, D = (int(x) for x in sys.argv[1:5])
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, N, 3, H, D, generator=g).to(torch.bfloat16).to(dev)
out, dout = torch.empty(B, N, H * D, dtype=torch.bfloat16, device=dev), torch.randn(B, N, H * D, generator=g).to(torch.bfloat16).to(dev)
lse, delta, dqkv = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev), torch.empty_like(qkv)
for _ in range(10):
    hip.attn_fwd(qkv, out, lse, B, N, H, D, D ** -0.5)
    hip.attn_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, D, D ** -0.5)
torch.cuda.synchronize()
