import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
layout, M, N, K = (int(x) for x in sys.argv[1:5])
A = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
B = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(30):
    hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, 0)
torch.cuda.synchronize()
