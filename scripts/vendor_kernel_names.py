"""Reference point, not product code: which kernels (tile, split) the vendor library behind torch.mm picks for the step's long-K, narrow-N GEMM
shapes -- run under `rocprofv3 --kernel-trace --stats`; the kernel names carry the macro tile (MT..) and the global split (GSU..)."""
import torch

dev = torch.device("cuda:0")
for (M, N, K) in ((8192, 768, 3072), (8192, 768, 2304), (3200, 768, 3072), (3200, 3072, 768), (11392, 768, 3072), (32768, 512, 3072),
                  (8192, 768, 768)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    Wt = W.t().contiguous()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(7):
        torch.mm(A, W.t(), out=C)
    torch.cuda.synchronize()
    for _ in range(5):
        torch.mm(A, Wt, out=C)
    torch.cuda.synchronize()
    print(M, N, K, "NT x7, NN x5", flush=True)
