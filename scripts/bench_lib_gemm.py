"""Reference point only (not used by the product): torch.matmul (hipBLASLt / rocBLAS) on the step's GEMM shapes."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in ((8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (3200, 768, 3072), (11392, 3072, 768),
                  (32768, 1536, 512), (32768, 3072, 512), (32768, 512, 3072), (12800, 3072, 512)):
    A = torch.randn(M, K, device=dev).bfloat16(); Wt = torch.randn(N, K, device=dev).bfloat16(); Wn = torch.randn(K, N, device=dev).bfloat16()
    nt = t(lambda: torch.matmul(A, Wt.t())); nn = t(lambda: torch.matmul(A, Wn))
    fl = 2.0 * M * N * K
    print(f"({M},{N},{K}) lib NT {nt*1e3:6.1f}us {fl/nt/1e9:5.0f}TF | lib NN {nn*1e3:6.1f}us {fl/nn/1e9:5.0f}TF", flush=True)
# wgrad-shaped TN: dW[M,N] = dY[K,M]^T X[K,N]
for (M, N, K) in ((768, 3072, 8192), (3072, 768, 8192), (512, 3072, 32768)):
    dY = torch.randn(K, M, device=dev).bfloat16(); X = torch.randn(K, N, device=dev).bfloat16()
    tn = t(lambda: torch.matmul(dY.t(), X))
    print(f"TN ({M},{N},{K}) lib {tn*1e3:6.1f}us {2.0*M*N*K/tn/1e9:5.0f}TF", flush=True)
