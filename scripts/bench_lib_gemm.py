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
# fp8 (OCP e4m3, per-tensor scales) through the vendor library: what the C5 "fp8 path" could buy on the same shapes
try:
    one = torch.ones((), device=dev)
    for (M, N, K) in ((8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (32768, 3072, 512), (32768, 512, 3072)):
        A8 = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn); W8 = torch.randn(N, K, device=dev).to(torch.float8_e4m3fn)
        f8 = t(lambda: torch._scaled_mm(A8, W8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
        Ab = torch.randn(M, K, device=dev).bfloat16(); Wb = torch.randn(N, K, device=dev).bfloat16()
        b16 = t(lambda: torch.matmul(Ab, Wb.t()))
        print(f"({M},{N},{K}) lib fp8 NT {f8*1e3:6.1f}us {2.0*M*N*K/f8/1e9:5.0f}TF | lib bf16 NT {b16*1e3:6.1f}us {2.0*M*N*K/b16/1e9:5.0f}TF", flush=True)
except Exception as exc:  # noqa: BLE001
    print("fp8 library GEMM not available here:", type(exc).__name__, str(exc)[:200])
