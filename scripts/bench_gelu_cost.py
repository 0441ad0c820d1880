"""What the GELU arithmetic costs inside the fc1 epilogue: the same GEMM with a plain bias epilogue, with GELU, and with GELU + the
saved derivative (byte-coded), isolated."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((8192, 3072, 768), (3200, 3072, 768), (11392, 3072, 768), (32768, 3072, 512), (12800, 3072, 512)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.empty(M, N, device=dev, dtype=torch.uint8)
    t = [timeit(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.BIAS, bias=bias)),
         timeit(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.BIAS | hip.GELU, bias=bias)),
         timeit(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8, bias=bias, aux_out=aux, ldaux=N))]
    fl = 2.0 * M * N * K
    print(f"({M},{N},{K}) bias {t[0]:6.1f} us {fl/t[0]/1e6:5.0f} TF | + GELU {t[1]:6.1f} us {fl/t[1]/1e6:5.0f} TF | + saved derivative {t[2]:6.1f} us {fl/t[2]/1e6:5.0f} TF", flush=True)
