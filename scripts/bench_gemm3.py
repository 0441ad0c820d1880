"""v1 (128^2 register-staged) vs DMA (256^2 LDS-DMA ring) GEMM on the bench shapes; batch timing, random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
shapes = [(0, 8192, 2304, 768), (0, 8192, 768, 768), (0, 8192, 3072, 768), (0, 8192, 768, 3072), (0, 32768, 1536, 512),
          (0, 32768, 3072, 512), (0, 32768, 512, 3072), (0, 32768, 512, 512), (1, 8192, 768, 3072), (1, 8192, 3072, 768),
          (1, 32768, 512, 3072), (1, 32768, 3072, 512), (2, 3072, 768, 8192), (2, 768, 3072, 8192), (2, 3072, 512, 32768),
          (2, 2304, 768, 8192), (0, 3200, 3072, 768), (0, 11392, 3072, 768), (0, 12800, 3072, 512), (0, 4096, 4096, 4096)]
for layout, M, N, K in shapes:
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
    atomic = layout == 2
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if atomic else torch.bfloat16)
    flags = (hip.OUT_F32 | hip.ATOMIC) if atomic else 0
    res = []
    for impl in ("v1", "dma"):
        kw = {} if impl == "v1" else {"impl": "dma"}
        os.environ["MH_GEMM_DMA"] = "0"
        f = lambda: hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, flags, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append((ms, 2 * M * N * K / ms / 1e9))
    print(f"layout={layout} M={M:6d} N={N:5d} K={K:6d}: v1 {res[0][0]*1e3:7.1f} us {res[0][1]:6.1f} TF | dma {res[1][0]*1e3:7.1f} us {res[1][1]:6.1f} TF | x{res[0][0]/res[1][0]:.2f}", flush=True)
