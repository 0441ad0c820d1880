import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
for layout, M, N, K, f32 in [(2, 4096, 4096, 4096, True), (2, 4096, 4096, 8192, True), (2, 3072, 3072, 32768, True), (1, 4096, 4096, 4096, False),
                             (0, 4096, 4096, 4096, False), (2, 4096, 4096, 4096, False)]:
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    flags = hip.OUT_F32 if f32 else 0
    out = []
    for impl in ("v1", "dma"):
        os.environ["MH_GEMM_DMA"] = "0"
        kw = {} if impl == "v1" else {"impl": "dma"}
        f = lambda: hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, flags, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out.append(2 * M * N * K / ms / 1e9)
    print(f"layout={layout} M={M} N={N} K={K} f32out={f32}: v1 {out[0]:7.1f} TF | dma {out[1]:7.1f} TF", flush=True)
