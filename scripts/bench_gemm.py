"""Micro-benchmark of mh_gemm_bf16 on the transformer shapes of the bench configs (prints TFLOP/s)."""
import sys, torch
sys.path.insert(0, ".")
from maestro_amd import hip

dev = torch.device("cuda:0")
shapes = [  # (layout, M, N, K, tag)
    (0, 8192, 2304, 768, "enc qkv fwd"), (0, 8192, 768, 768, "enc proj fwd"), (0, 8192, 3072, 768, "enc mlp1 fwd"),
    (0, 8192, 768, 3072, "enc mlp2 fwd"), (0, 32768, 1536, 512, "dec qkv fwd"), (0, 32768, 3072, 512, "dec mlp1 fwd"),
    (0, 32768, 512, 3072, "dec mlp2 fwd"), (1, 8192, 768, 3072, "enc mlp1 dgrad"), (1, 32768, 512, 3072, "dec mlp1 dgrad"),
    (2, 3072, 768, 8192, "enc mlp1 wgrad"), (2, 3072, 512, 32768, "dec mlp1 wgrad"), (2, 2304, 768, 8192, "enc qkv wgrad"),
    (0, 4096, 4096, 4096, "square 4k"),
]
for layout, M, N, K, tag in shapes:
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
    atomic = layout == 2
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if atomic else torch.bfloat16)
    flags = (hip.OUT_F32 | hip.ATOMIC) if atomic else 0
    run = lambda: hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, flags)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{tag:18s} layout={layout} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
