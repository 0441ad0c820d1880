"""Where a K step of the persistent LDS-DMA GEMM goes: the full kernel vs DMA-only / reads+MFMA-only / MFMA-only builds
(diagnostic instantiations, garbage outputs), full 256x256 tiles only, one round and several rounds."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((8192, 2048, 768), (16384, 4096, 768), (8192, 2048, 3072), (16384, 4096, 3072)):
    A = torch.randn(M, K).bfloat16().to(dev); B = torch.randn(N, K).bfloat16().to(dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gg = hip.GroupedGemm(0, [dict(A=A, B=B, C=C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)], dev, split=1)
    tiles, fl = gg.n_items, 2.0 * M * N * K
    out = []
    for abl, name in ((0, "full"), (1, "dma only"), (2, "reads+mfma"), (3, "mfma only")):
        f = lambda: hip.call("mh_gemm_grouped_ablate", ctypes.c_int(abl), gg.table, gg.items, ctypes.c_int(gg.n_items), ctypes.c_int(gg.n_workers))
        us = timeit(f)
        out.append(f"{name} {us:7.1f} us ({us / (tiles / 256) / (K / 32) * 1e3:6.0f} ns per K step)")
    hip.gemm(0, M, N, K, A, K, B, K, C, N, tile=hip.TILE_DMA_256)
    us_d = timeit(lambda: hip.gemm(0, M, N, K, A, K, B, K, C, N, tile=hip.TILE_DMA_256))
    print(f"M={M} N={N} K={K} tiles={tiles} ({tiles/256:.1f} rounds) {fl/1e9:.0f} GFLOP: " + " | ".join(out) + f" | d256 launch {us_d:7.1f} us", flush=True)
