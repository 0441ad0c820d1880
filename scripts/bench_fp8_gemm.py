"""fp8 (scaled-MFMA, 256x256x128 tile) vs bf16 kernels on the forward GEMM shapes of the C5 / C3 steps (isolated, random data)."""
import os, sys, torch
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
one = torch.ones(1, device=dev)
for (M, N, K) in ((512, 768, 768), (512, 3072, 768), (1152, 768, 3072), (1152, 768, 768), (2048, 512, 3072), (4608, 512, 3072), (4608, 2304, 768), (4608, 768, 768), (4608, 768, 3072), (1152, 3072, 768), (512, 768, 3072), (10880, 3072, 768), (8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 768), (4608, 3072, 768), (18432, 3072, 512),
                  (32768, 3072, 512), (32768, 512, 3072), (32768, 1536, 512), (16384, 4096, 4096)):
    a, b = torch.randn(M, K), torch.randn(N, K)
    A8, B8 = a.to(torch.float8_e4m3fn).view(torch.uint8).to(dev), b.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    A16, B16 = a.bfloat16().to(dev), b.bfloat16().to(dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    os.environ["MH_FP8_TILE"] = "128"
    t8s = timeit(lambda: hip.gemm_fp8(M, N, K, A8, K, B8, K, C, N, one, one))
    os.environ["MH_FP8_TILE"] = "128d"
    t8d = timeit(lambda: hip.gemm_fp8(M, N, K, A8, K, B8, K, C, N, one, one))
    os.environ["MH_FP8_TILE"] = "256"
    t8 = timeit(lambda: hip.gemm_fp8(M, N, K, A8, K, B8, K, C, N, one, one))
    os.environ.pop("MH_FP8_TILE")
    t16 = timeit(lambda: hip.gemm(0, M, N, K, A16, K, B16, K, C, N))
    t256 = timeit(lambda: hip.gemm(0, M, N, K, A16, K, B16, K, C, N, tile=hip.TILE_DMA_256))
    print(f"({M:5d},{N:4d},{K:4d}) fp8 128^2 {t8s:7.1f} us {fl/t8s/1e6:6.0f} TF | 4-stage {t8d:7.1f} us {fl/t8d/1e6:6.0f} TF | fp8 256^2 {t8:7.1f} us {fl/t8/1e6:6.0f} TF | bf16 auto {t16:7.1f} us {fl/t16/1e6:6.0f} TF | bf16 d256 {t256:7.1f} us {fl/t256/1e6:6.0f} TF", flush=True)
