"""A/B of the 256x256 LDS-DMA tile's main loop inside one process: lockstep (tile MH_TILE_DMA_256_LOCKSTEP) vs waves 4-7
half a K step behind waves 0-3 (MH_TILE_DMA_256), interleaved rounds; plain epilogue.  (Round 2 also A/B-ed the grouped
weight-gradient launch through an environment switch inside the library; the library reads no environment any more and
the grouped launch only has the staggered form.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(f, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def ab(f, rounds=5):
    res = {hip.TILE_DMA_256_LOCKSTEP: [], hip.TILE_DMA_256: []}
    for k in res:
        f(k); f(k)
    for _ in range(rounds):
        for k in res:
            res[k].append(timeit(lambda: f(k)))
    return min(res[hip.TILE_DMA_256_LOCKSTEP]), min(res[hip.TILE_DMA_256])
shapes = [(0, 8192, 3072, 768), (0, 8192, 768, 3072), (0, 8192, 2304, 768), (0, 11392, 3072, 768), (0, 32768, 3072, 512), (0, 32768, 512, 3072),
          (0, 32768, 1536, 512), (1, 32768, 3072, 512), (1, 32768, 512, 3072), (1, 8192, 768, 3072), (0, 16384, 4096, 4096), (0, 8192, 8192, 8192)]
for lay, M, N, K in shapes:
    A = torch.randn(M, K).bfloat16().to(dev)
    B = (torch.randn(N, K) if lay == 0 else torch.randn(K, N)).bfloat16().to(dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda tile: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], C, N, tile=tile)
    t0, t1 = ab(f)
    r = timeit(lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], C, N, tile=hip.TILE_REG_128))
    fl = 2.0 * M * N * K
    print(f"{'NT' if lay == 0 else 'NN'} ({M:5d},{N:4d},{K:4d}) d256 lockstep {t0:7.1f} us {fl/t0/1e6:6.0f} TF | staggered {t1:7.1f} us {fl/t1/1e6:6.0f} TF ({t0/t1:4.2f}x) | reg128 {r:7.1f} us {fl/r/1e6:6.0f} TF", flush=True)
