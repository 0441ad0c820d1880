"""Where is the operand-ingest limit of the 128x128 GEMM?  lda = 0 / ldb = 0 make every row of an operand tile the SAME 128-byte
line per K step (always an L1 hit after the first touch): if the kernel gets much faster, the limit sits behind the L1 (L2 -> L1
fill / miss handling) and L1 sharing between co-resident workgroups would pay; if not, it is the CU's own load path / LDS stores."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(8192, 3072, 768), (8192, 768, 3072), (32768, 3072, 512)]:
    A = torch.randn(M, K).bfloat16().to(dev); W = (torch.randn(N, K) / K ** 0.5).bfloat16().to(dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for tile, name in ((hip.TILE_REG_128, "reg128"), (hip.TILE_PP_128, "pp128"), (hip.TILE_DMA_256, "dma256")):
        out = []
        for lda, ldb in ((K, K), (0, K), (K, 0), (0, 0)):
            best = min(timeit(lambda: hip.gemm(0, M, N, K, A, lda, W, ldb, C, N, 0, tile=tile)) for _ in range(3))
            out.append(f"lda={lda:4d} ldb={ldb:4d}: {best:6.1f} us")
        print(f"({M},{N},{K}) {name}: " + " | ".join(out), flush=True)
