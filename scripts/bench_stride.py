"""Does the row stride of the K-minor operands matter (L2 channel spread of a tile's rows)?  Same GEMM, operands allocated with
lda = ldb = K + pad elements; plain bf16 epilogue; interleaved rounds."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(f, n=8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(8192, 3072, 768), (8192, 2304, 768), (8192, 768, 3072), (32768, 3072, 512), (32768, 1536, 512), (32768, 512, 3072), (11392, 3072, 768)]
pads = [0, 8, 32, 64, 128, 192]
for M, N, K in shapes:
    for tile, tname in ((hip.TILE_REG_128, "reg128"), (hip.TILE_DMA_256, "dma256")):
        fs = {}
        for pad in pads:
            ld = K + pad
            A = torch.randn(M, ld).bfloat16().to(dev); W = (torch.randn(N, ld) / K ** 0.5).bfloat16().to(dev)
            for ldc_pad in (0, 64):
                C = torch.empty(M, N + ldc_pad, dtype=torch.bfloat16, device=dev)
                fs[(pad, ldc_pad)] = (lambda A=A, W=W, C=C, ld=ld, ldc=N + ldc_pad: hip.gemm(0, M, N, K, A, ld, W, ld, C, ldc, 0, tile=tile))
        res = {k: [] for k in fs}
        for k, f in fs.items(): f()
        for _ in range(4):
            for k, f in fs.items(): res[k].append(timeit(f))
        line = f"({M:5d},{N:4d},{K:4d}) {tname}:"
        for k in fs:
            mn = min(res[k]); line += f"  pad{k[0]:3d}/c{k[1]:2d} {mn:6.1f}us {2.0*M*N*K/mn/1e6:5.0f}TF"
        print(line, flush=True)
