"""Decoder-aerial forward GEMMs (M = 32768) with their real epilogues: which tile wins in isolation?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
names = {0: "reg128", 1: "d256", 2: "d256x128", 3: "d128x256"}
for (N, K, kind) in ((1536, 512, "plain"), (512, 512, "res"), (3072, 512, "gelu"), (512, 3072, "res")):
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev); aux8 = torch.empty(M, N, device=dev, dtype=torch.uint8)
    C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); C32 = torch.empty(M, N, device=dev)
    out = []
    for tile in names:
        if kind == "plain": f = lambda: hip.gemm(0, M, N, K, A, K, W, K, C16, N, 0, tile=tile)
        elif kind == "res": f = lambda: hip.gemm(0, M, N, K, A, K, W, K, C32, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res, ldr=N, tile=tile)
        else: f = lambda: hip.gemm(0, M, N, K, A, K, W, K, C16, N, hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8, bias=bias, aux_out=aux8, ldaux=N, tile=tile)
        ms = t(f)
        out.append(f"{names[tile]} {ms*1e3:6.1f}us {2.0*M*N*K/ms/1e9:5.0f}TF")
    print(f"({M},{N},{K}) {kind:5s} " + " | ".join(out), flush=True)
