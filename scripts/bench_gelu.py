"""GEMMs with the GELU / GELU' epilogues at the step's fc1 / fc2-dgrad shapes vs the same GEMMs without epilogue math."""
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
for (M, N, K) in ((8192, 3072, 768), (32768, 3072, 512), (3200, 3072, 768)):
    A = torch.randn(M, K, device=dev).bfloat16(); Wt = torch.randn(N, K, device=dev).bfloat16(); Wn = torch.randn(K, N, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.randn(M, N, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    r = {"NT plain": t(lambda: hip.gemm(0, M, N, K, A, K, Wt, K, C, N, hip.BIAS, bias=bias)),
         "NT gelu+aux": t(lambda: hip.gemm(0, M, N, K, A, K, Wt, K, C, N, hip.BIAS | hip.GELU, bias=bias, aux_out=aux, ldaux=N)),
         "NN plain": t(lambda: hip.gemm(1, M, N, K, A, K, Wn, N, C, N, 0)),
         "NT gelu+daux": t(lambda: hip.gemm(0, M, N, K, A, K, Wt, K, C, N, hip.BIAS | hip.GELU | hip.AUX_DGELU, bias=bias, aux_out=aux, ldaux=N)),
         "NN dgelu": t(lambda: hip.gemm(1, M, N, K, A, K, Wn, N, C, N, hip.DGELU, aux_in=aux, ldaux=N)),
         "NN mulaux": t(lambda: hip.gemm(1, M, N, K, A, K, Wn, N, C, N, hip.MULAUX, aux_in=aux, ldaux=N))}
    print(f"({M},{N},{K}) " + " | ".join(f"{k} {v*1e3:6.1f}us {fl/v/1e9:5.0f}TF" for k, v in r.items()), flush=True)
