"""Split-K with fp32 atomics for the small-M, long-K GEMMs of C4 (M = 576 ... 1792 rows, K = 4096) against the fused-epilogue launch:
measured 40.0 -> 29.8 us at M = 576 (init copy included), flat at M >= 1152, slower at K = 1024 and on the C3 shapes -- not adopted."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((576, 1024, 4096), (1152, 1024, 4096), (1792, 1024, 4096), (576, 1024, 1024), (1792, 1024, 1024), (3200, 768, 3072), (8192, 768, 3072)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); Wn = torch.randn(K, N, device=dev).bfloat16()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev)
    C = torch.empty(M, N, device=dev); C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = timeit(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, bias=bias, res=res, ldr=N))
    def sk():
        C.copy_(res)    # stand-in for the (res + bias) initialisation pass
        hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.OUT_F32 | hip.ATOMIC)
    t1 = timeit(sk)
    t2 = timeit(lambda: hip.gemm(1, M, N, K, A, K, Wn, N, C16, N, 0))
    def sk2():
        C.zero_()
        hip.gemm(1, M, N, K, A, K, Wn, N, C, N, hip.OUT_F32 | hip.ATOMIC)
    t3 = timeit(sk2)
    fl = 2.0 * M * N * K
    print(f"({M},{N},{K}) NT fused epilogue {t0:6.1f} us {fl/t0/1e6:5.0f} TF | init + split-K atomics {t1:6.1f} us {fl/t1/1e6:5.0f} TF || NN bf16 {t2:6.1f} us | zero + split-K f32 {t3:6.1f} us", flush=True)
