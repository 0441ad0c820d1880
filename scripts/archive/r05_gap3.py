"""fc1-shaped GEMM (8192 x 3072 x 768, NT, plain bf16 output, library dispatch): how does the launch time depend on HOW MANY distinct
copies of one operand rotate between launches (everything else fixed)?  n = 1 is the hot case.  us, min of 5 interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
M, N, K = 8192, 3072, 768
RMAX = 96
def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(RMAX)]
Bs = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(RMAX)]
outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(32)]
def mk(ia, ib, io):
    A, B, out = As[ia], Bs[ib], outs[io]
    return lambda: hip.gemm(0, M, N, K, A, K, B, K, out, N, 0)
sets = {"hot": [mk(0, 0, 0)]}
for n in (2, 4, 8, 24, 96):
    sets[f"W x{n}"] = [mk(0, i, 0) for i in range(n)]
for n in (2, 8, 24, 96):
    sets[f"A x{n}"] = [mk(i, 0, 0) for i in range(n)]
for n in (2, 8, 32):
    sets[f"out x{n}"] = [mk(0, 0, i) for i in range(n)]
for n in (2, 8, 32):
    sets[f"all x{n}"] = [mk(i, i, i) for i in range(n)]
sets["A,W x32"] = [mk(i, i, 0) for i in range(32)]
sets["W,out x32"] = [mk(0, i, i) for i in range(32)]
sets["A,out x32"] = [mk(i, 0, i) for i in range(32)]
for v in sets.values():
    for f in v[:2]: f()
res_t = {k: [] for k in sets}
for _ in range(5):
    for k, v in sets.items(): res_t[k].append(timeit(v, max(32, 2 * len(v))))
for k, v in res_t.items():
    print(f"{k:12s} {min(v):6.1f} us", flush=True)
