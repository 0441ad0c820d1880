"""Does a read-and-discard sweep (mh_touch) of an HBM-cold weight matrix right before the GEMM recover the cold-W penalty?
fc1-shaped GEMM, W rotating over 96 copies.  us."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, lay) in [(8192, 3072, 768, 0), (8192, 768, 3072, 0), (8192, 768, 3072, 1), (32768, 3072, 512, 0), (3200, 3072, 768, 0)]:
    R = 96
    A = torch.randn(M, K, device=dev).bfloat16()
    Bs = [((torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)) / K ** 0.5).bfloat16() for _ in range(R)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    def g(i): return lambda: hip.gemm(lay, M, N, K, A, K, Bs[i], Bs[i].shape[1], out, N, 0)
    def t(i): return lambda: hip.touch(Bs[i])
    def tg(i):
        def f():
            hip.touch(Bs[i]); hip.gemm(lay, M, N, K, A, K, Bs[i], Bs[i].shape[1], out, N, 0)
        return f
    sets = {"hot": [g(0)], "W cold": [g(i) for i in range(R)], "touch only": [t(i) for i in range(R)], "touch+gemm": [tg(i) for i in range(R)]}
    for v in sets.values(): v[0]()
    res = {k: [] for k in sets}
    for _ in range(5):
        for k, v in sets.items(): res[k].append(timeit(v, 2 * R))
    m = {k: min(v) for k, v in res.items()}
    print(f"({M},{N},{K}) {'NT' if lay == 0 else 'NN'}: hot {m['hot']:.1f} | W cold {m['W cold']:.1f} | touch {m['touch only']:.1f} | touch+gemm {m['touch+gemm']:.1f} -> gemm after touch {m['touch+gemm'] - m['touch only']:.1f}", flush=True)
    del Bs
    torch.cuda.empty_cache()
