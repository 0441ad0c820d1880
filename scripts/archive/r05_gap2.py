"""Which operand's coldness costs the K = 768 GEMMs their 10-16 us?  fc1 / qkv shapes, library dispatch, rotating ONLY one of
{A, W, outputs} (the others stay hot), plain and with the step's epilogue.  us, min of 5 interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
FC1 = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
R = 24
def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, lay, M, N, K, fl in [("fc1 plain", 0, 8192, 3072, 768, 0), ("fc1 gelu", 0, 8192, 3072, 768, FC1), ("qkv", 0, 8192, 2304, 768, 0),
                               ("fc2 f32res", 0, 8192, 768, 3072, F32), ("dfc1 NN", 1, 8192, 768, 3072, 0)]:
    As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]
    Bs = [((torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)) / K ** 0.5).bfloat16() for _ in range(R)]
    outs = [torch.empty(M, N, dtype=torch.float32 if fl & hip.OUT_F32 else torch.bfloat16, device=dev) for _ in range(R)]
    auxs = [torch.empty(M, N, dtype=torch.uint8, device=dev) for _ in range(R)] if fl & hip.AUX_DGELU else [None] * R
    ress = [torch.randn(M, N, device=dev) for _ in range(R)] if fl & hip.RESIDUAL else [None] * R
    bias = torch.randn(N, device=dev)
    def mk(ia, ib, io):
        A, B, out, aux, res = As[ia], Bs[ib], outs[io], auxs[io], ress[io]
        kw = dict(bias=bias if fl & hip.BIAS else None, res=res, ldr=N if res is not None else 0, ldaux=N if aux is not None else 0, aux_out=aux)
        return lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, **kw)
    sets = {"hot": [mk(0, 0, 0)], "A cold": [mk(i, 0, 0) for i in range(R)], "W cold": [mk(0, i, 0) for i in range(R)],
            "out cold": [mk(0, 0, i) for i in range(R)], "all cold": [mk(i, i, i) for i in range(R)]}
    for v in sets.values():
        for f in v[:2]: f()
    res_t = {k: [] for k in sets}
    for _ in range(5):
        for k, v in sets.items(): res_t[k].append(timeit(v, 2 * R))
    print(f"{name:10s} ({M},{N},{K}) " + " | ".join(f"{k} {min(v):6.1f}" for k, v in res_t.items()), flush=True)
    del As, Bs, outs, auxs, ress, sets
    torch.cuda.empty_cache()
