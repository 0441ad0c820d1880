"""Rotated tile raster (MH_GEMM_ROT) against the round-4 raster, hot and cold operands, the step's heaviest shapes.  us."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
FC1 = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
DFC2 = hip.MULAUX | hip.AUX_U8 | hip.COLSUM
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SH = [("fc1", 0, 8192, 3072, 768, FC1), ("qkv", 0, 8192, 2304, 768, 0), ("dfc2", 1, 8192, 3072, 768, DFC2), ("fc2", 0, 8192, 768, 3072, F32),
      ("dfc1", 1, 8192, 768, 3072, 0), ("dqkv", 1, 8192, 768, 2304, 0), ("proj", 0, 8192, 768, 768, F32),
      ("dec fc1", 0, 32768, 3072, 512, FC1), ("dec dfc2", 1, 32768, 3072, 512, DFC2), ("dec fc2", 0, 32768, 512, 3072, F32), ("dec qkv", 0, 32768, 1536, 512, 0),
      ("s2 fc1", 0, 3200, 3072, 768, FC1), ("s2 fc2", 0, 3200, 768, 3072, F32), ("jnt fc1", 0, 11392, 3072, 768, FC1), ("jnt dfc1", 1, 11392, 768, 3072, 0)]
for name, lay, M, N, K, fl in SH:
    def make():
        A = torch.randn(M, K, device=dev).bfloat16()
        B = ((torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)) / K ** 0.5).bfloat16()
        out = torch.empty(M, N, dtype=torch.float32 if fl & hip.OUT_F32 else torch.bfloat16, device=dev)
        bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev) if fl & hip.RESIDUAL else None
        aux = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev) if fl & (hip.AUX_DGELU | hip.MULAUX) else None
        cs = torch.empty((M + 63) // 64, N, device=dev) if fl & hip.COLSUM else None
        kw = dict(bias=bias if fl & hip.BIAS else None, res=res, ldr=N if res is not None else 0, ldaux=N if aux is not None else 0,
                  aux_out=aux if fl & hip.AUX_DGELU else None, aux_in=aux if fl & hip.MULAUX else None, colsum=cs)
        nb = sum(t.numel() * t.element_size() for t in (A, B, out, res, aux) if t is not None)
        return (lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, **kw)), nb
    f0, nb = make()
    R = max(2, int(1.5 * 2 ** 30 / nb) + 1)
    cold = [f0] + [make()[0] for _ in range(R - 1)]
    res = {k: [] for k in ("hot0", "hot1", "cold0", "cold1")}
    for _ in range(5):
        for rot in (0, 1):
            hip.lib().mh_debug_set_gemm_rot(rot)
            res[f"hot{rot}"].append(timeit(cold[:1], 16))
            res[f"cold{rot}"].append(timeit(cold, 2 * R))
    m = {k: min(v) for k, v in res.items()}
    print(f"{name:9s} {'NT' if lay == 0 else 'NN'} ({M:5d},{N:4d},{K:4d}) hot {m['hot0']:6.1f} -> {m['hot1']:6.1f} | cold {m['cold0']:6.1f} -> {m['cold1']:6.1f}", flush=True)
    del cold; torch.cuda.empty_cache()
hip.lib().mh_debug_set_gemm_rot(0)
