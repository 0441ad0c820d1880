"""A/B inside one process: the persistent 128x128 tile with the epilogue of tile t inside the main loop of tile t + 1
(MH_TILE_PP_128, gemm_pp.hip) against the one-tile-per-workgroup register-staged kernel (MH_TILE_REG_128) and the library's own
choice (MH_TILE_AUTO: the 256x256 LDS-DMA tile on the M = 32768 decoder shapes), on the C3 step's GEMMs with their REAL epilogues.
Interleaved rounds, min and median per variant (guide rule 24)."""
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
FC1 = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
DFC2 = hip.MULAUX | hip.AUX_U8 | hip.COLSUM
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
cases = []
SUP = os.environ.get("PP_SHAPES", "") == "sup"      # the probe / finetune step: UNMASKED encoder sequences (B = 32: 32768 + 12800, joint 45568 rows)
for M in ((32768, 12800, 45568) if SUP else (8192, 11392, 3200, 12800, 32768)):
    dim, mlp, inner = (768, 3072, 768) if (SUP or M in (8192, 11392, 3200)) else (512, 3072, 512)
    cases += [("qkv", 0, M, 3 * inner, dim, 0), ("fc1", 0, M, mlp, dim, FC1), ("out", 0, M, dim, inner, F32), ("fc2", 0, M, dim, mlp, F32),
              ("dfc2", 1, M, mlp, dim, DFC2), ("dfc1", 1, M, dim, mlp, 0), ("dqkv", 1, M, dim, 3 * inner, 0)]
only = sys.argv[1:]
tot = {}
for name, lay, M, N, K, fl in cases:
    if only and name not in only: continue
    A = torch.randn(M, K).bfloat16().to(dev)
    B = ((torch.randn(N, K) if lay == 0 else torch.randn(K, N)) / K ** 0.5).bfloat16().to(dev)
    out = torch.empty(M, N, dtype=torch.float32 if fl & hip.OUT_F32 else torch.bfloat16, device=dev)
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev) if fl & hip.RESIDUAL else None
    aux = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev) if fl & (hip.AUX_DGELU | hip.MULAUX) else None
    cs = torch.empty((M + 63) // 64, N, device=dev) if fl & hip.COLSUM else None
    kw = dict(bias=bias if fl & hip.BIAS else None, res=res, ldr=N if res is not None else 0, ldaux=N if aux is not None else 0,
              aux_out=aux if fl & hip.AUX_DGELU else None, aux_in=aux if fl & hip.MULAUX else None, colsum=cs)
    variants = {"reg128": hip.TILE_REG_128, "pp128": hip.TILE_PP_128, "auto": hip.TILE_AUTO, "reg64": hip.TILE_REG_64, "reg192": hip.TILE_REG_192}
    if SUP:
        variants = {"reg128": hip.TILE_REG_128, "pp128": hip.TILE_PP_128, "d256": hip.TILE_DMA_256, "auto": hip.TILE_AUTO, "reg192": hip.TILE_REG_192}
    res_t = {k: [] for k in variants}
    ok = {}
    for k, t in variants.items():
        try:
            hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, tile=t, **kw); ok[k] = True
        except hip.HipExtensionError:
            ok[k] = False
    for _ in range(5):
        for k, t in variants.items():
            if ok[k]:
                res_t[k].append(timeit(lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, tile=t, **kw)))
    flop = 2.0 * M * N * K
    line = f"{name:5s} {'NT' if lay == 0 else 'NN'} ({M:5d},{N:4d},{K:4d}) tiles128={-(-M // 128) * (N // 128):5d}"
    for k in variants:
        if ok[k]:
            mn, md = min(res_t[k]), statistics.median(res_t[k])
            line += f" | {k} {mn:6.1f} {flop / mn / 1e6:5.0f}TF"
            tot.setdefault(k, 0.0); tot[k] += mn * (9 if M in (8192, 3200) else 3)
        else:
            line += f" | {k}      n/a"
    print(line, flush=True)
print("per-step sums (us, weighted by launches per C3 step):", {k: round(v) for k, v in tot.items()})
