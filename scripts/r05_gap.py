"""Round 5 gap table, the isolated columns: the C3 step's heaviest GEMM signatures through the library's own dispatch (MH_TILE_AUTO),
timed four ways inside one process (interleaved rounds, min over rounds, HIP events around runs of launches):
  plain_hot   no epilogue (bf16 out), ONE buffer set relaunched back to back (outputs / operands stay in the 256 MiB Infinity Cache)
  epi_hot     the epilogue the step runs the shape with, one buffer set
  plain_cold  no epilogue, ROTATING buffer sets (R sets, footprint >= 1.5 GiB): every launch finds its operands and outputs cold,
  epi_cold    as inside the step, where every layer has its own activation buffers
Weights rotate too (every layer has its own).  Output: one line per shape, us.  The in-step columns come from bench.py --shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
FC1 = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
DFC2 = hip.MULAUX | hip.AUX_U8 | hip.COLSUM
F32 = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL
SHAPES = [("fc1", 0, 8192, 3072, 768, FC1), ("dfc2", 1, 8192, 3072, 768, DFC2), ("fc2", 0, 8192, 768, 3072, F32),
          ("dfc1", 1, 8192, 768, 3072, 0), ("qkv", 0, 8192, 2304, 768, 0), ("dqkv", 1, 8192, 768, 2304, 0),
          ("proj", 0, 8192, 768, 768, F32), ("dproj", 1, 8192, 768, 768, 0),
          ("dec fc1", 0, 32768, 3072, 512, FC1), ("dec dfc2", 1, 32768, 3072, 512, DFC2), ("dec fc2", 0, 32768, 512, 3072, F32),
          ("dec dfc1", 1, 32768, 512, 3072, 0), ("s2 fc2", 0, 3200, 768, 3072, F32), ("s2 fc1", 0, 3200, 3072, 768, FC1),
          ("jnt fc1", 0, 11392, 3072, 768, FC1), ("jnt fc2", 0, 11392, 768, 3072, F32)]
only = sys.argv[1:]


def make(lay, M, N, K, fl):  # noqa: N803
    A = torch.randn(M, K, device=dev).bfloat16()  # noqa: N806
    B = ((torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)) / K ** 0.5).bfloat16()  # noqa: N806
    out = torch.empty(M, N, dtype=torch.float32 if fl & hip.OUT_F32 else torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if fl & hip.RESIDUAL else None
    aux = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev) if fl & (hip.AUX_DGELU | hip.MULAUX) else None
    cs = torch.empty((M + 63) // 64, N, device=dev) if fl & hip.COLSUM else None
    kw = dict(bias=bias if fl & hip.BIAS else None, res=res, ldr=N if res is not None else 0, ldaux=N if aux is not None else 0,
              aux_out=aux if fl & hip.AUX_DGELU else None, aux_in=aux if fl & hip.MULAUX else None, colsum=cs)
    nbytes = sum(t.numel() * t.element_size() for t in (A, B, out, res, aux) if t is not None)
    return (lambda: hip.gemm(lay, M, N, K, A, K, B, B.shape[1], out, N, fl, **kw)), nbytes


def timeit(fns, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':9s} {'lay':3s} {'(M, N, K)':20s} | plain_hot  epi_hot | plain_cold epi_cold | us; TF/s of epi_cold")
for name, lay, M, N, K, fl in SHAPES:
    if only and not any(o in name for o in only):
        continue
    f0, b0 = make(lay, M, N, K, 0)
    r = max(2, int(1.5 * 2 ** 30 / b0) + 1)
    plain = [f0] + [make(lay, M, N, K, 0)[0] for _ in range(r - 1)]
    f1, b1 = make(lay, M, N, K, fl)
    r1 = max(2, int(1.5 * 2 ** 30 / b1) + 1)
    epi = [f1] + [make(lay, M, N, K, fl)[0] for _ in range(r1 - 1)]
    for f in plain + epi:
        f()
    res = {k: [] for k in ("ph", "eh", "pc", "ec")}
    for _ in range(5):
        res["ph"].append(timeit(plain[:1], 16))
        res["eh"].append(timeit(epi[:1], 16))
        res["pc"].append(timeit(plain, 2 * len(plain)))
        res["ec"].append(timeit(epi, 2 * len(epi)))
    m = {k: min(v) for k, v in res.items()}
    print(f"{name:9s} {'NT' if lay == 0 else 'NN':3s} ({M:5d},{N:5d},{K:5d})  | {m['ph']:8.1f} {m['eh']:8.1f} | {m['pc']:9.1f} {m['ec']:8.1f} | "
          f"{2.0 * M * N * K / m['ec'] / 1e6:6.0f}", flush=True)
    del plain, epi
    torch.cuda.empty_cache()
