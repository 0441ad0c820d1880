"""Small-M long-K GEMMs (C4's per-group ViT-L encoders, C3's Sentinel-2 chain): one 128 x 128 tile per workgroup walks the whole K alone
on its CU and the launch takes K steps x one exposed memory round trip, whatever M is.  Does split-K with fp32 atomics
(MH_GEMM_ATOMIC: grid.y = splits, the epilogue adds into a C that already holds residual + bias) shorten them?
python scripts/bench_splitk_small_m.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(576, 1024, 4096), (1152, 1024, 4096), (1792, 1024, 4096), (576, 1024, 1024), (1792, 1024, 1024), (3200, 768, 3072),
          (3200, 768, 768), (8192, 768, 3072), (2304, 512, 4096), (7200, 512, 4096)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.capture_begin()
        for _ in range(reps):
            fn()
        g.capture_end()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


for (M, N, K) in SHAPES:
    gen = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=gen).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=gen) * 0.02).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=gen).to(dev)
    res = torch.randn(M, N, generator=gen).to(dev)
    C1, C2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    full = hip.OUT_F32 | hip.BIAS | hip.RESIDUAL

    def classic():
        hip.gemm(hip.GEMM_NT, M, N, K, A, K, W, K, C1, N, full, bias=bias, res=res, ldr=N)

    def prefill():
        torch.add(res, bias, out=C2)

    def atomic():
        hip.gemm(hip.GEMM_NT, M, N, K, A, K, W, K, C2, N, hip.OUT_F32 | hip.ATOMIC)

    def both():
        prefill(); atomic()
    classic(); both(); torch.cuda.synchronize()
    err = (C1 - C2).abs().max().item() / C1.abs().max().item()
    tc, tp, ta, tb = timed(classic), timed(prefill), timed(atomic), timed(both)
    print(f"NT ({M:5d}, {N:4d}, {K:4d}): rule {tc:6.1f} us | split-K atomic {ta:6.1f} + prefill {tp:5.1f} = {tb:6.1f} us | rel diff {err:.1e}", flush=True)
