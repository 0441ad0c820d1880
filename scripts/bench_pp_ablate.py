"""Where does the fc1 GEMM's time go?  Diagnostic builds of the persistent ping-pong tile (MH_TILE_PP_128_DIAG*), NT, on the
C3 step's fc1 / qkv shapes, interleaved rounds in one process:
  main    main loop only (no epilogue at all)
  valu    + the epilogue's arithmetic interleaved in the next tile's main loop, results discarded (no global stores)
  stores  + the epilogue's stores (bf16 + byte code), arithmetic replaced by moves
  full    the kernel (the epilogue's VALU instructions placed by the compiler's scheduler)
  valu pinned / valu behind reads   the full kernel with the epilogue's VALU instructions pinned behind every MFMA pair / partly
          behind the fragment reads (sched_group_barrier)
  reg128  the one-tile-per-workgroup kernel (epilogue exposed, LDS-staged) with the same epilogue, and with a plain bf16 one"""
# NEEDS a library built with MH_BUILD_FLAGS=-DMH_DIAG_TILES (the shipped one declines the DIAG tile ids)
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
for M, N, K in [(8192, 3072, 768), (11392, 3072, 768), (12800, 3072, 512), (32768, 3072, 512)]:
    A = torch.randn(M, K).bfloat16().to(dev); W = (torch.randn(N, K) / K ** 0.5).bfloat16().to(dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.empty(M, N, dtype=torch.uint8, device=dev)
    bias = torch.randn(N, device=dev)
    def run(tile, fl):
        kw = dict(bias=bias, aux_out=aux, ldaux=N) if fl else {}
        return lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, fl, tile=tile, **kw)
    v = {"main": run(8, FC1), "valu": run(9, FC1), "stores": run(10, FC1), "full": run(hip.TILE_PP_128, FC1), "valu pinned": run(11, FC1), "valu behind reads": run(12, FC1),
         "pp plain": run(hip.TILE_PP_128, 0), "pp plain main": run(8, 0), "reg128 gelu": run(hip.TILE_REG_128, FC1), "reg128 plain": run(hip.TILE_REG_128, 0)}
    res = {k: [] for k in v}
    for f in v.values(): f()
    for _ in range(5):
        for k, f in v.items(): res[k].append(timeit(f))
    print(f"({M},{N},{K}): " + " | ".join(f"{k} {min(r):6.1f}" for k, r in res.items()) + "  (us, min of 5 interleaved rounds)", flush=True)
