"""Round 6: the epilogue-heavy GEMMs of the step (fc1: bias + GELU + byte-coded GELU'; fc2 dgrad: GELU' multiply + column sums) on every
tile family, isolated and hot: where does an exposed epilogue cost the most, and does two-workgroups-per-CU (whose epilogues can
overlap the neighbour's main loop) already beat one 256 x 256 workgroup?  us per launch (TFLOP/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
names = {0: "reg128", 7: "pp128", 1: "d256", 2: "d256x128", 3: "d128x256", 5: "d128x4"}
FC1 = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
DFC2 = hip.MULAUX | hip.AUX_U8 | hip.COLSUM
for (lay, M, N, K, kind) in ((0, 32768, 3072, 512, "gelu"), (1, 32768, 3072, 512, "dfc2"), (0, 8192, 3072, 768, "gelu"), (1, 8192, 3072, 768, "dfc2"),
                             (0, 12800, 3072, 512, "gelu"), (0, 11392, 3072, 768, "gelu"), (0, 3200, 3072, 768, "gelu"), (0, 32768, 3072, 512, "plain")):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)).bfloat16()
    bias = torch.randn(N, device=dev); aux8 = torch.randint(0, 255, (M, N), device=dev, dtype=torch.uint8)
    C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); cs = torch.empty((M + 63) // 64, N, device=dev)
    out = []
    for tile in names:
        if kind == "plain": f = lambda: hip.gemm(lay, M, N, K, A, K, W, W.shape[1], C16, N, 0, tile=tile)
        elif kind == "gelu": f = lambda: hip.gemm(lay, M, N, K, A, K, W, W.shape[1], C16, N, FC1, bias=bias, aux_out=aux8, ldaux=N, tile=tile)
        else: f = lambda: hip.gemm(lay, M, N, K, A, K, W, W.shape[1], C16, N, DFC2, aux_in=aux8, ldaux=N, colsum=cs, tile=tile)
        try:
            ms = t(f)
            out.append(f"{names[tile]} {ms*1e3:6.1f} ({2.0*M*N*K/ms/1e9:4.0f})")
        except Exception as e:  # noqa: BLE001
            out.append(f"{names[tile]}   n/a")
    print(f"{'NT' if lay == 0 else 'NN'} ({M},{N},{K}) {kind:5s} " + " | ".join(out), flush=True)
