"""All wgrads of the C3 encoder-side backward segment as ONE grouped DMA launch vs the per-GEMM split-K launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def layer(Mtok, dim, mlp, inner):
    return [(dim, mlp, Mtok), (mlp, dim, Mtok), (dim, inner, Mtok), (3 * inner, dim, Mtok)]
sets = {"enc (2 groups x 9 layers)": [s for _ in range(9) for s in layer(8192, 768, 3072, 768) + layer(3200, 768, 3072, 768)],
        "joint (3 layers)": [s for _ in range(3) for s in layer(11392, 768, 3072, 768)],
        "dec (2 groups x 3 layers)": [s for _ in range(3) for s in layer(32768, 512, 3072, 512) + layer(12800, 512, 3072, 512)]}
for name, shapes in sets.items():
    probs = []
    for (M, N, K) in shapes:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        probs.append((A, B, torch.zeros(M, N, device=dev), M, N, K, M, N, N))
    g = hip.GroupedTN(probs, dev)
    def run_grouped(): g.launch()
    def run_split():
        for (A, B, C, M, N, K, lda, ldb, ldc) in probs:
            hip.gemm(2, M, N, K, A, lda, B, ldb, C, ldc, hip.OUT_F32 | hip.ATOMIC)
    os.environ["MH_GEMM_DMA"] = "0"
    res = []
    for f in (run_split, run_grouped):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print(f"{name:28s} {len(probs):3d} problems {g.tiles:5d} tiles {g.flops/1e12:6.2f} TFLOP: split-K v1 {res[0]:6.3f} ms ({g.flops/res[0]/1e9:6.1f} TF) | grouped DMA {res[1]:6.3f} ms ({g.flops/res[1]/1e9:6.1f} TF)", flush=True)
