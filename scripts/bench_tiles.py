"""Every GEMM signature of the C3 step (B=32) x every kernel tile: isolated TFLOP/s, to see what tuning can buy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
rows = {"enc_aerial": (8192, 768, 3072, 768), "enc_s2": (3200, 768, 3072, 768), "joint": (11392, 768, 3072, 768),
        "dec_aerial": (32768, 512, 3072, 512), "dec_s2": (12800, 512, 3072, 512)}
names = {0: "reg128", 1: "d256", 2: "d256x128", 3: "d128x256", 4: "d128", 5: "d128q"}
tot = {t: 0.0 for t in names}; best_tot = 0.0
for tag, (M, dim, mlp, inner) in rows.items():
    for lay, lname in ((0, "NT"), (1, "NN")):
        for (N, K) in ((3 * inner, dim), (dim, inner), (mlp, dim), (dim, mlp)):
            if lay == 1:
                N, K = K, N
            A = torch.randn(M, K, device=dev).bfloat16()
            B = (torch.randn(N, K, device=dev) if lay == 0 else torch.randn(K, N, device=dev)).bfloat16()
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            res = {}
            for t in names:
                try:
                    for _ in range(2): hip.gemm(lay, M, N, K, A, K, B, B.shape[1], C, N, 0, tile=t)
                except hip.HipExtensionError:
                    continue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): hip.gemm(lay, M, N, K, A, K, B, B.shape[1], C, N, 0, tile=t)
                e1.record(); torch.cuda.synchronize()
                res[t] = e0.elapsed_time(e1) / 10
                tot[t] += res[t]
            best_tot += min(res.values())
            fl = 2.0 * M * N * K
            print(f"{tag:11s} {lname} ({M:5d},{N:4d},{K:4d}) " + " ".join(f"{names[t]} {res[t]*1e3:6.0f}us {fl/res[t]/1e9:5.0f}TF |" for t in res), flush=True)
print("sum of per-layer GEMM times (ms): " + ", ".join(f"{names[t]} {tot[t]:.3f}" for t in tot) + f" ; best-per-shape {best_tot:.3f}")
