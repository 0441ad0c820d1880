"""Per layer-op of the C3 step (B = 32): the groups' GEMMs as separate launches (library rule, one stream) vs ONE persistent
grouped launch (mh_gemm_grouped), with the op's real fused epilogue.  Isolated timings, random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)
sets = {"enc (aerial 8192 + s2 3200)": ((8192, 3200), 768, 3072, 768), "joint (11392)": ((11392,), 768, 3072, 768),
        "dec (aerial 32768 + s2 12800)": ((32768, 12800), 512, 3072, 512)}
ops = [("qkv", 0, lambda d, m, i: (3 * i, d), 0), ("proj+res", 0, lambda d, m, i: (d, i), hip.OUT_F32 | hip.BIAS | hip.RESIDUAL),
       ("fc1+gelu", 0, lambda d, m, i: (m, d), hip.BIAS | hip.GELU | hip.AUX_DGELU), ("fc2+res", 0, lambda d, m, i: (d, m), hip.OUT_F32 | hip.BIAS | hip.RESIDUAL),
       ("d fc2", 1, lambda d, m, i: (m, d), hip.MULAUX | hip.COLSUM), ("d fc1", 1, lambda d, m, i: (d, m), 0),
       ("d proj", 1, lambda d, m, i: (i, d), 0), ("d qkv", 1, lambda d, m, i: (d, 3 * i), 0)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = {"single": 0.0, "grouped": 0.0}
for sname, (Ms, dim, mlp, inner) in sets.items():
    for oname, layout, nk, flags in ops:
        N, K = nk(dim, mlp, inner)
        probs = []
        for M in Ms:
            A = rnd(M, K).bfloat16().to(dev)
            B = (rnd(N, K) if layout == 0 else rnd(K, N)).bfloat16().to(dev)
            pr = dict(A=A, B=B, C=torch.empty(M, N, dtype=torch.float32 if flags & hip.OUT_F32 else torch.bfloat16, device=dev),
                      M=M, N=N, K=K, lda=K, ldb=B.shape[1], ldc=N, flags=flags)
            if flags & hip.BIAS: pr["bias"] = rnd(N).to(dev)
            if flags & hip.RESIDUAL: pr["res"], pr["ldr"] = rnd(M, N).to(dev), N
            if flags & hip.MULAUX: pr["aux_in"], pr["ldaux"] = rnd(M, N).bfloat16().to(dev), N
            if flags & hip.AUX_DGELU: pr["aux_out"], pr["ldaux"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev), N
            if flags & hip.COLSUM: pr["colsum"] = torch.empty((M + 63) // 64, N, device=dev)
            probs.append(pr)
        def single():
            for pr in probs:
                hip.gemm(layout, pr["M"], pr["N"], pr["K"], pr["A"], pr["lda"], pr["B"], pr["ldb"], pr["C"], pr["ldc"], pr["flags"],
                         bias=pr.get("bias"), res=pr.get("res"), ldr=pr.get("ldr", 0), aux_in=pr.get("aux_in"), aux_out=pr.get("aux_out"),
                         ldaux=pr.get("ldaux", 0), colsum=pr.get("colsum"))
        fl = sum(2.0 * pr["M"] * N * K for pr in probs)
        t_s = timeit(single)
        res = {}
        for sp in (None, 1):
            gg = hip.GroupedGemm(layout, probs, dev, split=sp)
            res[sp] = (timeit(gg.launch), gg.makespan, gg.ideal, gg.n_items)
        t_g = min(r[0] for r in res.values())
        tot["single"] += t_s; tot["grouped"] += t_g
        print(f"{sname:30s} {oname:9s} N={N:4d} K={K:4d}: single {t_s*1e3:6.1f} us {fl/t_s/1e9:6.0f} TF | grouped auto {res[None][0]*1e3:6.1f} us "
              f"{fl/res[None][0]/1e9:6.0f} TF (items {res[None][3]}, makespan {res[None][1]:.2f} vs ideal {res[None][2]:.2f}) | unsplit {res[1][0]*1e3:6.1f} us", flush=True)
print("sum over one layer of each stack (ms): single %.3f grouped %.3f" % (tot["single"], tot["grouped"]))
