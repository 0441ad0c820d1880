"""GEMM micro-bench with the real epilogues; batch timing vs per-launch event timing (as bench.py does)."""
import sys, torch
sys.path.insert(0, ".")
from maestro_amd import hip
dev = torch.device("cuda:0")

def run_case(tag, layout, M, N, K, flags, out_dtype, **kw):
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
    C = torch.zeros(M, N, device=dev, dtype=out_dtype)
    extra = {}
    if flags & hip.BIAS: extra["bias"] = torch.randn(N, device=dev)
    if flags & hip.GELU: extra["aux_out"] = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); extra["ldaux"] = N
    if flags & hip.DGELU: extra["aux_in"] = torch.randn(M, N, device=dev).bfloat16(); extra["ldaux"] = N
    if flags & hip.RESIDUAL: extra["res"] = torch.randn(M, N, device=dev); extra["ldr"] = N
    f = lambda: hip.gemm(layout, M, N, K, A, A.shape[1], B, B.shape[1], C, N, flags, **extra)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    batch = e0.elapsed_time(e1) / 20
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    single = sum(a.elapsed_time(b) for a, b in evs) / 20
    fl = 2 * M * N * K
    print(f"{tag:28s} M={M:6d} N={N:5d} K={K:5d}: batch {batch*1e3:7.1f} us {fl/batch/1e9:6.1f} TF | per-launch events {single*1e3:7.1f} us {fl/single/1e9:6.1f} TF", flush=True)

F32 = torch.float32; BF = torch.bfloat16
run_case("NT plain bf16", 0, 8192, 3072, 768, 0, BF)
run_case("NT bias+gelu+aux", 0, 8192, 3072, 768, hip.BIAS | hip.GELU, BF)
run_case("NT bias+res f32", 0, 8192, 768, 3072, hip.OUT_F32 | hip.BIAS | hip.RESIDUAL, F32)
run_case("NT plain bf16 N=768", 0, 8192, 768, 3072, 0, BF)
run_case("NN plain", 1, 8192, 3072, 768, 0, BF)
run_case("NN dgelu", 1, 8192, 3072, 768, hip.DGELU, BF)
run_case("NT dec mlp1 plain", 0, 32768, 3072, 512, 0, BF)
run_case("NT dec mlp1 gelu", 0, 32768, 3072, 512, hip.BIAS | hip.GELU, BF)
run_case("NN dec dgelu", 1, 32768, 3072, 512, hip.DGELU, BF)
run_case("TN atomic", 2, 3072, 768, 8192, hip.OUT_F32 | hip.ATOMIC, F32)
run_case("NT s2 M=3200", 0, 3200, 3072, 768, hip.BIAS | hip.GELU, BF)
