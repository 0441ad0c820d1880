"""Round-4 check of the MFMA-shape hypothesis on the real fc1 GEMM (profiles/r03_isa_budget.txt): the same 128 x 128 NT tile on
v_mfma_f32_16x16x32_bf16 (MH_TILE_REG_128), as the persistent ping-pong kernel (MH_TILE_PP_128) and on v_mfma_f32_32x32x16_bf16
(MH_TILE_M32_128, csrc/gemm_m32.hip -- EXPERIMENTAL: run tests/test_gemm_m32_gpu.py with MAESTRO_TEST_EXPERIMENTAL=1 first), each
with a plain bias epilogue and with bias + GELU + byte-coded GELU'.  Device time per launch from 20 launches per hipGraph.
Prediction: the GELU epilogue costs the 16x16x32 tile ~+30 % over its plain form and the 32x32x16 tile ~+8 %."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
TILES = (("16x16x32", hip.TILE_REG_128), ("ping-pong", hip.TILE_PP_128), ("32x32x16", hip.TILE_M32_128))
for (M, N, K) in ((8192, 3072, 768), (3200, 3072, 768), (11392, 3072, 768), (32768, 2048, 512), (12800, 2048, 512)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.empty(M, N, device=dev, dtype=torch.uint8)
    fl, out = 2.0 * M * N * K, []
    for name, tile in TILES:
        try:
            plain = t(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.BIAS if tile != hip.TILE_PP_128 else 0, bias=bias, tile=tile))
            gelu = t(lambda: hip.gemm(0, M, N, K, A, K, W, K, C, N, hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8, bias=bias, aux_out=aux,
                                      ldaux=N, tile=tile))
            out.append(f"{name}: plain {plain:6.1f} us {fl / plain / 1e6:4.0f} TF, GELU {gelu:6.1f} us {fl / gelu / 1e6:4.0f} TF (+{100 * (gelu / plain - 1):4.1f} %)")
        except hip.HipExtensionError as exc:
            out.append(f"{name}: {exc}")
    print(f"({M},{N},{K})  " + " | ".join(out), flush=True)
