"""Does the isolated advantage of the ping-pong tile survive SUSTAINED load?  Each variant runs back to back for ~2 s (captured in a
hipGraph of 200 launches, replayed), alternating, and reports TFLOP/s over the whole interval -- short bursts (scripts/bench_pp.py)
run on a cool chip at a high clock; the training step is a sustained ~1.3 kW load."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maestro_amd import hip
dev = torch.device("cuda:0")
FC1 = hip.BIAS | hip.GELU | hip.AUX_DGELU | hip.AUX_U8
def graph_of(fn, n=200):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    return g
for name, M, N, K, fl in [("fc1", 8192, 3072, 768, FC1), ("qkv", 8192, 2304, 768, 0), ("fc1 dec", 32768, 3072, 512, FC1)]:
    A = torch.randn(M, K).bfloat16().to(dev); W = (torch.randn(N, K) / K ** 0.5).bfloat16().to(dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.empty(M, N, dtype=torch.uint8, device=dev); bias = torch.randn(N, device=dev)
    kw = dict(bias=bias, aux_out=aux, ldaux=N) if fl else {}
    graphs = {k: graph_of(lambda t=t: hip.gemm(0, M, N, K, A, K, W, K, C, N, fl, tile=t, **kw)) for k, t in
              (("reg128", hip.TILE_REG_128), ("pp128", hip.TILE_PP_128), ("dma256", hip.TILE_DMA_256))}
    out = []
    for rnd in range(2):
        for k, g in graphs.items():
            reps = max(1, int(2.0 / (200 * 60e-6)))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): g.replay()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out.append(f"{k} {dt / (reps * 200) * 1e6:6.1f} us {2.0 * M * N * K * reps * 200 / dt / 1e12:5.0f} TF")
    print(f"{name} ({M},{N},{K}): " + " | ".join(out), flush=True)
