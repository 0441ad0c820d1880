"""Proxy experiment for sub-batch streams: ONE engine at B = 32 against TWO independent engines at B = 16 stepping concurrently on
two HIP streams of the same GPU (same total tiles per pair of steps).  If the pair is not clearly faster, splitting the batch of a
group over parallel streams inside the engine (to overlap one half's VALU-bound attention / LayerNorm launches and kernel tails with
the other half's GEMMs) is not worth building."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch

dev = torch.device("cuda:0")
torch.set_num_threads(4)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(loops, batches, streams, tiles):
    def pair():
        for lp, b, st in zip(loops, batches, streams):
            with torch.cuda.stream(st):
                lp.step(b)
    for _ in range(6):
        pair()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        pair()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return tiles * STEPS / dt, 1e3 * dt / STEPS


res = {}
for name, nb in (("one engine, B=32", (32,)), ("two engines, B=16 each", (16, 16)), ("one engine, B=16", (16,))):
    loops, batches, streams = [], [], []
    for i, b in enumerate(nb):
        torch.manual_seed(42)
        ds, model = bench.build_model("c3")
        loops.append(PretrainLoop(model, b, dev, total_steps=200))
        batches.append(synthetic_batch(ds.dataset, b, dev, seed=i))
        streams.append(torch.cuda.Stream())
    torch.manual_seed(43)
    v, ms = run(loops, batches, streams, sum(nb))
    res[name] = (v, ms)
    print(f"{name:26s} {v:8.1f} tiles/s  {ms:7.2f} ms per (pair of) step(s)", flush=True)
    del loops, batches, streams
    import gc
    gc.collect()
    torch.cuda.empty_cache()
