"""Would two half-batch chains running side by side beat one full-batch chain?  Two independent engines (B = 16 each) on two
streams vs one engine at B = 32 (upper bound for an intra-batch stream split: the weight gradients would still be shared)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
torch.set_num_threads(4)
dev = torch.device("cuda:0")
def make(B, seed):
    torch.manual_seed(seed)
    ds, model = bench.build_model("c3")
    return PretrainLoop(model, B, dev, total_steps=200), synthetic_batch(ds.dataset, B, dev, seed=seed)
def run(loops, streams, steps):
    for _ in range(steps):
        for (loop, batch), st in zip(loops, streams):
            with torch.cuda.stream(st):
                loop.step(batch)
one = [make(32, 0)]
s0 = torch.cuda.current_stream()
run(one, [s0], 6); torch.cuda.synchronize(); t0 = time.perf_counter(); run(one, [s0], 30); torch.cuda.synchronize()
print(f"one engine  B=32          : {32 * 30 / (time.perf_counter() - t0):7.1f} tiles/s", flush=True)
del one
two = [make(16, 1), make(16, 2)]
sts = [torch.cuda.Stream(), torch.cuda.Stream()]
run(two, sts, 6); torch.cuda.synchronize(); t0 = time.perf_counter(); run(two, sts, 30); torch.cuda.synchronize()
print(f"two engines B=16 + B=16   : {32 * 30 / (time.perf_counter() - t0):7.1f} tiles/s (concurrent streams)", flush=True)
run(two[:1], sts[:1], 6); torch.cuda.synchronize(); t0 = time.perf_counter(); run(two[:1], sts[:1], 30); torch.cuda.synchronize()
print(f"one engine  B=16          : {16 * 30 / (time.perf_counter() - t0):7.1f} tiles/s", flush=True)
