"""Does the HBM-bound AdamW pass hide under the MFMA-bound grouped weight-gradient launch?  (round 6, experiment 15)
C3 at B = 32 on one GPU: after three steps (so that every buffer holds real data) time, with HIP events,
  (a) the step's grouped TN launch alone, (b) mh_adamw over the whole flat buffer alone, (c) both at once on two streams,
  (d) the grouped launch with HALF of the AdamW range beside it -- the form a pipelined optimizer tail would have.
usage: python scripts/ovl_wgrad_adamw.py [c3|c4|...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from maestro_amd import hip
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(42)
ds, model = bench.build_model(cfg, "pretrain")
B = 32
loop = PretrainLoop(model, B, dev, loss="l2_norm", total_steps=100, world_size=1)
batch = synthetic_batch(ds.dataset, B, dev, seed=0)
for _ in range(3):
    loop.step(batch)
torch.cuda.synchronize()
eng, opt = loop.engine, loop.opt
tables = list(eng._wgrad_tables.values())
assert len(tables) == 1, [k for k in eng._wgrad_tables]
grouped = tables[0][0]
ps = eng.store
n = ps.total
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def adamw(lo, hi):
    hip.adamw(ps.flat[lo:hi], ps.grad[lo:hi], opt.m[lo:hi], opt.v[lo:hi], ps.half[lo:hi], hi - lo, 1e-6, opt.betas[0], opt.betas[1],
              opt.eps, opt.wd, 10, 1.0)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def both(lo, hi):
    def run():
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1):
            grouped.launch()
        with torch.cuda.stream(s2):
            adamw(lo, hi)
        main.wait_stream(s1); main.wait_stream(s2)
    return run


print(f"{cfg}: {n / 1e6:.1f} M parameters")
ta = timed(grouped.launch)
tb = timed(lambda: adamw(0, n))
th = timed(lambda: adamw(0, n // 2))
tc = timed(both(0, n))
td = timed(both(0, n // 2))
print(f"(a) grouped wgrad alone          {ta:7.3f} ms")
print(f"(b) AdamW alone, whole buffer    {tb:7.3f} ms   half the buffer {th:7.3f} ms")
print(f"(c) both on two streams          {tc:7.3f} ms   sum {ta + tb:7.3f}  -> hidden {ta + tb - tc:6.3f} ms")
print(f"(d) grouped + half the AdamW     {td:7.3f} ms   sum {ta + th:7.3f}  -> hidden {ta + th - td:6.3f} ms")
