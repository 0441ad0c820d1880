"""GPU time of forward / backward / optimizer per step from the first step on (events on the main stream): which phase is slow in
the first replays after the capture?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import maestro_amd.engine as E
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
E.RING = int(os.environ.get("RING", E.RING))
SYNC_AT = int(os.environ.get("SYNC_AT", 4))       # -1: never
STEPS = int(os.environ.get("STEPS", 16))
dev = torch.device("cuda:0")
torch.set_num_threads(4)
torch.manual_seed(42)
ds, model = bench.build_model("c3")
loop = PretrainLoop(model, 32, dev, total_steps=100)
batch = synthetic_batch(ds.dataset, 32, dev)
eng = loop.engine
if os.environ.get("EAGER"):
    eng.use_graphs = False
rows = []
for it in range(STEPS):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    loss = eng.forward(batch)
    eng.zero_grad()
    ev[1].record()
    eng.backward()
    ev[2].record()
    loop._optimizer_step(1.0)
    loop.it += 1
    ev[3].record()
    rows.append(ev)
    if it == SYNC_AT:
        torch.cuda.synchronize()      # (what the bench does after its warm-up)
torch.cuda.synchronize()
for it, ev in enumerate(rows):
    if os.environ.get("BRIEF"):
        continue
    print(f"step {it:2d}: forward {ev[0].elapsed_time(ev[1]):7.2f}  backward {ev[1].elapsed_time(ev[2]):7.2f}  adamw {ev[2].elapsed_time(ev[3]):6.2f}  total {ev[0].elapsed_time(ev[3]):7.2f} ms")
print("RING", E.RING, "SYNC_AT", SYNC_AT, "forward ms:", " ".join(f"{ev[0].elapsed_time(ev[1]):.2f}" for ev in rows))
