"""GPU time between the END of step t (after AdamW + the patch-embed weight repack) and the FIRST forward kernel of step t + 1
(behind the mask uploads), with the host running ahead as in bench.py: HIP events on the main stream, no profiler.
   MAESTRO_UPLOAD_STREAM=0|1 python scripts/r05_boundary.py"""
import os, sys, torch
os.environ.setdefault("MAESTRO_WARM_PASSES", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
dev = torch.device("cuda:0")
torch.set_num_threads(4); torch.manual_seed(42)
ds, model = bench.build_model("c3")
loop = PretrainLoop(model, 32, dev, total_steps=100)
batch = synthetic_batch(ds.dataset, 32, dev)
eng = loop.engine
starts, ends = [], []
inner = eng._segment
def seg(name, key, fn):
    if name.startswith("forward"):
        ev = torch.cuda.Event(enable_timing=True); ev.record(); starts.append(ev)
    return inner(name, key, fn)
eng._segment = seg
for it in range(30):
    loop.step(batch)
    ev = torch.cuda.Event(enable_timing=True); ev.record(); ends.append(ev)
torch.cuda.synchronize()
gaps = [ends[i].elapsed_time(starts[i + 1]) * 1e3 for i in range(8, 29)]
steps = [ends[i].elapsed_time(ends[i + 1]) for i in range(8, 29)]
print(f"UPLOAD_STREAM={os.environ.get('MAESTRO_UPLOAD_STREAM', '1')} GRAPHS={os.environ.get('MAESTRO_GRAPHS', '1')}: step end -> first forward launch: median {sorted(gaps)[len(gaps) // 2]:.1f} us (min {min(gaps):.1f}, max {max(gaps):.1f}); step median {sorted(steps)[len(steps) // 2]:.3f} ms")
