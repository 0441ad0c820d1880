"""Single-GPU cost of the data-parallel launch plan (gradient hook set, no communication): segments + per-chunk grouped wgrads."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
dev = torch.device("cuda:0")
torch.set_num_threads(4)
torch.manual_seed(0)
ds, model = bench.build_model("c3")
loop = PretrainLoop(model, 32, dev, total_steps=100)
batch = synthetic_batch(ds.dataset, 32, dev)
for hook in (False, True, False, True):
    loop.engine.grad_hook = (lambda lo, hi: None) if hook else None
    for _ in range(6): loop.step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): loop.step(batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"hook={hook}: {32 * 30 / dt:7.1f} tiles/s  plan={loop.engine._plan}", flush=True)
