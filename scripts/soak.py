"""Longer training runs on the bench workload (synthetic C3 / C5 batches, a few fixed batches cycled): loss trajectory in bf16 and
fp8 side by side, finiteness, peak memory -- `python scripts/soak.py [config] [steps]`."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
out = {}
for dtype in ("bf16", "fp8"):
    torch.manual_seed(42)
    ds, model = bench.build_model(config, "pretrain")
    loop = PretrainLoop(model, 32, dev, total_steps=steps, dtype=dtype)
    batches = [synthetic_batch(ds.dataset, 32, dev, seed=s) for s in range(4)]
    torch.manual_seed(7)
    losses, t0 = [], time.time()
    for it in range(steps):
        loss = loop.step(batches[it % 4])
        if it % 25 == 0 or it == steps - 1:
            losses.append((it, round(float(loss.item()), 5)))
    torch.cuda.synchronize()
    out[dtype] = dict(losses=losses, seconds=round(time.time() - t0, 2), peak_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2),
                      finite=all(l == l and abs(l) < 1e6 for _, l in losses))
    print(dtype, json.dumps(out[dtype]), flush=True)
    del loop, model
    torch.cuda.empty_cache()
a, b = dict(out["bf16"]["losses"]), dict(out["fp8"]["losses"])
print("max |loss_fp8 - loss_bf16| / loss_bf16 over the logged steps:", max(abs(a[k] - b[k]) / abs(a[k]) for k in a))
