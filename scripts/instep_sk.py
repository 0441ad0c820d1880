"""Round 6: the stream-K tiles INSIDE the step.  MAESTRO_INSTEP_TUNE=1 MAESTRO_INSTEP_SK=1: the first step is recomputed once per candidate
tile (same inputs and draws, eager, one stream), every GEMM launch bracketed by HIP events; prints, per GEMM signature the stream-K
tiles serve, the summed time of its launches under the rule and under each stream-K tile, and what the tuner would keep (3 % margin)."""
import os
import sys

os.environ["MAESTRO_INSTEP_TUNE"] = "1"
os.environ["MAESTRO_INSTEP_SK"] = "1"
os.environ["MAESTRO_WARM_PASSES"] = "0"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from maestro_amd import hip  # noqa: E402
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
torch.manual_seed(42)
ds, model = bench.build_model(cfg)
loop = PretrainLoop(model, 32, dev, total_steps=10)
batch = synthetic_batch(ds.dataset, 32, dev)
loop.step(batch)
torch.cuda.synchronize()
rep = getattr(loop.engine, "tile_report", {})
names = {hip.TILE_AUTO: "rule", hip.TILE_SK_DMA_256: "skd256", hip.TILE_SK_192: "sk192", hip.TILE_SK_256: "sk256", hip.TILE_REG_128: "reg128",
         hip.TILE_PP_128: "pp128", hip.TILE_REG_64: "reg64", hip.TILE_REG_192: "reg192", hip.TILE_DMA_256: "d256"}
tot = {}
print(f"{'layout (M, N, K) flags':40s} " + " ".join(f"{n:>8s}" for n in names.values()) + "   kept")
for key, (pick, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1].get(hip.TILE_AUTO, 0)):
    lay, M, N, K, fl = key
    if not any(t in ms for t in hip.SK_TILES):
        continue
    print(f"{('NT', 'NN', 'TN')[lay]} ({M}, {N}, {K}) 0x{fl:x}".ljust(40) + " " +
          " ".join(f"{ms[t]:8.3f}" if t in ms else "       -" for t in names) + f"   {names[pick]}")
    for t in names:
        if t in ms:
            tot.setdefault(t, 0.0)
    base = ms[hip.TILE_AUTO]
    for t in names:
        tot[t] = tot.get(t, 0.0) + ms.get(t, base)
print("sum over these signatures, ms per step (a tile that does not serve a signature counts the rule's time):")
print(" ".join(f"{names[t]} {v:.3f}" for t, v in tot.items()))
