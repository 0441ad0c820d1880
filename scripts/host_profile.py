import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
ds, model = bench.build_model("c3")
loop = PretrainLoop(model, 32, dev, total_steps=100)
batch = synthetic_batch(ds.dataset, 32, dev)
for _ in range(5): loop.step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(30): loop.step(batch)
t1 = time.perf_counter()
pr.disable(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/30:.2f} ms/step, total {1e3*(t2-t0)/30:.2f} ms/step, torch threads {torch.get_num_threads()}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
