"""Throughput and host issue time of the DROP-IN surface as Lightning drives it (SSLModule.training_step -> loss.backward() ->
optimizer.step() -> scheduler.step(), optimizer from configure_optimizers()), C3 at B = 32, next to the built-in PretrainLoop."""
import cProfile, os, pstats, sys, time, torch
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import maestro_amd.conf as conf
from maestro_amd.train.model import SSLModule
from maestro_amd.train.trainer import synthetic_batch

dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
STEPS = 30
torch.manual_seed(42)
ds, _ = bench.build_model("c3")
mod = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=3, model="mae",
                model_size="medium", loss="l2_norm", use_ema=False)
mod.trainer = SimpleNamespace(ssl_phase="pretrain", train_dataloader=SimpleNamespace(batch_size=32), accumulate_grad_batches=1,
                              num_nodes=1, num_devices=1, base_lr=3e-5, wd=0.01, b1=0.9, b2=0.99, final_factor=1e7,
                              estimated_stepping_batches=1000, max_epochs=5)
mod.log = lambda *a, **k: None
batch = synthetic_batch(ds.dataset, 32, dev)
cfg = mod.configure_optimizers()
opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
print("optimizer:", type(opt).__name__, {k: v for k, v in opt.defaults.items() if k in ("fused", "foreach")}, "torch threads", torch.get_num_threads())


def step(i):
    out = mod.training_step(batch, i)
    opt.zero_grad(set_to_none=True)
    out["loss"].backward()
    opt.step()
    sched.step()


for i in range(6):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for i in range(STEPS):
    step(i)
t1 = time.perf_counter()
pr.disable()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"Lightning-style step: {32 * STEPS / (t2 - t0):.1f} tiles/s, {1e3 * (t2 - t0) / STEPS:.2f} ms/step, host issue {1e3 * (t1 - t0) / STEPS:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
