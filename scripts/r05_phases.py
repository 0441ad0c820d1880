"""Wall time of the step's phases (HIP events on the main stream, eager launches): the group-parallel sections as they run (two
streams) and group by group (one stream), the joint encoder, the backward segments, the grouped weight gradients, AdamW.
   python scripts/r05_phases.py [config]"""
import os, sys, collections, torch
os.environ["MAESTRO_GRAPHS"] = "0"
os.environ["MAESTRO_WARM_PASSES"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
torch.set_num_threads(4); torch.manual_seed(42)
ds, model = bench.build_model(cfg)
loop = PretrainLoop(model, 32, dev, total_steps=100)
batch = synthetic_batch(ds.dataset, 32, dev)
eng = loop.engine
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((name, ev))
orig_par = eng._run_parallel
mode = {"serial": False}
def run_parallel(fns):
    mark(f"par{len(fns)}:begin")
    if mode["serial"]:
        for i, fn in enumerate(fns):
            fn(); mark(f"par:group{i}")
    else:
        orig_par(fns)
    mark("par:end")
eng._run_parallel = run_parallel
for name in ("_bwd_decoder_side", "_bwd_joint", "_bwd_encoder_side", "_launch_wgrads"):
    orig = getattr(eng, name)
    def wrap(orig=orig, name=name):
        def f(*a, **k):
            mark(name + ":begin"); r = orig(*a, **k); mark(name + ":end"); return r
        return f
    setattr(eng, name, wrap())
def one_step():
    marks.clear()
    mark("step:begin")
    loss = eng.forward(batch); mark("forward:end")
    eng.zero_grad(); eng.backward(); mark("backward:end")
    loop._optimizer_step(1.0); loop.it += 1; mark("adamw:end")
for serial in (False, True):
    mode["serial"] = serial
    acc = collections.OrderedDict()
    for it in range(8):
        one_step(); torch.cuda.synchronize()
        if it < 3: continue
        for i, ((n0, e0), (n1, e1)) in enumerate(zip(marks, marks[1:])):
            k = f"{i:2d} {n0} -> {n1}"
            acc.setdefault(k, []).append(e0.elapsed_time(e1))
        acc.setdefault("TOTAL", []).append(marks[0][1].elapsed_time(marks[-1][1]))
    print(("== groups one after the other (one stream)" if serial else "== groups on parallel streams (the step as it runs)"))
    seen = collections.OrderedDict()
    for k, v in acc.items():
        if sum(v) / len(v) >= 0.02:
            print(f"  {k:60s} {sum(v) / len(v):7.3f} ms")
