"""Would a single-chain part of the step run faster as TWO half-batch chains on two streams?  (round 6, experiment 16)
The joint encoder of C3 (3 layers, 11392 rows) and the aerial decoder run alone on the card; every kernel in them waits for the one
before it.  Here a stack's forward + dgrad chain is captured twice into a hipGraph -- once as the engine runs it, once as two
half-batch views (rows [0, M/2) and [M/2, M), same weights, same buffers) on two streams -- and replayed.
usage: python scripts/split_stack.py [c3] [joint|dec:<name>|enc:<group>]"""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from maestro_amd import hip
from maestro_amd.engine import Stack
from maestro_amd.train.trainer import PretrainLoop, synthetic_batch

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
which = sys.argv[2] if len(sys.argv) > 2 else "joint"
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
eng = loop.engine
print("stacks:", list(eng.enc), list(eng.dec))
if which == "joint":
    st = eng.joint
else:
    kind, name = which.split(":")
    st = (eng.dec if kind == "dec" else eng.enc)[name]
M, Bn = st.M, st.Bn
print(f"{cfg} {which}: M = {M} rows ({Bn} x {st.N}), dim {st.dim}, depth {st.depth}, heads {st.H} x {st.Dh}")


def view(st, part, parts):
    """A Stack-shaped object over rows [part * M / parts, (part + 1) * M / parts) of every buffer."""
    v = types.SimpleNamespace(**st.__dict__)
    v.Bn, v.M = st.Bn // parts, st.M // parts
    assert st.Bn % parts == 0 and v.M % 64 == 0

    def cut(t):
        if not isinstance(t, torch.Tensor) or t.dim() == 0 or t.shape[0] % parts:
            return t
        n = t.shape[0] // parts
        return t[part * n:(part + 1) * n]
    v.xs = [cut(t) for t in st.xs]
    v.saved = [{k: cut(t) for k, t in s.items()} for s in st.saved]
    for k in ("dxa", "dxb", "dx0_16", "dh2", "do", "delta", "cs_ws"):
        setattr(v, k, cut(getattr(st, k)))
    v.ln_ws = st.ln_ws        # (only the in-line weight-gradient path uses it; the chain below defers)
    v.f8 = None
    return v


def chain(v):
    Stack.forward(v)
    Stack.backward(v, v.dxa, defer=True, ready=False)


s_a, s_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def split(parts):
    views = [view(st, p, parts) for p in range(parts)]
    streams = [s_a, s_b, torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)][:parts]

    def run():
        main = torch.cuda.current_stream()
        for s, v in zip(streams, views):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                chain(v)
        for s in streams:
            main.wait_stream(s)
    return run


def capture(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cs):
        g.capture_begin()
        fn()
        g.capture_end()
    return g


def timed(g, reps=15):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


full = lambda: chain(st)
g1, g2 = capture(full), capture(split(2))
t1, t2 = timed(g1), timed(g2)
t1b, t2b = timed(g1), timed(g2)
print(f"forward + dgrad chain, one stream      {t1:7.3f} / {t1b:7.3f} ms")
print(f"two half-batch chains on two streams   {t2:7.3f} / {t2b:7.3f} ms   ({100 * (t2 + t2b) / (t1 + t1b) - 100:+.1f} %)")
if st.Bn % 4 == 0 and (st.M // 4) % 64 == 0:
    g4 = capture(split(4))
    t4 = timed(g4)
    print(f"four quarter-batch chains              {t4:7.3f} ms   ({100 * t4 / t1 - 100:+.1f} %)")
