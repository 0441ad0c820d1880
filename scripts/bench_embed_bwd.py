"""mh_embed_finish_bwd (GroupNorm backward of the patch embed: statistics pass + apply pass) on the C3 step's two launches, us per call."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maestro_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for (B, D, L, E) in ((32, 1, 1024, 768), (32, 16, 25, 768), (32, 4, 225, 1024)):
    Lg = D * L
    dxg = torch.randn(B, Lg, E, device=dev)
    y = torch.randn(B * D, L, E, device=dev)
    stats = torch.rand(B * D, 2, device=dev) + 0.5
    gamma = torch.randn(E, device=dev)
    dyc = torch.empty(B * D * L, E, dtype=torch.bfloat16, device=dev)
    dg, db, sums = torch.zeros(E, device=dev), torch.zeros(E, device=dev), torch.zeros(B * D, 2, device=dev)
    t = timed(lambda: hip.embed_finish_bwd(dxg, y, stats, gamma, dyc, dg, db, sums, B, D, L, E, 0, Lg))
    mb = B * D * L * E * (4 + 4 + 4 + 4 + 2) / 1e6
    print(f"B {B} D {D} L {L} E {E}: {t:7.1f} us  ({mb:.0f} MB over both passes: {mb / t / 1e3 * 1e3:.0f} GB/s)", flush=True)
