"""Diagnostic: C5 at ViT-B width in fp8 against the fp32 oracle over several forward passes (delayed scaling: pass 1 casts the
activations with scale 1, later passes with scales derived from the previous pass' absmax).  Prints, per pass, the relative loss
error, the per-modality reconstruction error and the worst parameter-gradient error.  python scripts/fp8_c5_diag.py [stress]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import maestro_amd.conf as conf  # noqa: E402
from maestro_amd.ssl import mae as pmae  # noqa: E402
from maestro_amd.train.trainer import synthetic_batch  # noqa: E402
from oracle import mae as om  # noqa: E402
from oracle.gen_golden import init_weights, stress_raster  # noqa: E402

COMMON = dict(interpolate="nearest", fusion_mode="group", inter_depth=3, model="mae", num_levels=1)
stress = "stress" in sys.argv
dtype = "bf16" if "bf16" in sys.argv else "fp8"
dev = torch.device("cuda:0")
w = bench.WORKLOADS["c5"]
ds = w["ds"]()
torch.set_float32_matmul_precision("highest")
oracle = om.build_oracle(ds, conf.MaskConfig(), model_size=w["size"], **COMMON)
init_weights(oracle, 102)
model = getattr(pmae, f"mae_{w['size']}")(datasets=ds, mask=conf.MaskConfig(), **COMMON)
model.load_state_dict(oracle.state_dict(), strict=True)
B = 2
batch = synthetic_batch(ds.dataset, B, "cpu", seed=3)
if stress:
    g = torch.Generator().manual_seed(99)
    for m, c in ds.dataset.inputs.items():
        batch[m] = stress_raster(batch[m], c.patch_size.mae, g)
eng = model.engine(B, dev, loss="l2_norm", dtype=None if dtype == "bf16" else "fp8")
torch.manual_seed(17)
noise, struct = eng.draw_masks()
ob, orec, omsk, _ = oracle({k: v.clone() for k, v in batch.items()}, "pretrain", noise=noise,
                           struct_masks={g: s[:, :, None] for g, s in struct.items()})
oloss = om.compute_loss_rec(ob, orec, omsk, oracle.out_grid_size, om.norm_bands_of(ds.dataset), "l2_norm")
oracle.zero_grad()
oloss.backward()
ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
gmax = max(g.abs().max().item() for g in ograds.values())
dbatch = {k: v.to(dev) for k, v in batch.items()}
for it in range(3):
    loss = eng.forward(dbatch, noise=noise, struct=struct).clone()
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    pixels, masks = eng.reconstructions()
    rel = lambda a, b: ((a - b).double().norm() / b.double().norm().clamp(min=1e-12)).item()  # noqa: E731
    pix = {m: round(rel(pixels[m].cpu(), orec[m].detach()), 5) for m in orec}
    worst = (0.0, None)
    for k, p in model.named_parameters():
        if k in ograds:
            got, want = eng.store.g(p).cpu(), ograds[k]
            err, ref = (got - want).double().norm().item(), want.double().norm().item()
            if ref > 1e-4 * gmax * want.numel() ** 0.5 and err / ref > worst[0]:
                worst = (err / ref, k)
    print(f"[{dtype}{' stress' if stress else ''}] pass {it + 1}: loss {loss.item():.6f} vs {oloss.item():.6f} rel {abs(loss.item() - oloss.item()) / abs(oloss.item()):.2e}; "
          f"pixels {pix}; worst grad {worst[0]:.3e} {worst[1]}", flush=True)
    if eng.fp8 is not None:
        sc = eng.fp8.asc.scale
        print(f"    activation scales: min {float(sc.min()):.3g} max {float(sc.max()):.3g}; weight scales min {float(eng.fp8.wsc.scale.min()):.3g} "
              f"max {float(eng.fp8.wsc.scale.max()):.3g}", flush=True)
