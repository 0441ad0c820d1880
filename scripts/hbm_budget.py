"""The step's HBM budget: per kernel family, bytes per launch from the PMC passes (profiles/rNN_hbm_traffic.json:
2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction) against ALGORITHMIC bytes -- operands read once, outputs written once --
computed from the C3 step's shape list (profiles/rNN_shapes.txt = `bench.py --shapes`) and the epilogue each shape runs with.
usage: python scripts/hbm_budget.py r04   -> profiles/r04_hbm_budget.md"""
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

root = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
pmc = json.load(open(root / "profiles" / f"{tag}_hbm_traffic.json"))
kern = pmc["kernels"]
STEPS = pmc["steps_in_trace"]
E, DEC, MLP = 768, 512, 3072          # C3: ViT-B encoder width, decoder width, MLP width (decoder MLP = E x ratio)


def gemm_bytes(name, M, N, K):  # noqa: N803
    """Algorithmic bytes of one GEMM launch of the step, by layout and shape (which epilogue runs on it)."""
    a, b = 2 * M * K, 2 * N * K
    nt, nn, tn = "<NT" in name or ",NT>" in name, "<NN" in name or ",NN>" in name, "<TN" in name
    if tn:                                        # in-line weight gradient: two bf16 operands, fp32 output
        return 2 * K * (M + N) + 4 * M * N, "wgrad: dY^T X, fp32 out"
    if nt and N == MLP:                           # fc1: bias + GELU, bf16 out + one byte of GELU'
        return a + b + 3 * M * N, "fc1: bf16 out + u8 GELU'"
    if nt and N in (E, DEC) and K >= 512:         # out-proj / fc2 / enc->dec: fp32 out + fp32 residual in
        return a + b + 8 * M * N, "fp32 out + fp32 residual"
    if nn and N == MLP:                           # fc2 dgrad: x GELU' (u8 in), bf16 out, column sums
        return a + b + 3 * M * N, "fc2 dgrad: bf16 out, u8 GELU' in"
    return a + b + 2 * M * N, "bf16 out"


rows = []                                         # (family, launches/step, algorithmic bytes/step)
fam = defaultdict(lambda: [0.0, 0.0, set()])
for ln in open(root / "profiles" / f"{tag}_shapes.txt"):
    m = re.match(r"\s*([\d.]+) ms/step (\S+)\s+\(([^)]*)\)\s+x\s*(\d+)/step", ln)
    if not m:
        continue
    name, shape, n = m.group(2), m.group(3), int(m.group(4))
    if name.startswith("gemm") and "grouped" not in name:
        M, N, K = (int(x) for x in shape.split(","))  # noqa: N806
        by, what = gemm_bytes(name, M, N, K)
        fam[name][0] += n
        fam[name][1] += n * by
        fam[name][2].add(what)
    elif name in ("attn_fwd", "attn_bwd"):
        B, N, H, D = (int(x) for x in shape.split(","))  # noqa: N806
        qkv, o = 2 * B * N * 3 * H * D, 2 * B * N * H * D
        if name == "attn_fwd":
            fam["attn_fwd_kernel"][0] += n
            fam["attn_fwd_kernel"][1] += n * (qkv + o + 4 * B * H * N)
        else:                                     # two launches: dQ (qkv, o, do -> dq), dK/dV (qkv, do -> dk, dv)
            fam["attn_bwd (dq + dkv)"][0] += n
            fam["attn_bwd (dq + dkv)"][1] += n * ((qkv + 2 * o + qkv // 3) + (qkv + o + 2 * qkv // 3))

# the fixed-size passes
P = 176.2e6
fam["adamw_kernel"][0], fam["adamw_kernel"][1] = 1, 30 * P          # p, g, m, v read; p, m, v, bf16 shadow written
# grouped weight gradients: every (X, dY) pair read once, fp32 gradient written: sum over the step's GEMM weights
wg = 0.0
for ln in open(root / "profiles" / f"{tag}_shapes.txt"):
    m = re.match(r"\s*[\d.]+ ms/step (gemm\S+)\s+\(([^)]*)\)\s+x\s*(\d+)/step", ln)
    if m and "grouped" not in m.group(1) and ("<NT" in m.group(1) or ",NT>" in m.group(1)) :
        M, N, K = (int(x) for x in m.group(2).split(","))  # noqa: N806
        if K >= 512 and N >= 512:                 # a transformer-layer forward GEMM: its weight gradient is in the grouped launch
            wg += int(m.group(3)) * (2 * M * (N + K) + 4 * N * K)
fam["gemm_dma_grouped_tn_kernel"][0], fam["gemm_dma_grouped_tn_kernel"][1] = 1, wg


def measured(prefix):
    tot = n = 0
    for k, v in kern.items():
        if k.startswith(prefix):
            tot += v["hbm_bytes_per_launch"] * v["launches_in_trace"] / STEPS
            n += v["launches_in_trace"] / STEPS
    return tot, n


lines = []
for name, (n, alg, what) in fam.items():
    key = "attn_bwd_d" if name.startswith("attn_bwd") else name
    meas, nl = measured(key)
    lines.append((meas - alg, name, n, nl, alg, meas, "; ".join(sorted(what)) if what else ""))
lines.sort(reverse=True)
out = [f"# HBM budget of the C3 step (B = 32), {tag}: PMC bytes vs algorithmic bytes per kernel family, sorted by wasted GB per step",
       "",
       f"PMC collection: profiles/{tag}_hbm_traffic.json (commit {pmc.get('commit', '?')}): 2 x FETCH_SIZE + WRITE_SIZE per launch, separate passes "
       f"of `bench.py --steps 2 --warmup 3`; whole step {pmc['hbm_bytes_per_step'] / 1e9:.1f} GB.  FETCH_SIZE counts L2 misses, i.e. traffic that "
       "the 256 MiB Infinity Cache may still absorb: 'measured' is an upper bound on DRAM bytes.  Algorithmic = every operand read once, every "
       "output written once, with the epilogue the shape runs with (scripts/hbm_budget.py).",
       "",
       "| kernel family | launches / step | algorithmic GB / step | measured GB / step | ratio | wasted GB / step | note |",
       "|---|---|---|---|---|---|---|"]
ta = tm = 0.0
for waste, name, n, nl, alg, meas, what in lines:
    ta += alg
    tm += meas
    out.append(f"| `{name}` | {n:.0f} | {alg / 1e9:.2f} | {meas / 1e9:.2f} | {meas / max(alg, 1):.2f} | {waste / 1e9:+.2f} | {what} |")
rest = pmc["hbm_bytes_per_step"] - tm
out.append(f"| everything else (LayerNorm, embed, masks, loss, casts, copies, fills) | | | {rest / 1e9:.2f} | | | see below |")
out.append(f"| **sum** | | {ta / 1e9:.1f} (+ rest) | {pmc['hbm_bytes_per_step'] / 1e9:.1f} | | {(tm - ta) / 1e9:+.1f} | |")
out += ["", "Elementwise / reduction kernels (measured only; their algorithmic bytes are in DESIGN §4 per element):", "",
        "| kernel | launches / step | MB / launch | GB / step |", "|---|---|---|---|"]
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_in_trace"]):
    if k.startswith(("gemm", "attn_", "adamw")):
        continue
    gb = v["hbm_bytes_per_launch"] * v["launches_in_trace"] / STEPS / 1e9
    if gb >= 0.05:
        out.append(f"| `{k[:70]}` | {v['launches_in_trace'] / STEPS:.1f} | {v['hbm_bytes_per_launch'] / 1e6:.1f} | {gb:.2f} |")
out += ["",
        "Reading.  (1) 14 GB of the step's 63 GB are GEMM operand re-fetches: the 128 x 128 kernels (`gemm_kernel`, `gemm_pp_kernel`) move 1.6-1.9x "
        "their algorithmic bytes -- 64 workgroups under one XCD's 4 MiB L2 share an 8 x 8 block of tiles whose 16 operand panels (3.1 MiB at K = 768) are "
        "the most that fits, so every tile re-fetches 49 KiB; those re-fetches are Infinity-Cache hits (a launch's operands, <= 100 MB, stay resident), "
        "and the all-L1-hit experiment of round 3 (profiles/r03_gemm_l1hit.txt: 6-10 % faster) bounds what removing them could buy.  The 256 x 256 "
        "LDS-DMA tiles sit at 1.2-1.3x; the grouped weight-gradient launch at 1.32x.  (2) Attention and AdamW are at 1.00-1.02x: nothing to recover. "
        "(3) `FillFunctor` (0.6 GB 'per step') and most of `__amd_rocclr_copyBuffer` are NOT step traffic: they are the engine's one-time buffer "
        "allocations (`torch.zeros`) and the parameter copies into the flat buffer, divided by the 5 steps of the counter pass; a steady-state step "
        "issues no torch fill.  (4) The largest non-GEMM item is the LayerNorm backward (5.9 GB per step, 16 bytes per element by construction: "
        "dy, x, the residual gradient in; dx fp32 + bf16 out), then the LayerNorm forward (2.3 GB, 6 bytes per element).  Target of the round-3 review "
        "(whole step <= 55 GB): not reached -- the only lever of that size is the 128 x 128 tiles' re-fetch (8 GB), which is cache traffic that the "
        "main loop does not wait for; fixing it means 256-wide tiles on the K = 768 shapes, which lose to their tile-count quantisation (DESIGN §4)."]
(root / "profiles" / f"{tag}_hbm_budget.md").write_text("\n".join(out) + "\n")
print("\n".join(out))
