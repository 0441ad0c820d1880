"""Table of scripts/tile_sweep.sh: per (layout, M, N, K) the in-step ms/step under every forced tile, the best one, and what AUTO took."""
import re, sys
from pathlib import Path
phase = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
d = Path(__file__).resolve().parent.parent / "gpurun_out" / f"sweep_{phase}"
names = {"auto": "auto", "0": "reg128", "1": "d256", "7": "pp128", "13": "reg64", "14": "reg192"}
tab = {}
for t, nm in names.items():
    for line in (d / f"tile_{t}.shapes").read_text().splitlines():
        m = re.match(r"\s*([\d.]+) ms/step (\S+)\s+\((\d+), (\d+), (\d+)\)\s+x\s*(\d+)/step", line)
        if not m or "gemm" not in m.group(2):
            continue
        lay = re.search(r"<(?:[^,>]*,)?\s*(NT|NN|TN)", m.group(2))
        lay = lay.group(1) if lay else "?"
        key = (lay, int(m.group(3)), int(m.group(4)), int(m.group(5)))
        e = tab.setdefault(key, {})
        e[nm] = e.get(nm, 0.0) + float(m.group(1))
        if nm == "auto":
            e["auto_kernel"] = m.group(2)
tot = {n: 0.0 for n in names.values()}
best_tot = 0.0
for key, e in sorted(tab.items(), key=lambda kv: -kv[1].get("auto", 0)):
    if "auto" not in e:
        continue
    cand = {n: e[n] for n in names.values() if n in e}
    best = min(cand, key=cand.get)
    best_tot += cand[best]
    for n in tot:
        tot[n] += e.get(n, e["auto"])
    print(f"{key[0]} {str(key[1:]):22s} " + " ".join(f"{n} {e.get(n, float('nan')):6.3f}" for n in names.values()) +
          f" | best {best:7s} {100 * (e['auto'] / cand[best] - 1):+5.1f}% vs auto ({e.get('auto_kernel')})")
print("sums:", {n: round(v, 3) for n, v in tot.items()}, "best-of:", round(best_tot, 3))
