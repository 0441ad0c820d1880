"""Turn rocprofv3 CSVs under gpurun_out/prof into the committed summaries under profiles/.

usage: python scripts/summarize_profile.py r01 <steps_in_trace>
  gpurun_out/prof/<tag>_kernel_stats.csv            -> profiles/<tag>_kernel_stats.csv  (verbatim copy)
  gpurun_out/prof/<tag>_{fetch,write}_counter_collection.csv -> profiles/<tag>_hbm_traffic.json
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Separate --pmc passes as the guide prescribes.
"""
import collections, csv, json, shutil, sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
tag, steps = sys.argv[1], int(sys.argv[2])
src, dst = root / "gpurun_out" / "prof", root / "profiles"
dst.mkdir(exist_ok=True)
shutil.copy(src / f"{tag}_kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")

_LAYOUT = {("false", "false"): "NT", ("false", "true"): "NN", ("true", "true"): "TN"}
_TILE = {"2, 4, 4, 8": "256x256", "2, 2, 3, 8": "256x128", "1, 4, 3, 8": "128x256", "1, 2, 4, 8": "128x128",
         "2, 2, 4, 4": "128x128q"}


def short(name):
    """rocprof kernel name -> the label maestro_amd.hip.KernelTimer / bench.py use for the same kernel."""
    import re
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gemm_kernel<(\w+), (\w+)(?:, (\d+))?>", name)     # (+ the tile height in 32-row units since round 3)
    if m:
        return f"gemm_kernel<{_LAYOUT[m.groups()[:2]]}{ {'2': ',64x128', '6': ',192x128'}.get(m.group(3), '') }>"
    m = re.match(r"gemm_dma_kernel<Tile<([\d, ]+)>, (\w+), (\w+)(?:, \w+)?>", name)   # (+ the STAGGER flag since round 2)
    if m:
        return f"gemm_dma_kernel<{_TILE[m.group(1)]},{_LAYOUT[(m.group(2), m.group(3))]}>"
    m = re.match(r"gemm_pp_kernel<(\w+), \d+(?:, \d+)?>", name)      # (B_KMAJOR, epilogue form[, diagnostic build])
    if m:
        return f"gemm_pp_kernel<{'NN' if m.group(1) == 'true' else 'NT'}>"
    if name.startswith("gemm_dma_grouped_tn_kernel"):
        return "gemm_dma_grouped_tn_kernel"
    return name.split("(")[0]


agg = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
for kind, col in (("fetch", "FETCH_SIZE_KB"), ("write", "WRITE_SIZE_KB")):
    f = src / f"{tag}_{kind}_counter_collection.csv"
    if not f.exists():
        continue
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][col] += float(r["Counter_Value"])
        n[k] += 1
    for k, c in n.items():
        agg[k]["launches"] = max(agg[k]["launches"], c)
out = {}
for k, v in agg.items():
    if not v["launches"]:
        continue
    f, w = v["FETCH_SIZE_KB"] / v["launches"], v["WRITE_SIZE_KB"] / v["launches"]
    out[k] = {"launches_in_trace": v["launches"], "fetch_kb_per_launch_raw": round(f, 1), "write_kb_per_launch": round(w, 1),
              "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
PMC_STEPS = 5   # profile_round.sh runs the counter passes with --steps 2 --warmup 3
per_step = sum(v["hbm_bytes_per_launch"] * v["launches_in_trace"] for v in out.values()) / PMC_STEPS
import subprocess
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip() or "unknown"
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "maestro_amd", "bench.py"], cwd=root, capture_output=True, text=True).stdout.strip())
json.dump({"note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE correction applied", "steps_in_trace": PMC_STEPS,
           "commit": head + ("+uncommitted changes" if dirty else ""),
           "hbm_bytes_per_step": int(per_step), "kernels": out},
          open(dst / f"{tag}_hbm_traffic.json", "w"), indent=1, sort_keys=True)
rows = list(csv.DictReader(open(src / f"{tag}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{tag}: {tot / 1e6 / steps:.2f} ms of kernel time per step over {steps} steps")
for r in rows[:10]:
    print(f"  {short(r['Name'])[:40]:40s} {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
