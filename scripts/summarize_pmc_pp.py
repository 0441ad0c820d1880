"""gpurun_out/pmc_pp/*.csv (scripts/pmc_pp.sh) -> profiles/r03_gemm_pmc.txt: per-launch averages of the SQ counters of the fc1 GEMM
(8192 x 3072 x 768) on the one-tile-per-workgroup kernel and on the persistent ping-pong tile."""
import collections, csv, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / "pmc_pp"
variants = ["reg_plain", "reg_gelu", "pp_plain", "pp_gelu", "pp_main"]
data = {v: collections.defaultdict(list) for v in variants}
for v in variants:
    for f in sorted(src.glob(f"{v}_*.csv")):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                data[v][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in variants for c in data[v]})
lines = ["rocprofv3 --pmc <4 SQ counters per pass> -- python3 scripts/gemm_one.py 0 8192 3072 768 tile<k> {plain|gelu}   (bash scripts/pmc_pp.sh)",
         "fc1 GEMM of the C3 step (M 8192, N 3072, K 768), 20 launches per pass, averages per launch.  reg = gemm_kernel<NT> (1536 workgroups, exposed",
         "LDS-staged epilogue); pp = gemm_pp_kernel<NT> (512 persistent workgroups, epilogue of tile t inside the main loop of tile t + 1);",
         "'gelu' = bias + GELU + byte-coded derivative, 'plain' = bf16 store only, 'pp_main' = the diagnostic build without any epilogue.", "",
         f"{'counter':28s}" + "".join(f"{v:>14s}" for v in variants)]
avg = {v: {c: (sum(x) / len(x) if x else float('nan')) for c, x in data[v].items()} for v in variants}
for c in names:
    lines.append(f"{c:28s}" + "".join(f"{avg[v].get(c, float('nan')):14.4e}" for v in variants))
lines.append("")
for v in variants:
    a = avg[v]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CU_CYCLES" in a:
        lines.append(f"{v:10s} MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * a['SQ_BUSY_CU_CYCLES']):.3f};"
                     f"  VALU instructions per MFMA = {a.get('SQ_INSTS_VALU', float('nan')) / a.get('SQ_INSTS_MFMA', float('nan')):.2f};"
                     f"  wave cycles {a.get('SQ_WAVE_CYCLES', float('nan')):.3e}")
(root / "profiles" / "r03_gemm_pmc.txt").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
