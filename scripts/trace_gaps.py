"""Idle gaps of the GPU (no kernel running on any stream) from a rocprofv3 --kernel-trace CSV: total, histogram, and the largest
gaps per step with the kernels before / after them.  usage: python scripts/trace_gaps.py <kernel_trace.csv> <steps>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_lo, t_hi = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t_lo + (t_hi - t_lo) * 0.5
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
gaps, cur_e, last = [], None, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_e is not None and s > cur_e:
        gaps.append(((s - cur_e) / 1e3, short(last["Kernel_Name"]), short(r["Kernel_Name"])))
    if cur_e is None or e > cur_e:
        cur_e, last = e, r
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
n_adamw = sum(1 for r in rows if "adamw" in r["Kernel_Name"])
print(f"window {span:.2f} ms = {n_adamw} steps; idle {sum(g[0] for g in gaps) / 1e3:.2f} ms in {len(gaps)} gaps = {sum(g[0] for g in gaps) / max(n_adamw, 1):.0f} us per step")
hist = collections.Counter()
for g in gaps:
    hist["<2us" if g[0] < 2 else "2-5us" if g[0] < 5 else "5-20us" if g[0] < 20 else "20-100us" if g[0] < 100 else ">=100us"] += g[0]
print({k: round(v / max(n_adamw, 1), 1) for k, v in hist.items()}, "us per step by gap size")
by = collections.defaultdict(lambda: [0, 0.0])
for us, a, b in gaps:
    by[(a, b)][0] += 1; by[(a, b)][1] += us
for (a, b), (c, us) in sorted(by.items(), key=lambda x: -x[1][1])[:25]:
    print(f"{us / max(n_adamw, 1):8.1f} us/step  x{c / max(n_adamw, 1):5.1f}  avg {us / c:7.1f} us   {a}  ->  {b}")
