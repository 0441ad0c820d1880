"""Per-(kernel, grid) durations, busy-union time and idle gaps from a rocprofv3 --kernel-trace CSV.
usage: python scripts/trace_breakdown.py <kernel_trace.csv> <steps> [skip_fraction]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_lo = int(rows[0]["Start_Timestamp"]); t_hi = int(rows[-1]["End_Timestamp"])
cut = t_lo + (t_hi - t_lo) * skip            # keep the steady-state tail of the run (graph replays)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]
agg = collections.defaultdict(lambda: [0, 0.0])
busy, cur_s, cur_e = 0.0, None, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e6
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += (cur_e - cur_s) / 1e6
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += (cur_e - cur_s) / 1e6
tot = sum(v[1] for v in agg.values())
print(f"window {span:.2f} ms, {len(rows)} dispatches; sum of kernel durations {tot:.2f} ms; busy (union) {busy:.2f} ms; idle {span - busy:.2f} ms")
byname = collections.defaultdict(float)
for (n, gx, gy), (c, ms) in agg.items(): byname[n] += ms
print("-- by kernel (share of summed durations)")
for n, ms in sorted(byname.items(), key=lambda x: -x[1])[:25]: print(f"{ms:9.3f} ms {100*ms/tot:5.1f}%  {n}")
print("-- by kernel and grid")
for (n, gx, gy), (c, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:45]:
    print(f"{ms:9.3f} ms  x{c:5d}  avg {1e3*ms/c:8.1f} us  grid ({gx},{gy})  {n}")
