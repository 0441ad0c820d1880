"""Idle time inside the step from a rocprofv3 kernel trace: python scripts/step_bubbles.py <kernel_trace.csv> [steps_to_skip]
Takes the dispatches of the trace in time order, cuts it into steps at the AdamW launches, and reports per step: wall time between
the first start and the last end, the union of the kernels' busy intervals, the idle remainder, and the ten longest idle gaps with the
kernels on either side."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], r.get("Queue_Id", "")))
rows.sort()
cuts = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
print(f"{len(rows)} dispatches, {len(cuts)} AdamW launches")
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for si in range(skip, len(cuts) - 1):
    seg = rows[cuts[si] + 1: cuts[si + 1] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy, end, gaps, last = 0, t0, [], seg[0]
    for r in seg:
        if r[0] > end:
            gaps.append((r[0] - end, last[2], r[2]))
            busy += 0
            end = r[0]
        if r[1] > end:
            busy += r[1] - max(end, r[0])
            end = r[1]
            last = r
    wall = t1 - t0
    print(f"step {si}: wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f}, idle {(wall - busy) / 1e6:.3f} ms in {len(gaps)} gaps; queues {len(set(r[3] for r in seg))}; launches {len(seg)}")
    if si == len(cuts) - 2:
        hist = collections.Counter(min(g[0] // 1000, 20) for g in gaps)
        print("  gap histogram (us: count):", dict(sorted(hist.items())))
        for g in sorted(gaps, reverse=True)[:12]:
            print(f"  {g[0] / 1e3:7.1f} us  after {g[1]:45s} before {g[2]}")
