#!/bin/bash
# kernel-trace stats of the default bench (two-stream graph replay and single-stream), per-step normalised summary
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MAESTRO_WARM_PASSES=0
common="--cpu-seconds 0 --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks1 -o r05_single_stream -- python $R/bench.py --steps 10 --warmup 3 --single-stream $common > $out/ks1.log 2>&1 || exit 3
find $out/ks1 -name "*kernel_stats.csv" -exec cp {} $out/ \;
rm -rf $out/ks1
python - $out/r05_single_stream_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    ms=float(r["TotalDurationNs"])/13/1e6; tot+=ms
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:60]
    if ms>=0.02: print(f"{ms:7.3f} ms/step  {int(r['Calls'])/13:7.1f} calls/step  {float(r['AverageNs'])/1e3:8.1f} us  {n}")
print("total", round(tot,3))
PY
