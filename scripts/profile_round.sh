#!/bin/bash
# Runs on the GPU box (gpurun): default bench line + rocprofv3 kernel stats (graph/multi-stream and single-stream) +
# separate PMC passes for HBM traffic.  usage: bash scripts/profile_round.sh r01   -> gpurun_out/prof/<tag>_*
set -o pipefail
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $out/${tag}_bench_c3_n1.json 2> $out/${tag}_bench.err || exit 1
tail -c 600 $out/${tag}_bench_c3_n1.json; echo
common="--cpu-seconds 0 --no-kernel-timing"
# the traces are normalised per step (13 = 3 warm-up + 10 timed; counters: 5): no start-up passes of the first step in them
export MAESTRO_WARM_PASSES=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ${tag} -- python $R/bench.py --steps 10 --warmup 3 $common > $out/ks.log 2>&1 || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks1 -o ${tag}_single_stream -- python $R/bench.py --steps 10 --warmup 3 --single-stream $common > $out/ks1.log 2>&1 || exit 3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o ${tag}_fetch -- python $R/bench.py --steps 2 --warmup 3 $common > $out/pf.log 2>&1 || exit 4
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o ${tag}_write -- python $R/bench.py --steps 2 --warmup 3 $common > $out/pw.log 2>&1 || exit 5
for d in ks ks1 pf pw; do find $out/$d -name "*.csv" -exec cp {} $out/ \; ; done
rm -rf $out/ks $out/ks1 $out/pf $out/pw
# the per-dispatch traces are large: keep the stats and the counter collections only
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
ls -la $out
