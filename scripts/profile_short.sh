#!/bin/bash
# Short form of profile_round.sh for a tight GPU budget: default bench line + the two kernel-trace summaries (no PMC passes).
# usage: bash scripts/profile_short.sh r03b   -> gpurun_out/prof/<tag>_*
set -o pipefail
tag=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $out/${tag}_bench_c3_n1.json 2> $out/${tag}_bench.err || exit 1
tail -c 300 $out/${tag}_bench_c3_n1.json; echo
common="--cpu-seconds 0 --no-kernel-timing"
export MAESTRO_WARM_PASSES=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ${tag} -- python $R/bench.py --steps 10 --warmup 3 $common > $out/ks.log 2>&1 || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks1 -o ${tag}_single_stream -- python $R/bench.py --steps 10 --warmup 3 --single-stream $common > $out/ks1.log 2>&1 || exit 3
for d in ks ks1; do find $out/$d -name "*.csv" -exec cp {} $out/ \; ; done
rm -rf $out/ks $out/ks1
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
ls -la $out
