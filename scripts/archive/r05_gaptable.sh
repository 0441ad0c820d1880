#!/bin/bash
# the four columns of profiles/r05_gap_table.md on ONE box: isolated hot / cold (scripts/r05_gap.py), in-step eager single stream
# (bench.py --shapes), in-step two-stream graph replay (rocprofv3 kernel trace, per (kernel, grid) average)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/scripts/r05_gap.py > $O/gt_isolated.txt 2>/dev/null || exit 1
python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --shapes > $O/gt_bench.json 2> $O/gt_shapes.txt || exit 2
export MAESTRO_WARM_PASSES=0
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o gt -- python $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-kernel-timing > $O/kt.log 2>&1 || exit 3
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_breakdown.py $f 10 0.5 > $O/gt_two_stream.txt
rm -rf $O/kt
head -5 $O/gt_two_stream.txt
