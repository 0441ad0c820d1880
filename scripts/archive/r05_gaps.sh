#!/bin/bash
# idle-gap analysis of the default run under rocprofv3 --kernel-trace: bash scripts/r05_gaps.sh [ENV=VAL ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MAESTRO_WARM_PASSES=0
for kv in "X=1" "$@"; do
  env $kv rocprofv3 --kernel-trace --output-format csv -d $O/ktg -o gt -- python $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-kernel-timing > $O/kt.log 2>&1 || exit 3
  f=$(find $O/ktg -name "*kernel_trace.csv" | head -1)
  echo "=== $kv"; python $R/scripts/trace_gaps.py $f 10 | head -12
  rm -rf $O/ktg
done
