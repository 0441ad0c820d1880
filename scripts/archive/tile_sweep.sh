#!/bin/bash
# In-step time of every GEMM shape of a step under each forced tile (MH_GEMM_TILE; ineligible problems fall back to the AUTO rule):
# bench.py's eager single-stream leg with HIP events around every launch.  usage: PHASE=pretrain bash scripts/tile_sweep.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
PHASE=${PHASE:-pretrain}
o=$R/gpurun_out/sweep_$PHASE; mkdir -p $o
c="--steps 20 --warmup 5 --cpu-seconds 0 --phase $PHASE --shapes ${EXTRA:-}"
for t in auto 0 1 7 13 14; do
  if [ $t = auto ]; then unset MH_GEMM_TILE; else export MH_GEMM_TILE=$t; fi
  timeout -k 10 200 python $R/bench.py $c > $o/tile_$t.json 2> $o/tile_$t.shapes || exit 1
  python -c "import json;d=json.load(open('$o/tile_$t.json'));print('tile $t',d['value'],d['ms_per_step'])"
done
