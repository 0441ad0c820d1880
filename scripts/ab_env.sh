#!/bin/bash
# Whole-step A/B of one environment switch on one box: bash scripts/ab_env.sh VAR=VALUE [bench args]   (two alternating rounds)
set -o pipefail
kv=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
o=$R/gpurun_out/abenv; mkdir -p $o
c="--steps 30 --warmup 5 --cpu-seconds 0 --no-kernel-timing $*"
run() { name=$1; shift; timeout -k 10 200 "$@" > $o/$name.json 2>> $o/err.log || exit 1; python -c "import json;d=json.load(open('$o/$name.json'));print('$name',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for r in a b; do
  run base_$r python $R/bench.py $c
  run ${kv%%=*}_$r env $kv python $R/bench.py $c
done
