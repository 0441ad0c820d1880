#!/bin/bash
# Same-box A/B of the MH_TILE_AUTO rule (default) against the round-2 dispatch (MH_GEMM_PP=0) on the pretrain / probe / finetune steps.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
o=$R/gpurun_out/ab3; mkdir -p $o
c="--steps 30 --warmup 5 --cpu-seconds 0 --no-kernel-timing"
run() { name=$1; shift; timeout -k 10 200 "$@" > $o/$name.json 2>> $o/err.log || exit 1; python -c "import json;d=json.load(open('$o/$name.json'));print('$name',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for r in a b; do
  MH_GEMM_PP=1 run pre_pp1_$r python $R/bench.py $c
  MH_GEMM_PP=0 run pre_pp0_$r python $R/bench.py $c
  MH_GEMM_PP=1 run probe_pp1_$r python $R/bench.py --phase probe $c
  MH_GEMM_PP=0 run probe_pp0_$r python $R/bench.py --phase probe $c
  MH_GEMM_PP=1 run ft_pp1_$r python $R/bench.py --phase finetune $c
  MH_GEMM_PP=0 run ft_pp0_$r python $R/bench.py --phase finetune $c
done
