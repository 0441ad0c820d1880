#!/bin/bash
# A/B on one box: fully persistent ping-pong workgroups (default build) vs at most N tiles per workgroup (rebuilt on the box), and
# the round-2 dispatch (MH_GEMM_PP=0), on the pretrain / probe / finetune steps.   usage: bash scripts/ab_pp_limit.sh 3
set -o pipefail
N=${1:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
o=$R/gpurun_out/ab4; mkdir -p $o
c="--steps 30 --warmup 5 --cpu-seconds 0 --no-kernel-timing"
run() { name=$1; shift; timeout -k 10 200 "$@" > $o/$name.json 2>> $o/err.log || exit 1; python -c "import json;d=json.load(open('$o/$name.json'));print('$name',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
run pre_full python $R/bench.py $c
run probe_full python $R/bench.py --phase probe $c
run ft_full python $R/bench.py --phase finetune $c
MH_GEMM_PP=0 run pre_pp0 python $R/bench.py $c
MH_GEMM_PP=0 run probe_pp0 python $R/bench.py --phase probe $c
cd $R && MH_BUILD_FLAGS="-DMH_PP_TILES_PER_WG=$N" python -m maestro_amd.csrc.build > $o/build.log 2>&1 || exit 2
export MH_BUILD_FLAGS="-DMH_PP_TILES_PER_WG=$N"
run pre_lim python $R/bench.py $c
run probe_lim python $R/bench.py --phase probe $c
run ft_lim python $R/bench.py --phase finetune $c
run pre_lim2 python $R/bench.py $c
run probe_lim2 python $R/bench.py --phase probe $c
