#!/bin/bash
# bash scripts/r05_ab_env.sh "VAR=VAL [VAR2=VAL2]" ... : default vs each setting, two alternating rounds, 40 steps each
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-kernel-timing --cpu-seconds 0"
one() { env $1 $B 2>>$O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['config']['final_loss'])"; }
for r in 1 2; do
  one X=1
  for kv in "$@"; do one "$kv"; done
done
