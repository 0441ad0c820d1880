#!/bin/bash
# bash scripts/r05_ab_lib.sh <ab_old name> : whole-step A/B of the shipped library against ab_old/<name>.so (two alternating rounds) + GEMM tests
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
B="bench.py --steps 40 --warmup 5 --no-kernel-timing --cpu-seconds 0"
one() { "$@" 2>>$O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['config']['final_loss'])"; }
for r in 1 2; do
  echo -n "old: "; one python scripts/ab_lib.py ab_old/$1.so $B
  echo -n "new: "; one python $B
done
