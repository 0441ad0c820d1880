#!/bin/bash
# bash scripts/r05_ab_libs.sh <ab_old name> [<ab_old name> ...] : whole-step A/B of the shipped library against ab_old/<name>.so builds (two alternating rounds)
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
B="bench.py --steps 40 --warmup 5 --no-kernel-timing --cpu-seconds 0"
one() { "$@" 2>>$O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['config']['final_loss'])"; }
for r in 1 2 3; do
  for n in "$@"; do echo -n "$n: "; one python scripts/ab_lib.py ab_old/$n.so $B; done
  echo -n "new: "; one python $B
done
