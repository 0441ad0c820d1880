#!/bin/bash
# bash scripts/r05_ab_kernels.sh <ab_old name> ... : per-kernel times of the default bench (single-stream eager leg) under other library builds
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
B="bench.py --steps 20 --warmup 5 --cpu-seconds 0"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_times_ms_per_step']; h=d['hbm_substages']
print('$1', d['ms_per_step'], ' '.join(f'{n}={v}' for n,v in k.items() if 'grouped' in n or 'pp' in n), ' '.join(f\"{n}={v['avg_us']}us/{v['gb_s']:.0f}\" for n,v in h.items()))
"; }
for r in 1 2; do
  for n in "$@"; do python scripts/ab_lib.py ab_old/$n.so $B 2>>$O/ab.err | show $n; done
  python $B 2>>$O/ab.err | show new
done
