#!/bin/bash
# Round 6, VERDICT r05 item 3: XCD-partitioned group streams (MAESTRO_CU_MASK) against the shared-chip group streams; eager launches
# (MAESTRO_GRAPHS=0: a replayed graph does not keep a stream's CU mask, and round 5 measured eager = replay), alternating twice.
cd ${GRAFT_REPO_ROOT:-.}
python scripts/xcc_probe.py
for rep in 1 2; do
  for cfg in "c3 none" "c3 6,2" "c3 5,3" "c4 none" "c4 4,2,2" "c4 3,3,2"; do
    set -- $cfg
    if [ "$2" = none ]; then unset MAESTRO_CU_MASK; else export MAESTRO_CU_MASK=$2; fi
    MAESTRO_GRAPHS=0 python bench.py --config $1 --steps 20 --warmup 4 --cpu-seconds 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'mask=$2', d['value'], 'tiles/s', d['ms_per_step'], 'ms median', d['step_ms']['median'])"
  done
done
