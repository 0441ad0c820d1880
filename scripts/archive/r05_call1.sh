#!/bin/bash
# round 5, first GPU call: baseline x2 on one box + stream-priority A/B (eager and graph replay)
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-kernel-timing --cpu-seconds 0"
run() { name=$1; shift; echo "== $name" | tee -a $O/call1.log; env "$@" $B > $O/$name.json 2>> $O/call1.err; python - "$O/$name.json" <<'PY' | tee -a gpurun_out/r05/call1.log
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["step_ms"]["median"], d["step_ms"]["min"])
PY
}
python bench.py --steps 20 --warmup 5 --shapes > $O/base_full.json 2> $O/base_shapes.txt || exit 1
python - <<'PY' | tee -a gpurun_out/r05/call1.log
import json; d=json.load(open("gpurun_out/r05/base_full.json")); print("full", d["value"], d["ms_per_step"], d["step_ms"]["median"], d["roofline"]["frac"])
PY
run base1 X=1 || exit 1
run prio_main1 MAESTRO_MAIN_PRIORITY=-1 || exit 1
run base2 X=1 || exit 1
run prio_main2 MAESTRO_MAIN_PRIORITY=-1 || exit 1
run eager_base MAESTRO_GRAPHS=0 || exit 1
run eager_prio MAESTRO_GRAPHS=0 MAESTRO_MAIN_PRIORITY=-1 || exit 1
run eager_base2 MAESTRO_GRAPHS=0 || exit 1
run eager_prio2 MAESTRO_GRAPHS=0 MAESTRO_MAIN_PRIORITY=-1 || exit 1
python -c "import torch; print(torch.cuda.Stream(priority=-1).priority, torch.cuda.Stream(priority=0).priority, torch.cuda.Stream(priority=1).priority)" | tee -a $O/call1.log
