#!/bin/bash
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
for v in 0 3 0 3; do
  MH_GEMM_ROT=1 MAESTRO_TOUCH_W=$v python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --shapes > $O/rt$v.json 2> $O/rt${v}_shapes.txt || exit 1
  python - $O/rt$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_times_ms_per_step"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["step_ms"]["median"], "gemm ms:", round(sum(v for n,v in k.items() if "gemm" in n),3), "loss", d["config"]["final_loss"])
PY
done
