#!/bin/bash
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_mae_gpu.py -x -q -k "optimizer_under or adamw" 2>&1 | tail -4
for v in 1 0 1 0; do
  MAESTRO_OPT_UNDER_WGRAD=$v python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-kernel-timing > $O/optw$v.json 2>> $O/optw.err || exit 1
  python - $O/optw$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"], d["step_ms"]["median"], d["step_ms"]["min"], "loss", d["config"]["final_loss"])
PY
done
