#!/bin/bash
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
python scripts/r05_gap5.py 2>/dev/null | tee $O/gap5_rot.txt
for v in 0 1 0 1; do
  MH_GEMM_ROT=$v python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --shapes > $O/rot$v.json 2> $O/rot${v}_shapes.txt || exit 1
  python - $O/rot$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernel_times_ms_per_step"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["step_ms"]["median"], "gemm ms:", round(sum(v for n,v in k.items() if "gemm" in n),3), "loss", d["config"]["final_loss"])
PY
done
