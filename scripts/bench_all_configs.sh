#!/bin/bash
# Every BASELINE configuration + the supervised phases with the final code, one bench line each (GPU box; no CPU leg).
# usage: bash scripts/bench_all_configs.sh r03   -> gpurun_out/cfg/<tag>_bench_<name>.json
set -o pipefail
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/cfg
mkdir -p $out
common="--steps 20 --warmup 5 --cpu-seconds 0"
for cfg in c2 c3p c4 c5; do
  timeout -k 10 240 python $R/bench.py --config $cfg $common > $out/${tag}_bench_${cfg}.json 2>> $out/err.log || exit 1
  python -c "import json;d=json.load(open('$out/${tag}_bench_${cfg}.json'));print('$cfg',d['value'],d['ms_per_step'],d['whole_step']['mfma_frac'],d['roofline']['kernel'],d['roofline']['achieved'])"
done
for phase in finetune probe; do
  timeout -k 10 240 python $R/bench.py --config c3 --phase $phase $common > $out/${tag}_bench_c3_${phase}.json 2>> $out/err.log || exit 2
  python -c "import json;d=json.load(open('$out/${tag}_bench_c3_${phase}.json'));print('$phase',d['value'],d['ms_per_step'],d['whole_step']['mfma_frac'])"
done
timeout -k 10 240 python $R/bench.py --config c4 --phase finetune $common > $out/${tag}_bench_c4_finetune.json 2>> $out/err.log || exit 3
python -c "import json;d=json.load(open('$out/${tag}_bench_c4_finetune.json'));print('c4 finetune',d['value'],d['ms_per_step'])"
timeout -k 10 240 python $R/bench.py --batch 64 $common > $out/${tag}_bench_c3_b64.json 2>> $out/err.log || exit 4
python -c "import json;d=json.load(open('$out/${tag}_bench_c3_b64.json'));print('c3 B=64',d['value'],d['ms_per_step'],d['whole_step']['mfma_frac'])"
