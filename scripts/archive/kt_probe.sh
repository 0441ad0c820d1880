set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
o=$R/gpurun_out/kt5; mkdir -p $o
c="--steps 20 --warmup 5 --cpu-seconds 0 --phase ${PHASE:-probe} --shapes"
MH_GEMM_PP=1 timeout -k 10 200 python $R/bench.py $c > $o/probe_pp1.json 2> $o/probe_pp1.shapes || exit 1
MH_GEMM_PP=0 timeout -k 10 200 python $R/bench.py $c > $o/probe_pp0.json 2> $o/probe_pp0.shapes || exit 1
for k in pp1 pp0; do python -c "import json;d=json.load(open('$o/probe_$k.json'));print('$k',d['value'],d['ms_per_step'],d['kernel_times_ms_per_step'])"; done
echo PP1; grep "ms/step" $o/probe_pp1.shapes | head -24
echo PP0; grep "ms/step" $o/probe_pp0.shapes | head -24
