#!/bin/bash
# Whole-step A/B of SEVERAL builds of the library on one box: bash scripts/ab_libs_many.sh "ab_old/a.so ab_old/b.so" [bench args]
# (two rounds; each round = the shipped library, then every variant in order)
set -o pipefail
libs=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
o=$R/gpurun_out/ablib; mkdir -p $o
c="--steps 30 --warmup 5 --cpu-seconds 0 --no-kernel-timing $*"
show() { python -c "import json;d=json.loads(open('$o/$1.json').read().strip().splitlines()[-1]);print('$1',d['value'],d['ms_per_step'],d['step_ms']['median'],d['config']['final_loss'])"; }
for r in ${ROUNDS:-a b}; do
  timeout -k 10 200 python $R/bench.py $c > $o/base_$r.json 2>> $o/err.log || exit 1; show base_$r
  for lib in $libs; do
    n=$(basename $lib .so)
    timeout -k 10 200 python $R/scripts/ab_lib.py $R/$lib $R/bench.py $c > $o/${n}_$r.json 2>> $o/err.log || exit 1; show ${n}_$r
  done
done
