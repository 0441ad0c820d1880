#!/bin/bash
# What clock and power does the card hold during the step?  Samples rocm-smi beside a running bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/clock_probe.txt
: > $out
python $R/bench.py --steps 1500 --warmup 5 --cpu-seconds 0 --no-kernel-timing > $R/gpurun_out/clock_bench.json 2>/dev/null &
pid=$!
sleep 14
for i in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy|use" >> $out
  echo "--" >> $out
  sleep 1.5
done
wait $pid
tail -c 400 $R/gpurun_out/clock_bench.json >> $out
cat $out | head -80
