#!/bin/bash
# same-box A/B of the attention kernels: ab_old/$1.so (baseline) vs the shipped library; then the attention parity tests
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
base=${1:-head}
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3
for r in 1 2; do
  echo "--- base ($base) round $r"; python scripts/ab_lib.py ab_old/$base.so scripts/bench_attn.py 2>/dev/null | cut -c1-110
  echo "--- new round $r"; python scripts/bench_attn.py 2>/dev/null | cut -c1-110
done
