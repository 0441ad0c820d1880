#!/bin/bash
# SQ counters of the fc1 GEMM (8192 x 3072 x 768, 128x128 register-staged kernel) with a plain epilogue and with its real one
# (bias + GELU + byte-coded derivative): separate --pmc passes, kernel-trace only.  -> gpurun_out/pmc_fc1/{plain,gelu}_*.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_fc1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for epi in plain gelu; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/tmp_${epi}_$i -o ${epi}_$i -- python3 $R/scripts/gemm_one.py 0 8192 3072 768 v1 $epi > $out/log_${epi}_$i.txt 2>&1 || exit 1
    find $out/tmp_${epi}_$i -name "*counter_collection.csv" -exec cp {} $out/${epi}_$i.csv \;
    rm -rf $out/tmp_${epi}_$i
  done
done
ls $out
