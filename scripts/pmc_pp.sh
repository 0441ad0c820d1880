#!/bin/bash
# SQ counters of the fc1 GEMM (8192 x 3072 x 768) on the persistent ping-pong tile (gemm_pp.hip) against the one-tile-per-workgroup
# kernel: separate --pmc passes, kernel-trace only.  -> gpurun_out/pmc_pp/<variant>_<pass>.csv ; scripts/summarize_pmc_pp.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_pp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "reg_gelu tile0 gelu" "pp_gelu tile7 gelu" "pp_main tile8 gelu" "pp_plain tile7 plain" "reg_plain tile0 plain"; do
  set -- $v; name=$1; impl=$2; epi=$3
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d $out/tmp_${name}_$i -o ${name}_$i -- python3 $R/scripts/gemm_one.py 0 8192 3072 768 $impl $epi > $out/log_${name}_$i.txt 2>&1 || { tail -3 $out/log_${name}_$i.txt; continue; }
    find $out/tmp_${name}_$i -name "*counter_collection.csv" -exec cp {} $out/${name}_$i.csv \;
    rm -rf $out/tmp_${name}_$i
  done
done
ls $out | head -40
