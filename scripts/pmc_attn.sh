#!/bin/bash
# SQ counters of the attention kernels on the decoder shape of C3 (B 32, N 1024, 16 heads x 32) and the encoder shape (N 256, 12 x 64)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_attn
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shape in "32 1024 16 32" "32 256 12 64"; do
  tag=$(echo $shape | tr ' ' '_')
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/tmp_${tag}_$i -o a -- python3 $R/scripts/attn_one.py $shape > $out/log_${tag}_$i.txt 2>&1 || exit 1
    find $out/tmp_${tag}_$i -name "*counter_collection.csv" -exec cp {} $out/${tag}_$i.csv \;
    rm -rf $out/tmp_${tag}_$i
  done
done
ls $out
