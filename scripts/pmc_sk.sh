#!/bin/bash
# SQ counters of one long-K narrow-N GEMM in THROUGHPUT shape (M = 65536, N = 768, K = 3072, NT, plain bf16 output: 8 tiles per CU, so the
# tile count does not matter) on the register-staged 128 x 128 / 192 x 128 tiles (two workgroups per CU) and on the stream-K tiles (one
# four-wave workgroup per CU): separate --pmc passes, kernel-trace only.  -> gpurun_out/pmc_sk/<variant>_<pass>.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_sk
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
LAY=${LAY:-0}
VARIANTS=${VARIANTS:-reg128:tile0 reg192:tile14 sk192:tile15 sk256:tile16}
for v in $VARIANTS; do
  name=${v%%:*}; impl=${v##*:}
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d $out/tmp_${name}_$i -o ${name}_$i -- python3 $R/scripts/gemm_one.py $LAY 65536 768 3072 $impl plain > $out/log_${name}_$i.txt 2>&1 || { tail -3 $out/log_${name}_$i.txt; continue; }
    find $out/tmp_${name}_$i -name "*counter_collection.csv" -exec cp {} $out/${name}_$i.csv \;
    rm -rf $out/tmp_${name}_$i
  done
done
python3 - <<PY
import collections, csv, glob, os
out = "$out"
data = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/*_[0-9].csv")):
    v = os.path.basename(f).rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            data[v][r["Counter_Name"]].append(float(r["Counter_Value"]))
vs = sorted(data)
names = sorted({c for v in vs for c in data[v]})
print(f"{'counter':28s}" + "".join(f"{v:>14s}" for v in vs))
for c in names:
    print(f"{c:28s}" + "".join(f"{(sum(data[v][c]) / len(data[v][c]) if data[v][c] else float('nan')):14.4e}" for v in vs))
PY
