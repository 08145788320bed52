#!/bin/bash
# Collects PMC counters for the bench workload, one rocprofv3 pass per counter group
# (no trace domains combined with --pmc; see task notes).  Usage: tools/pmc_run.sh <outdir>
set -u
R=$PWD
OUT=${1:-gpurun_out/pmc_r01}
case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -o pmc -- \
      python $R/bench.py --steps 1 --warmup 1 --cpu-rays 0 --train-steps 0 --pmc off > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
cd $R
find "$OUT" -name "*counter_collection.csv" | head
