#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/r03v
mkdir -p "$OUT"
export TMPDIR=/tmp
for x in 0 1; do
  OBJNERF_WGRAD_XCD=$x timeout 300 python tools/train_bench.py > "$OUT/train_xcd$x.txt" 2>&1; echo "xcd=$x: $(tail -1 $OUT/train_xcd$x.txt)"
done
cd /tmp
OBJNERF_WGRAD_XCD=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $R/tools/train_bench.py > "$OUT/prof.log" 2>&1; echo "rocprof rc=$?"
cd $R
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) 2>/dev/null | head -8 | cut -c1-150
