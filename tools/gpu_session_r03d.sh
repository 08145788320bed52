#!/bin/bash
# Round-3 GPU session D: grouped stream-K weight gradients -- gradient tests, determinism, train bench A/B, kernel trace.
set -u
R=$PWD
OUT=$R/gpurun_out/r03d
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_callers.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -12 "$OUT/pytest_gpu.log"
timeout 300 python tools/train_bench.py > "$OUT/train_streamk.txt" 2>&1; echo "train bench rc=$?"; tail -1 "$OUT/train_streamk.txt"
OBJNERF_WGRAD=atomic timeout 300 python tools/train_bench.py > "$OUT/train_atomic.txt" 2>&1; echo "train bench (atomic) rc=$?"; tail -1 "$OUT/train_atomic.txt"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $R/tools/train_bench.py > "$OUT/prof.log" 2>&1; echo "rocprof rc=$?"
cd $R
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) > "$OUT/train_kernel_stats.md" 2>/dev/null; head -24 "$OUT/train_kernel_stats.md" | cut -c1-200
