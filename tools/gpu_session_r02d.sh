#!/bin/bash
# Round-2 GPU session D: new stage / training tests, small-batch table, config-4 kernel trace, training step timing.
set -u
R=$PWD
OUT=$R/gpurun_out/r02d
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_train.py tests/test_gpu_render.py tests/test_gpu_dist.py tests/test_gpu_callers.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -12 "$OUT/pytest_gpu.log"
timeout 300 python tools/small_batch.py "$OUT/small_batch.md" > "$OUT/small_batch.log" 2>&1; echo "small rc=$?"; cat "$OUT/small_batch.log" | tail -9
timeout 300 python tools/train_bench.py > "$OUT/train_bench.log" 2>&1; echo "train rc=$?"; tail -2 "$OUT/train_bench.log"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_c4" -o c4 -- python $R/bench.py --config 4 --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace_c4.log" 2>&1; echo "trace rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_train" -o tr -- python $R/tools/train_bench.py > "$OUT/trace_train.log" 2>&1; echo "trace train rc=$?"
cd $R
for t in c4 train; do DB=$(ls $OUT/trace_$t/*/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" "$OUT/kernel_stats_$t.md" | cut -c1-160 | head -14; done
