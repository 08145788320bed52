#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/r03t
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_render.py -m gpu -q -p no:cacheprovider -x -k "hoist or matches_reference or multi" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1
timeout 300 python tools/small_batch.py "$OUT/small_batch.md" > "$OUT/small_batch.log" 2>&1; echo "small batch rc=$?"; tail -7 "$OUT/small_batch.md"
OBJNERF_HOIST=0 timeout 300 python tools/small_batch.py "$OUT/small_batch_nohoist.md" > "$OUT/small_batch2.log" 2>&1; echo "small batch (no hoist) rc=$?"; tail -7 "$OUT/small_batch_nohoist.md"
