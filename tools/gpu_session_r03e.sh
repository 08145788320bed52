#!/bin/bash
# Round-3 GPU session E: slice-length sweep of the grouped weight-gradient pass.
set -u
R=$PWD
OUT=$R/gpurun_out/r03e
mkdir -p "$OUT"
export TMPDIR=/tmp
for k in 32 64 128; do
  OBJNERF_WGRAD_KITERS=$k timeout 300 python tools/train_bench.py > "$OUT/train_k$k.txt" 2>&1; echo "k=$k: $(tail -1 $OUT/train_k$k.txt)"
done
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -x -k "reproducible or layerwise or voxel_train" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -3 "$OUT/pytest_gpu.log"
