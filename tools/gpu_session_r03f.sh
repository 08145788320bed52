#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/r03f
mkdir -p "$OUT"
export TMPDIR=/tmp
for k in 32 16; do
OBJNERF_WGRAD_KITERS=$k timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -x -k "voxel_reference_batch or reproducible" > "$OUT/pytest_k$k.log" 2>&1; echo "k=$k pytest rc=$?"
tail -5 "$OUT/pytest_k$k.log"
done
