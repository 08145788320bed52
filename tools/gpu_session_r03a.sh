#!/bin/bash
# Round-3 GPU session A: full GPU test suite (both arithmetic modes, incl. the image-scale and checkpoint tests),
# image-scale parity table, smoke.
set -u
R=$PWD
OUT=$R/gpurun_out/r03a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python tools/frame_parity.py "$OUT/r03_frame_parity.md" > "$OUT/frame_parity.log" 2>&1; echo "frame parity rc=$?"
tail -12 "$OUT/frame_parity.log"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -30 "$OUT/pytest_gpu.log"
timeout 200 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
tail -2 "$OUT/smoke.log"
