#!/bin/bash
# Round-2 GPU session B: GPU tests after the float64-accumulated sampler, parity table, kernel trace of the headline bench.
set -u
R=$PWD
OUT=$R/gpurun_out/r02b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_dist.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -15 "$OUT/pytest_gpu.log"
timeout 300 python tools/parity_report.py "$OUT/r02_parity.md" > "$OUT/parity.log" 2>&1; echo "parity rc=$?"
tail -28 "$OUT/parity.log"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
cd $R
DB=$(ls $OUT/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" "$OUT/kernel_stats.md" | head -8
