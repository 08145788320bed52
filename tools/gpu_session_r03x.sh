#!/bin/bash
# quick check after the transposed ray_bias kernel: stage + render + frames + multi tests, headline bench, kernel trace
set -u
R=$PWD
OUT=$R/gpurun_out/r03x
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_render.py tests/test_gpu_frames.py tests/test_gpu_checkpoint.py tests/test_gpu_edges.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1
timeout 300 python bench.py --pmc off --cpu-rays 0 --steps 5 --split-bf16-steps 3 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
cd $R
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/trace_kernel_stats.md" 2>/dev/null && python tools/hbm_rates.py "$db" > "$OUT/hbm_rates.md" 2>&1
head -9 "$OUT/trace_kernel_stats.md" | cut -c1-170; cat "$OUT/hbm_rates.md"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.4e, %.2f ms/step, frac %.4f, b3 %s" % (d["value"], d["ms_per_step"], r["frac"], d["split_bf16_mode"]["value"]))
PY
