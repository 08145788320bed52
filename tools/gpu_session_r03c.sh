#!/bin/bash
# Round-3 GPU session C: compositing in the MLP epilogue -- stage tests (restructured K3), bit-equality vs the two-kernel form,
# parity suites, bench A/B (fused vs separate), kernel trace.
set -u
R=$PWD
OUT=$R/gpurun_out/r03c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_render.py tests/test_gpu_frames.py tests/test_gpu_callers.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -15 "$OUT/pytest_gpu.log"
timeout 300 python bench.py --pmc off --cpu-rays 0 --steps 5 > "$OUT/bench_fused.json" 2> "$OUT/bench_fused.err"; echo "bench fused rc=$?"
OBJNERF_COMPOSITE=separate timeout 300 python bench.py --pmc off --cpu-rays 0 --steps 5 > "$OUT/bench_separate.json" 2> "$OUT/bench_separate.err"; echo "bench separate rc=$?"
for f in "$OUT"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("value %.4e, %.2f ms/step, frac %.4f, avg launch %.2f ms, b3 %s" % (d["value"], d["ms_per_step"], r.get("frac", -1), r.get("avg_launch_ms", -1),
          {k: v for k, v in (d.get("split_bf16_mode") or {}).items() if k in ("value",)}))
except Exception as e:
    print("unparsable:", e)
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $R/bench.py --pmc off --cpu-rays 0 --steps 3 --split-bf16-steps 0 > "$OUT/prof.log" 2>&1; echo "rocprof rc=$?"
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -15 "$f"
