#!/bin/bash
# Round-3 GPU session G: hoisted per-ray terms -- stage test, parity suites, bench A/B (hoist on / off).
set -u
R=$PWD
OUT=$R/gpurun_out/r03g
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_render.py tests/test_gpu_frames.py tests/test_gpu_callers.py tests/test_gpu_checkpoint.py tests/test_gpu_edges.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -12 "$OUT/pytest_gpu.log"
timeout 300 python bench.py --pmc off --cpu-rays 0 --steps 5 --split-bf16-steps 3 > "$OUT/bench_hoist.json" 2> "$OUT/bench_hoist.err"; echo "bench hoist rc=$?"
OBJNERF_HOIST=0 timeout 300 python bench.py --pmc off --cpu-rays 0 --steps 5 --split-bf16-steps 3 > "$OUT/bench_nohoist.json" 2> "$OUT/bench_nohoist.err"; echo "bench no-hoist rc=$?"
timeout 300 python bench.py --config 4 --pmc off --cpu-rays 0 --steps 5 --split-bf16-steps 0 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
for f in "$OUT"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("value %.4e, %.2f ms/step, frac %.4f, avg launch %.2f ms, b3 %s" % (d["value"], d["ms_per_step"], r.get("frac", -1), r.get("avg_launch_ms", -1), (d.get("split_bf16_mode") or {}).get("value")))
except Exception as e:
    print("unparsable:", e)
PY
done
