#!/bin/bash
# Round-2 GPU session C: full GPU suite after the single-enqueue render_rays_multi; bench lines of configs 1 and 4.
set -u
R=$PWD
OUT=$R/gpurun_out/r02c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -25 "$OUT/pytest_gpu.log"
timeout 400 python bench.py --pmc off > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"; echo "bench c1 rc=$?"
timeout 400 python bench.py --config 4 --pmc off > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
for f in "$OUT"/bench_c*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("value %.4e %s, %.1f ms/step, frac %.3f, mlp frac of step %s, launches %s avg %.2f ms, b3 %s" % (
        d["value"], d["unit"], d["ms_per_step"], r.get("frac", -1), r.get("mlp_time_frac_of_step"), r.get("launches"), r.get("avg_launch_ms", 0),
        {k: v for k, v in (d.get("split_bf16_mode") or {}).items() if k in ("value", "ms_per_step", "error")}))
except Exception as e:
    print("unparsable:", e)
PY
done
