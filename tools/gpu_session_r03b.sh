#!/bin/bash
# Round-3 GPU session B: f2 shim replay + sharding tests, band replay (8 ranks' shares on one GPU), bench lines.
set -u
R=$PWD
OUT=$R/gpurun_out/r03b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_callers.py tests/test_gpu_dist.py tests/test_gpu_edges.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -15 "$OUT/pytest_gpu.log"
timeout 600 python tools/band_replay.py "$OUT/r03_band_replay.md" > "$OUT/band_replay.log" 2>&1; echo "band replay rc=$?"
tail -8 "$OUT/band_replay.log"
timeout 500 python bench.py > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"; echo "bench c1 rc=$?"
timeout 300 python bench.py --config 4 --pmc off --cpu-rays 0 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
for f in "$OUT"/bench_c*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("value %.3e %s, %.1f ms/step, scaling %s, frac %.3f, traffic %s, cpu %s, psnr %s, b3 %s" % (
        d["value"], d["unit"], d["ms_per_step"], d["scaling"], r.get("frac", -1), r.get("traffic"), (d.get("cpu_baseline") or {}),
        d.get("psnr_vs_cpu_oracle_db"), {k: v for k, v in (d.get("split_bf16_mode") or {}).items() if k in ("value", "roofline", "error")}))
except Exception as e:
    print("unparsable:", e)
PY
done
tail -3 "$OUT/bench_c1.err"
