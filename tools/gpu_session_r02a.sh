#!/bin/bash
# Round-2 GPU session A: full GPU test suite (both arithmetic modes), parity table, bench lines of every config, smoke.
set -u
R=$PWD
OUT=$R/gpurun_out/r02a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest_gpu.log"
timeout 300 python tools/parity_report.py "$OUT/r02_parity.md" > "$OUT/parity.log" 2>&1; echo "parity rc=$?"
tail -25 "$OUT/parity.log"
timeout 400 python bench.py > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"; echo "bench c1 rc=$?"
for c in 0 2 4; do
  timeout 300 python bench.py --config $c --pmc off > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"; echo "bench c$c rc=$?"
done
MASTER_PORT=29540 timeout 300 python bench.py --config 3 --dist --pmc off --cpu-rays 0 > "$OUT/bench_c3_dist.json" 2> "$OUT/bench_c3_dist.err"; echo "bench c3 rc=$?"
timeout 200 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
for f in "$OUT"/bench_c*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("value %.3e %s, %.1f ms/step, frac %.3f, traffic %s, cpu %s, psnr %s, b3 %s" % (
        d["value"], d["unit"], d["ms_per_step"], r.get("frac", -1), r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"),
        d.get("psnr_vs_cpu_oracle_db"), {k: v for k, v in (d.get("split_bf16_mode") or {}).items() if k in ("value", "roofline", "error")}))
except Exception as e:
    print("unparsable:", e)
PY
done
