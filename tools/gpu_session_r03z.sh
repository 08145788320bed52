#!/bin/bash
# Round-3 final artefacts: full GPU suite, bench lines of every config, kernel trace + PMC passes of the headline, training
# step + trace, 4K frame, parity tables, band replay.  Everything under gpurun_out/r03z/ (copied to profiles/ afterwards).
set -u
R=$PWD
OUT=$R/gpurun_out/r03z
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
for c in 0 2 4; do
  timeout 400 python bench.py --config $c --cpu-rays 0 > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"; echo "bench c$c rc=$?"
done
MASTER_PORT=29540 timeout 300 python bench.py --config 3 --dist --pmc off --cpu-rays 0 > "$OUT/bench_c3_dist.json" 2> "$OUT/bench_c3_dist.err"; echo "bench c3 rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_c4" -o c4 -- python $R/bench.py --config 4 --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace_c4.log" 2>&1; echo "trace c4 rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_train" -o tr -- python $R/tools/train_bench.py > "$OUT/trace_train.log" 2>&1; echo "trace train rc=$?"
cd $R
for t in trace trace_c4 trace_train; do
  db=$(find $OUT/$t -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/${t}_kernel_stats.md" 2>/dev/null
done
db=$(find $OUT/trace -name "*.db" | head -1); [ -n "$db" ] && python tools/hbm_rates.py "$db" > "$OUT/hbm_rates.md" 2>&1
bash tools/pmc_run.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py "$OUT/pmc" "$OUT/r03_pmc.json" > "$OUT/pmc_summary.log" 2>&1; echo "pmc summary rc=$?"
timeout 300 python tools/train_bench.py > "$OUT/train_bench.txt" 2>&1; echo "train rc=$?"; tail -1 "$OUT/train_bench.txt"
timeout 300 python tools/big_frame.py "$OUT/big_frame.md" > "$OUT/big_frame.log" 2>&1; echo "big frame rc=$?"; tail -2 "$OUT/big_frame.log"
timeout 400 python tools/parity_report.py "$OUT/r03_parity.md" > "$OUT/parity.log" 2>&1; echo "parity rc=$?"; tail -28 "$OUT/parity.log"
timeout 300 python tools/frame_parity.py "$OUT/r03_frame_parity.md" > "$OUT/frame_parity.log" 2>&1; echo "frame parity rc=$?"; tail -9 "$OUT/frame_parity.log"
timeout 600 python tools/band_replay.py "$OUT/r03_band_replay.md" > "$OUT/band_replay.log" 2>&1; echo "band replay rc=$?"; tail -6 "$OUT/band_replay.log"
timeout 300 python tools/small_batch.py "$OUT/small_batch.md" > "$OUT/small_batch.log" 2>&1; echo "small batch rc=$?"
timeout 200 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
for f in "$OUT"/bench*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("value %.4e %s, %.2f ms/step, scaling %s, frac %.4f, traffic %s, cpu %s, psnr %s, b3 %s" % (
        d["value"], d["unit"], d["ms_per_step"], d["scaling"], r.get("frac", -1), r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"),
        d.get("psnr_vs_cpu_oracle_db"), {k: v for k, v in (d.get("split_bf16_mode") or {}).items() if k in ("value",)}))
except Exception as e:
    print("unparsable:", e)
PY
done
