#!/bin/bash
# Round-3 refresh after the ray_bias tweak: headline line (live PMC), kernel trace, band replay (medians), traffic attribution.
set -u
R=$PWD
OUT=$R/gpurun_out/r03y
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
for h in 1 0; do
  OBJNERF_HOIST=$h timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_hoist$h" -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/fetch_hoist$h.log" 2>&1; echo "fetch pass hoist=$h rc=$?"
done
cd $R
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > "$OUT/trace_kernel_stats.md" 2>/dev/null && python tools/hbm_rates.py "$db" > "$OUT/hbm_rates.md" 2>&1
python - <<'PY'
import csv, glob, os
for h in (1, 0):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(os.environ.get("OUT_DIR", "gpurun_out/r03y"), "fetch_hoist%d" % h, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mlp_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
                tot += float(row["Counter_Value"]); n += 1
    print("hoist=%d: FETCH_SIZE raw %.1f MB per launch over %d launches (x2 corrected: %.1f MB)" % (h, tot * 1024 / max(n, 1) / 1e6, n, 2 * tot * 1024 / max(n, 1) / 1e6))
PY
timeout 600 python tools/band_replay.py "$OUT/r03_band_replay.md" > "$OUT/band_replay.log" 2>&1; echo "band replay rc=$?"; tail -6 "$OUT/band_replay.log"
head -8 "$OUT/trace_kernel_stats.md" | cut -c1-160; cat "$OUT/hbm_rates.md"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.4e, %.2f ms/step, frac %.4f, traffic %s, cpu %s, psnr %s, b3 %s" % (d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), d["cpu_baseline"]["value"], d.get("psnr_vs_cpu_oracle_db"), d["split_bf16_mode"]["value"]))
PY
