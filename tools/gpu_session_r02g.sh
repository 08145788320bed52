#!/bin/bash
# Round-2 final artefacts: headline bench line, kernel trace of the same command, PMC passes, bench lines of the other configs.
set -u
R=$PWD
OUT=$R/gpurun_out/r02g
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 python $R/bench.py > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
cd $R
bash tools/pmc_run.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py "$OUT/pmc" "$OUT/pmc.json" > "$OUT/pmc.md" 2>&1; echo "pmc summary rc=$?"
for c in 0 2 4; do timeout 300 python bench.py --config $c --pmc off > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"; echo "bench c$c rc=$?"; done
MASTER_PORT=29541 timeout 300 python bench.py --config 3 --dist --pmc off --cpu-rays 0 > "$OUT/bench_c3_dist.json" 2> "$OUT/bench_c3_dist.err"; echo "bench c3 rc=$?"
DB=$(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" "$OUT/kernel_stats.md" | cut -c1-150 | head -8
[ -n "$DB" ] && python tools/hbm_rates.py "$DB" > "$OUT/hbm_rates.md" 2>&1; tail -6 "$OUT/hbm_rates.md"
tail -c 400 "$OUT/bench_c1.json"
