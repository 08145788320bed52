#!/bin/bash
# One GPU-box session: bench line, rocprofv3 kernel trace of the same command, PMC passes.
# Usage (via gpurun): bash tools/profile_round.sh <tag>      outputs under gpurun_out/<tag>/
set -u
R=$PWD
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 python $R/bench.py --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
cd $R
bash tools/pmc_run.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1; echo "pmc rc=$?"
tail -c 600 "$OUT/bench.json"
