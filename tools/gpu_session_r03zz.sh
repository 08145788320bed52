#!/bin/bash
# Round-3 closing session after the weight-gradient rewrite: full GPU suite, smoke, training bench + kernel trace, and a short
# headline bench to confirm the rebuilt library renders at the recorded rate (the inference artefacts stay those of r03z)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03zz; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
timeout 200 python tools/train_bench.py > $O/train_bench.txt 2>&1; echo "train rc=$?"; tail -2 $O/train_bench.txt
timeout 200 python bench.py --steps 5 --warmup 2 --cpu-rays 0 --split-bf16-steps 2 --pmc off > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"; tail -1 $O/bench_short.json | cut -c1-300
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_train -o tr -- python $R/tools/train_bench.py > $O/trace_train.log 2>&1; echo "trace rc=$?"
cd $R
db=$(find $O/trace_train -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/trace_train_kernel_stats.md 2>/dev/null
rm -rf $O/trace_train
head -14 $O/trace_train_kernel_stats.md | cut -c1-150
