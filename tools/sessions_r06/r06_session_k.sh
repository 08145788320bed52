R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "stages or render or edges" > $O/tests_k.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests_k.txt | cut -c1-250
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_bench.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_bench -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.md 2>/dev/null; rm -rf $O/trace_bench; head -8 $O/bench_kernel_stats.md | cut -c1-200
(cd /tmp && TRAIN_BENCH_FREE_STEPS=24 timeout 600 rocprofv3 --kernel-trace -d $O/trace_train -o tr -- python $R/tools/train_bench.py > $O/trace_train.log 2>&1); echo "train trace rc=$?"; tail -2 $O/trace_train.log | cut -c1-250
db=$(find $O/trace_train -name "*.db" | head -1); [ -n "$db" ] && { python tools/rocpd_stats.py "$db" > $O/train_kernel_stats.md 2>/dev/null; python tools/rocpd_timeline.py "$db" "mlp_kernel" 2 -3 > $O/train_timeline.md; }; rm -rf $O/trace_train; tail -2 $O/train_timeline.md
