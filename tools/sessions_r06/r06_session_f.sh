R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/gpu_tests.txt | tail -12 | cut -c1-250
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_bench.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_bench -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.md 2>/dev/null; rm -rf $O/trace_bench; head -10 $O/bench_kernel_stats.md | cut -c1-200
for rep in 1 2; do for nodes in 1 2; do echo "NODES=$nodes"; OBJNERF_TRAIN_NODES=$nodes timeout 300 python tools/train_bench.py 2>&1 | tail -2 | cut -c1-250; done; done | tee $O/train_nodes_ab.txt
timeout 600 python tools/small_batch.py $O/small_batch.md > $O/small_batch.log 2>&1; tail -3 $O/small_batch.log
