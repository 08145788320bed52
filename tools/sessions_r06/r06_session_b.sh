R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python tools/coarse_worst_ray.py $O/coarse_worst_ray.md > $O/worst.log 2>&1; echo "worst rc=$?"; tail -60 $O/worst.log | cut -c1-260
(cd /tmp && SMALL_BATCH_ONLY=1024 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_small -o tr -- python $R/tools/small_batch.py > $O/trace_small.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_small -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/small_1024_kernel_stats.md 2>/dev/null; rm -rf $O/trace_small; head -30 $O/small_1024_kernel_stats.md | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt | cut -c1-300
