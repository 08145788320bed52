R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python tools/callers_train_diag.py > $O/callers_diag.txt 2>&1; echo "diag rc=$?"; grep -v "^Voxel\|^Filling" $O/callers_diag.txt | tail -40 | cut -c1-220
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed\|^FAILED" $O/gpu_tests.txt | tail -12 | cut -c1-250
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_bench.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_bench -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.md 2>/dev/null; rm -rf $O/trace_bench; head -10 $O/bench_kernel_stats.md | cut -c1-200
OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_coarsent.so python tools/small_batch.py 2>/dev/null | tail -2
(cd /tmp && OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_coarsent.so timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_nt -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_nt.log 2>&1); db=$(find $O/trace_nt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 2>/dev/null | grep "sample_coarse" | cut -c1-160; rm -rf $O/trace_nt
