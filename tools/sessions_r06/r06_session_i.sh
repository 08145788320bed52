R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train.txt 2>&1; echo "train tests rc=$?"; tail -5 $O/tests_train.txt | cut -c1-300
# A/B: embedding gradients folded into the chain (default) vs the two GEMMs (OBJNERF_BWD_DX=0), alternating
bash tools/train_ab.sh r06i ship 3 OBJNERF_BWD_DX=0 2>&1 | tail -8
# kernel trace of the bench (ray-side kernels of this library) + ray_bias counters (left over from session h)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_bench.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_bench -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.md 2>/dev/null; rm -rf $O/trace_bench; head -12 $O/bench_kernel_stats.md | cut -c1-200
