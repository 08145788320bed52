R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "stages or render or edges or train" > $O/tests_k.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests_k.txt | cut -c1-250
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_bench.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_bench -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.md 2>/dev/null; rm -rf $O/trace_bench; head -10 $O/bench_kernel_stats.md | cut -c1-200
PMC_KERNEL=ray_bias_kernel timeout 900 bash tools/pmc_quick.sh > $O/pmc_ray_bias.txt 2>&1; tail -4 $O/pmc_ray_bias.txt | cut -c1-300
