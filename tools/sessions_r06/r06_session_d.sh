R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R; export TMPDIR=/tmp
python tools/input_digests.py > $O/digests.json 2>/dev/null; cat $O/digests.json
timeout 900 python -m pytest tests -x -q -m gpu -k "stages or render or edges" > $O/tests_k.txt 2>&1; echo "tests rc=$?"; tail -12 $O/tests_k.txt | cut -c1-300
timeout 900 python tools/frame_parity.py --coarse $O/coarse_variants.md "shipped=:1" > $O/coarse.log 2>&1; echo "coarse rc=$?"; tail -6 $O/coarse.log | cut -c1-200
timeout 900 python tools/frame_parity.py $O/frame_parity.md > $O/frame_parity.log 2>&1; echo "frame parity rc=$?"; tail -6 $O/frame_parity.log | cut -c1-200
timeout 600 python tools/small_batch.py $O/small_batch.md > $O/small_batch.log 2>&1; echo "small rc=$?"; tail -8 $O/small_batch.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o tr -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 > $O/trace_bench.log 2>&1); echo "trace rc=$?"
db=$(find $O/trace_bench -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/bench_kernel_stats.md 2>/dev/null; rm -rf $O/trace_bench; head -16 $O/bench_kernel_stats.md | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed" $O/gpu_tests.txt | tail -3; tail -30 $O/gpu_tests.txt | cut -c1-250
