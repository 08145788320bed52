R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06aj; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_arch.py -q -m gpu > $O/tests_arch.txt 2>&1; echo "arch tests rc=$?"; tail -25 $O/tests_arch.txt | cut -c1-300
timeout 600 python tools/arch_bench.py $O/arch_bench.md > $O/arch_bench.log 2>&1; echo "arch rc=$?"; tail -6 $O/arch_bench.log | cut -c1-250
