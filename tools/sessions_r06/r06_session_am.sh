R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06am; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python tools/arch_bench.py $O/arch_bench.md > $O/arch_bench.log 2>&1; echo "arch rc=$?"; tail -11 $O/arch_bench.log | cut -c1-200
OBJNERF_GENERIC_CHAIN=0 timeout 600 python tools/arch_bench.py $O/arch_bench_gemm.md > $O/arch_bench_gemm.log 2>&1; echo "arch(gemm) rc=$?"; tail -5 $O/arch_bench_gemm.log | cut -c1-200
