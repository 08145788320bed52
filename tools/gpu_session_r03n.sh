#!/bin/bash
# round 3, session n: 256 x 256 weight-gradient tiles (wgrad_big_kernel): gradient tests, A/B against the 128 x 128 kernel, trace
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/test_train.txt 2>&1; echo "train tests rc=$?"; tail -3 $O/test_train.txt
OBJNERF_WGRAD_BIG=0 timeout 200 python tools/train_bench.py > $O/train_bench_big0.txt 2>&1; echo "big=0: $(tail -1 $O/train_bench_big0.txt | cut -c1-110)"
timeout 200 python tools/train_bench.py > $O/train_bench_big1.txt 2>&1; echo "big=1: $(tail -1 $O/train_bench_big1.txt | cut -c1-110)"
OBJNERF_WGRAD_BIG=0 timeout 200 python tools/train_bench.py > $O/train_bench_big0b.txt 2>&1; echo "big=0: $(tail -1 $O/train_bench_big0b.txt | cut -c1-110)"
timeout 200 python tools/train_bench.py > $O/train_bench_big1b.txt 2>&1; echo "big=1: $(tail -1 $O/train_bench_big1b.txt | cut -c1-110)"
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_train -o tr -- python $R/tools/train_bench.py > $O/trace_train.log 2>&1; echo "trace rc=$?"
cd $R
db=$(find $O/trace_train -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/train_kernel_stats.md 2>/dev/null
rm -rf $O/trace_train
head -12 $O/train_kernel_stats.md | cut -c1-150
