#!/bin/bash
# round 3, session h: wgrad with global (not flat) operand loads + pipelined fragment reads: gradient tests, train bench, trace
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/test_train.txt 2>&1; echo "train tests rc=$?"; tail -3 $O/test_train.txt
timeout 200 python tools/train_bench.py > $O/train_bench.txt 2>&1; echo "bench rc=$?"; tail -2 $O/train_bench.txt
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_train -o tr -- python $R/tools/train_bench.py > $O/trace_train.log 2>&1; echo "trace rc=$?"
cd $R
db=$(find $O/trace_train -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/train_kernel_stats.md 2>/dev/null
head -16 $O/train_kernel_stats.md | cut -c1-200
