#!/bin/bash
# round 3, session i: wgrad issue-priority experiment (convoy hypothesis)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
for prio in 0 1 2 0 1; do
  OBJNERF_WGRAD_PRIO=$prio timeout 200 python tools/train_bench.py > $O/train_bench_p$prio.txt 2>&1; echo "prio $prio: $(tail -1 $O/train_bench_p$prio.txt | cut -c1-120)"
done
export TMPDIR=/tmp; cd /tmp
OBJNERF_WGRAD_PRIO=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_train -o tr -- python $R/tools/train_bench.py > $O/trace_train.log 2>&1; echo "trace rc=$?"
cd $R
db=$(find $O/trace_train -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/train_kernel_stats_p1.md 2>/dev/null
head -6 $O/train_kernel_stats_p1.md | cut -c1-160
