#!/bin/bash
# round 3, session j: timing ablations of the full-tile wgrad kernel (what bounds it)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${SESSION:-r03j}; mkdir -p $O
export TMPDIR=/tmp
for v in ${VARIANTS:-base NOSTEP NOMFMA NOFRAG}; do
  lib=$R/object_nerf_amd/tune/libobjnerf_wg_$v.so; [ $v = base ] && lib=$R/object_nerf_amd/libobjnerf_hip.so
  cd /tmp
  OBJNERF_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o tr -- python $R/tools/train_bench.py > $O/trace_$v.log 2>&1; echo "$v trace rc=$?"
  cd $R
  db=$(find $O/trace_$v -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/stats_$v.md 2>/dev/null
  grep "wgrad_units_kernel<false>" $O/stats_$v.md | cut -c1-140
  tail -1 $O/trace_$v.log | cut -c1-110
  rm -rf $O/trace_$v
done
