#!/bin/bash
# round 3, session m: spread-staging wgrad: cached-operand ablations + XCD map A/B (kernel time from a trace each)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  cd /tmp
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o tr -- python $R/tools/train_bench.py > $O/trace_$tag.log 2>&1; echo "$tag trace rc=$?"
  cd $R
  db=$(find $O/trace_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/stats_$tag.md 2>/dev/null
  grep "wgrad_units_kernel<false>" $O/stats_$tag.md | cut -c1-130
  rm -rf $O/trace_$tag
}
run base X=1
run xcd OBJNERF_WGRAD_XCD=1
run A32 OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_wg_A32.so
run A1 OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_wg_A1.so
run A36 OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_wg_A36.so
run k128 OBJNERF_WGRAD_KITERS=128
run k32 OBJNERF_WGRAD_KITERS=32
