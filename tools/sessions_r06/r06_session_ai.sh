R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06ai; mkdir -p $O; cd $R; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/trace_lw -o tr -- python $R/tools/layerwise_trace.py > $O/trace_lw.log 2>&1); echo "lw trace rc=$?"
db=$(find $O/trace_lw -name "*.db" | head -1); [ -n "$db" ] && { python tools/rocpd_stats.py "$db" > $O/lw_kernel_stats.md 2>/dev/null; python tools/rocpd_timeline.py "$db" "sample_coarse" 1 -2 > $O/lw_timeline.md; }; rm -rf $O/trace_lw; head -14 $O/lw_kernel_stats.md | cut -c1-200; tail -2 $O/lw_timeline.md
