#!/bin/bash
# One GPU session (a `gpurun` call), parameterised: tools/gpu_session.sh TAG STEP [STEP ...]
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r04a tests smoke mesh bench dist2'
# Every step writes under gpurun_out/TAG/ (merged back into the build container) and prints one status line.
# Steps:
#   tests       pytest -m gpu (whole suite, -x)            tests:EXPR   only the tests matching -k EXPR
#   smoke       __graft_entry__.smoke()
#   bench       python bench.py (driver's default: steps 20, warmup 5)       bench:K   with --config K
#   dist2       bench.py --gpus 2 --one-gpu (two ranks on cuda:0 over gloo: the N > 1 code path on one GPU)      dist8  the same with 8 ranks
#   trainpmc    tools/train_pmc.sh (counter passes of the training step)     timeline   launch-order timeline of one training step
#   train       tools/train_bench.py                       mesh   tools/mesh_query_bench.py 512
#   trace       rocprofv3 --kernel-trace --stats of bench.py (headline) -> kernel table
#   trace:train the same of tools/train_bench.py           trace:mesh  of the density query
#   pmc         tools/pmc_run.sh (counter passes of the headline frame)
#   parity      tools/parity_report.py + tools/frame_parity.py (per-key tables against the reference goldens)
#   small       tools/small_batch.py                       edit   tools/edit_bench.py          arch   tools/arch_bench.py
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
trace() {   # trace NAME CMD...: kernel trace + per-kernel table (rocpd database summarised by tools/rocpd_stats.py)
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o tr -- "$@" > $O/trace_$name.log 2>&1); echo "trace $name rc=$?"
  local db=$(find $O/trace_$name -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/trace_${name}_kernel_stats.md 2>/dev/null
  rm -rf $O/trace_$name
  head -12 $O/trace_${name}_kernel_stats.md | cut -c1-160
}
for step in "$@"; do
  arg=${step#*:}; [ "$arg" == "$step" ] && arg=""
  case ${step%%:*} in
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -x -q -rP -m gpu -k "$arg" > $O/gpu_tests_k.txt 2>&1; echo "gpu tests -k '$arg' rc=$?"; tail -3 $O/gpu_tests_k.txt
           else timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt ;;
    bench) timeout 900 python bench.py --steps 20 --warmup 5 ${arg:+--config $arg} > $O/bench${arg:+_c$arg}.json 2> $O/bench${arg:+_c$arg}.err; echo "bench $arg rc=$?"; tail -1 $O/bench${arg:+_c$arg}.json | cut -c1-400 ;;
    dist2) timeout 900 python bench.py --gpus 2 --one-gpu --steps 5 --warmup 2 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err; echo "dist2 rc=$?"; tail -1 $O/bench_two_ranks_one_gpu.json | cut -c1-300 ;;
    dist8) timeout 900 python bench.py --gpus 8 --one-gpu --steps 2 --warmup 1 > $O/bench_eight_ranks_one_gpu.json 2> $O/bench_eight_ranks_one_gpu.err; echo "dist8 rc=$?"; tail -1 $O/bench_eight_ranks_one_gpu.json | cut -c1-300 ;;
    trainpmc) timeout 1500 bash tools/train_pmc.sh $O/train_pmc.md > $O/train_pmc.log 2>&1; echo "trainpmc rc=$?"; tail -3 $O/train_pmc.log | cut -c1-200 ;;
    timeline) (cd /tmp && TRAIN_BENCH_FREE_STEPS=24 timeout 600 rocprofv3 --kernel-trace -d $O/trace_tl -o tr -- python $R/tools/train_bench.py > $O/trace_tl.log 2>&1); echo "timeline rc=$?"
              db=$(find $O/trace_tl -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_timeline.py "$db" "mlp_kernel" 2 -3 > $O/train_timeline.md; rm -rf $O/trace_tl; tail -1 $O/train_timeline.md ;;
    train) timeout 300 python tools/train_bench.py > $O/train_bench.txt 2>&1; echo "train rc=$?"; tail -2 $O/train_bench.txt ;;
    mesh)  timeout 600 python tools/mesh_query_bench.py 512 $O/mesh_query.md > $O/mesh_query.log 2>&1; echo "mesh rc=$?"; tail -8 $O/mesh_query.log ;;
    small) timeout 600 python tools/small_batch.py $O/small_batch.md > $O/small_batch.log 2>&1; echo "small rc=$?"; tail -8 $O/small_batch.log ;;
    parity) timeout 900 python tools/parity_report.py $O/parity.md > $O/parity.log 2>&1; echo "parity rc=$?"; tail -16 $O/parity.md | cut -c1-160
            timeout 600 python tools/frame_parity.py $O/frame_parity.md > $O/frame_parity.log 2>&1; echo "frame parity rc=$?"; tail -6 $O/frame_parity.md | cut -c1-200 ;;
    arch)  timeout 600 python tools/arch_bench.py $O/arch_bench.md > $O/arch_bench.log 2>&1; echo "arch rc=$?"; tail -6 $O/arch_bench.log ;;
    edit)  timeout 600 python tools/edit_bench.py > $O/edit_bench.txt 2>&1; echo "edit rc=$?"; tail -3 $O/edit_bench.txt ;;
    trace) case "$arg" in
             train) TRAIN_BENCH_FREE_STEPS=16 trace train python $R/tools/train_bench.py ;;
             mesh)  trace mesh python $R/tools/mesh_query_bench.py 256 ;;
             *)     trace bench python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off --train-steps 0 ;;
           esac ;;
    pmc)   timeout 900 bash tools/pmc_run.sh $O > $O/pmc.log 2>&1; echo "pmc rc=$?"; tail -5 $O/pmc.log ;;
    *) echo "unknown step $step" ;;
  esac
done
