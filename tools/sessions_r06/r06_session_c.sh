R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python tools/coarse_worst_ray.py $O/coarse_worst_ray.md > $O/worst.log 2>&1; echo "worst rc=$?"; sed -n '/Compositing alone/,$p' $O/worst.log | cut -c1-260
timeout 600 python tools/small_batch.py $O/small_batch.md > $O/small_batch.log 2>&1; echo "small rc=$?"; tail -8 $O/small_batch.log
timeout 600 python -m pytest tests -x -q -m gpu -k "edges or stages" > $O/tests_k.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests_k.txt | cut -c1-200
