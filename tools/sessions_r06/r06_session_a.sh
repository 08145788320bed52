R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python tools/frame_parity.py --coarse $O/coarse_variants.md "shipped=:1,nohoist=:0,nodbl=object_nerf_amd/tune/libobjnerf_nodbl.so:1,nodbl_nohoist=object_nerf_amd/tune/libobjnerf_nodbl.so:0,dbl2=object_nerf_amd/tune/libobjnerf_dbl2.so:1" > $O/coarse.log 2>&1; echo "coarse rc=$?"; tail -20 $O/coarse.log | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu -k "invalidate_packed or deep_copied or graph_replayed or ddp_wrapper or training_step_updates or capturable" > $O/new_tests.txt 2>&1; echo "new tests rc=$?"; tail -15 $O/new_tests.txt
timeout 600 python tools/small_batch.py $O/small_batch.md > $O/small_batch.log 2>&1; echo "small rc=$?"; tail -8 $O/small_batch.log
for v in shipped nodbl dbl2 shipped nodbl dbl2; do
  if [ $v == shipped ]; then unset OBJNERF_LIB; else export OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_$v.so; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --cpu-rays 0 --pmc off --train-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['value'], r['roofline']['frac'])" | tee -a $O/variant_bench.txt
done
unset OBJNERF_LIB
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt
