R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06ad; mkdir -p $O; cd $R; export TMPDIR=/tmp
OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_sd.so timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train_sd.txt 2>&1; echo "train tests (sd) rc=$?"; tail -2 $O/tests_train_sd.txt | cut -c1-300
for i in 1 2 3 4; do for t in ship sd; do
  L=$R/object_nerf_amd/libobjnerf_hip.so; [ $t != ship ] && L=$R/object_nerf_amd/tune/libobjnerf_$t.so
  OBJNERF_LIB=$L python bench.py --steps 1 --warmup 1 --cpu-rays 0 --train-steps 40 --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['train_step']; p=t['phases_ms']['ms']
print('$t:', 'train ms %.3f' % t['ms_per_step'], 'forward %.3f dgrad %.3f wgrad %.3f' % (p['forward'], p['dgrad'], p['wgrad']))" | tee -a $O/train_ab_direct.txt
done; done
