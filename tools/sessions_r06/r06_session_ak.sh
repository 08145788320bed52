R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06ak; mkdir -p $O; cd $R; export TMPDIR=/tmp
for i in 1 2 3; do for t in ship hc512 hc256; do
  L=$R/object_nerf_amd/libobjnerf_hip.so; [ $t != ship ] && L=$R/object_nerf_amd/tune/libobjnerf_$t.so
  OBJNERF_LIB=$L python bench.py --steps 1 --warmup 1 --cpu-rays 0 --train-steps 40 --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['train_step']; p=t['phases_ms']['ms']
print('$t:', 'train ms %.3f' % t['ms_per_step'], 'wgrad %.3f' % p['wgrad'])" | tee -a $O/train_ab_heads.txt
done; done
