O=gpurun_out/r04m; mkdir -p $O
for t in ship xcd1 ship xcd1; do
  L=$PWD/object_nerf_amd/libobjnerf_hip.so; [ $t != ship ] && L=$PWD/object_nerf_amd/tune/libobjnerf_$t.so
  OBJNERF_LIB=$L python bench.py --steps 8 --warmup 2 --cpu-rays 0 --train-steps 0 --pmc on 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$t', 'ms %.2f' % d['ms_per_step'], 'mlp launch %.2f ms' % r['avg_launch_ms'], 'frac %.4f' % r['frac'], 'traffic %.3f GB (fetch %.3f)' % (r['traffic']/1e9, r['traffic_fetch_bytes_per_launch']/1e9))" | tee -a $O/xcd_ab.txt
done
