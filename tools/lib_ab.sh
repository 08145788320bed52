#!/bin/bash
# A/B of a build variant (object_nerf_amd/tune/libobjnerf_<NAME>.so, built by hand with a -D switch) against the shipped
# library on the headline frame: time, roofline fraction, the memory-side fetch counter, and the result bits.
#   tools/lib_ab.sh TAG NAME [ROUNDS]
O=gpurun_out/${1:-ab}; N=${2:?variant name}; R=${3:-3}; mkdir -p $O
for i in $(seq $R); do for t in ship $N; do
  L=$PWD/object_nerf_amd/libobjnerf_hip.so; [ $t != ship ] && L=$PWD/object_nerf_amd/tune/libobjnerf_$t.so
  OBJNERF_LIB=$L python bench.py --steps 8 --warmup 2 --cpu-rays 0 --train-steps 0 --pmc ${PMC:-on} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$t:', 'ms %.2f' % d['ms_per_step'], 'mlp launch %.2f ms' % r['avg_launch_ms'], 'frac %.4f' % r['frac'], ('traffic %.3f GB' % (r['traffic']/1e9)) if r.get('traffic') else 'traffic n/a', 'bits', d['config']['bits_rgb_fine'])" | tee -a $O/ab_$N.txt
done; done
