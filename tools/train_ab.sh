#!/bin/bash
# A/B of the training step (bench.py's train_step leg, 2048 rays x (64 + 64)) between the shipped library and a build variant
# (object_nerf_amd/tune/libobjnerf_<NAME>.so) and / or an environment switch:  tools/train_ab.sh TAG NAME [ROUNDS] [ENV=VAL ...]
O=gpurun_out/${1:-ab}; N=${2:-ship}; R=${3:-2}; shift 3 2>/dev/null; mkdir -p $O
for i in $(seq $R); do for t in ship $N; do
  L=$PWD/object_nerf_amd/libobjnerf_hip.so; E=""
  if [ $t != ship ]; then [ -f $PWD/object_nerf_amd/tune/libobjnerf_$t.so ] && L=$PWD/object_nerf_amd/tune/libobjnerf_$t.so; E="$*"; fi
  env OBJNERF_LIB=$L $E python bench.py --steps 1 --warmup 1 --cpu-rays 0 --train-steps 40 --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['train_step']; 'error' in t and sys.exit(str(t))
print('$t $E:', 'train ms %.3f' % t['ms_per_step'], 'host %.2f' % t['host_enqueue_ms_per_step'], 'frac %.4f' % t['roofline']['frac'], 'loss %.6f -> %.6f' % (t['loss_first'], t['loss_last']))" | tee -a $O/train_ab_$N.txt
done; done
