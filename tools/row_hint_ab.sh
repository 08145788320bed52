#!/bin/bash
# A/B of the row_width hint (column-strip visiting order) on the headline frame: time and the memory-side fetch counter
O=gpurun_out/${1:-r04o}; mkdir -p $O
for t in 1 0 1 0; do
  OBJNERF_BENCH_ROW_HINT=$t python bench.py --steps 8 --warmup 2 --cpu-rays 0 --split-bf16-steps 0 --train-steps 0 --pmc on 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('row hint $t:', 'ms %.2f' % d['ms_per_step'], 'mlp launch %.2f ms' % r['avg_launch_ms'], 'frac %.4f' % r['frac'], 'traffic %.3f GB (fetch %.3f)' % (r['traffic']/1e9, r['traffic_fetch_bytes_per_launch']/1e9), 'bits', d['config']['bits_rgb_fine'])" | tee -a $O/row_hint_ab.txt
done
