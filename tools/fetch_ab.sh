#!/bin/bash
# A/B of build variants on the bench frame: MLP-kernel time (tools/tune_mlp.py) and L2-side fetch traffic (FETCH_SIZE pass).
# Usage (GPU box): bash tools/fetch_ab.sh tagA tagB ...   ('ship' = libobjnerf_hip.so, else object_nerf_amd/tune/libobjnerf_<tag>.so)
R=${GRAFT_REPO_ROOT:-$PWD}
python $R/tools/tune_mlp.py "$@" 2>&1 | tail -$#
export TMPDIR=/tmp
cd /tmp
for t in "$@"; do
  L=$R/object_nerf_amd/tune/libobjnerf_$t.so; [ $t = ship ] && L=$R/object_nerf_amd/libobjnerf_hip.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/f_$t
    OBJNERF_LIB=$L rocprofv3 --pmc $c --output-format csv -d /tmp/f_$t -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-rays 0 > /dev/null 2>&1
    python - "$t" "$c" <<'PY'
import csv, glob, sys
t, c = sys.argv[1], sys.argv[2]
tot = n = 0
for f in glob.glob("/tmp/f_%s/*counter_collection.csv" % t):
    for r in csv.DictReader(open(f)):
        if "mlp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
            tot += float(r["Counter_Value"]); n += 1
print("%s %s per launch: %.2f GB raw (x2 for FETCH_SIZE on gfx950) over %d launches" % (t, c, tot * 1024 / max(n, 1) / 1e9, n))
PY
  done
done
