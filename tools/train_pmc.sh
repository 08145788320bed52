#!/bin/bash
# PMC look at the kernels of one training step (tools/train_bench.py): matrix pipe, waits, vector-memory latency, L2.
# One rocprofv3 pass per counter group, no trace domains (a TA_* group hung rocprofv3 on this pool's boxes: not collected).  Usage (GPU box): bash tools/train_pmc.sh [outfile.md]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-/tmp/train_pmc.md}
case "$OUT" in /*) ;; *) OUT=$PWD/$OUT ;; esac
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tq; mkdir -p /tmp/tq
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  TRAIN_BENCH_FREE_STEPS=0 timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/tq/pass$i -o pmc -- python $R/tools/train_bench.py > /tmp/tq/log$i 2>&1
  echo "pass $i rc=$? ($grp)"; tail -1 /tmp/tq/log$i
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
wall = collections.defaultdict(float); cnt = collections.defaultdict(int)
def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("objnerf::", "")
    return n[:44]
for pi, f in enumerate(sorted(glob.glob("/tmp/tq/pass*/*counter_collection.csv"))):
    seen = {}
    for r in csv.DictReader(open(f)):
        key = short(r["Kernel_Name"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[(key, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if pi == 0:
        for (key, _), dt in seen.items():
            wall[key] += dt * 1e-9; cnt[key] += 1
hdr = ["kernel", "launches", "avg ms", "GHz", "mfma busy", "wait_inst_any", "active", "LDS-wait", "VALU/MFMA", "LDS/MFMA", "VMEM rd/MFMA",
       "L1->L2 rd latency cyc", "L2 rd req", "L2 hit", "EA rd GB/launch", "EA wr GB/launch", "TCP pending stall/cyc", "bank conf/LDS active"]
lines = ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
for key in sorted(wall, key=lambda k: -wall[k])[:10]:
    s = agg[key]; n = cnt[key]
    cyc = s["GRBM_GUI_ACTIVE"] / 8
    if not cyc:
        continue
    mf = max(s.get("SQ_INSTS_MFMA", 0.0), 1.0)
    wc = max(s.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    rd = max(s.get("SQ_INSTS_VMEM_RD", 0.0), 1.0)
    ea_rd = (s.get("TCC_EA0_RDREQ_sum", 0) - s.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64 + s.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    ea_wr = (s.get("TCC_EA0_WRREQ_64B_sum", 0)) * 64 + (s.get("TCC_EA0_WRREQ_sum", 0) - s.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
    lines.append("| %s | %d | %.3f | %.2f | %.3f | %.3f | %.3f | %.3f | %.2f | %.2f | %.3f | %.0f | %.3e | %.3f | %.3f | %.3f | %.3f | %.3f |" % (
        key, n, wall[key] / n * 1e3, cyc / wall[key] / 1e9, s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
        s["SQ_WAIT_INST_ANY"] / wc, s["SQ_ACTIVE_INST_ANY"] / wc, s.get("SQ_WAIT_INST_LDS", 0) / wc,
        (s["SQ_INSTS_VALU"] - s.get("SQ_INSTS_MFMA", 0)) / mf, s.get("SQ_INSTS_LDS", 0) / mf, s.get("SQ_INSTS_VMEM_RD", 0) / mf,
        s.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(s.get("TCP_TCC_READ_REQ_sum", 0), 1.0),
        s.get("TCP_TCC_READ_REQ_sum", 0) / n,
        s.get("TCC_HIT_sum", 0) / max(s.get("TCC_HIT_sum", 0) + s.get("TCC_MISS_sum", 0), 1.0),
        ea_rd / n / 1e9, ea_wr / n / 1e9, s.get("TCP_PENDING_STALL_CYCLES_sum", 0) / max(cyc * 256, 1.0),
        s.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s.get("SQ_LDS_IDX_ACTIVE", 1))))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
