#!/bin/bash
# PMC look at the training GEMM (csrc/gemm.h) on the shapes of tools/gemm_bench.py: matrix-pipe busy, wait split, LDS.
# Usage (GPU box): bash tools/gemm_pmc.sh [outfile]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-/tmp/gemm_pmc.md}
case "$OUT" in /*) ;; *) OUT=$PWD/$OUT ;; esac
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/gq; mkdir -p /tmp/gq
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/gq/pass$i -o pmc -- python $R/tools/gemm_bench.py ship > /tmp/gq/log$i 2>&1
done
tail -3 /tmp/gq/log1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
# aggregate per (kernel template, grid size) = one GEMM shape
agg = collections.defaultdict(lambda: collections.defaultdict(float))
wall = collections.defaultdict(float); cnt = collections.defaultdict(int)
for pi, f in enumerate(sorted(glob.glob("/tmp/gq/pass*/*counter_collection.csv"))):
    seen = {}
    for r in csv.DictReader(open(f)):
        if "gemm_kernel" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("gemm_kernel")[1][:22], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[(key, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if pi == 0:
        for (key, _), dt in seen.items():
            wall[key] += dt * 1e-9; cnt[key] += 1
lines = ["| kernel<A_KC,B_KC,TAIL> | grid | launches | avg ms | clock GHz | matrix pipe busy | wait_inst_any | active | LDS-wait | VALU/MFMA | LDS instr/MFMA | bank-conflict cyc / LDS active |", "|" + "---|" * 12]
for key in sorted(agg, key=lambda k: -wall[k]):
    s = agg[key]
    if not s.get("SQ_INSTS_MFMA") or not wall[key]:
        continue
    cyc = s["GRBM_GUI_ACTIVE"] / 8
    lines.append("| %s | %s | %d | %.3f | %.2f | %.3f | %.3f | %.3f | %.3f | %.2f | %.2f | %.3f |" % (
        key[0], key[1], cnt[key], wall[key] / cnt[key] * 1e3, cyc / wall[key] / 1e9, s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
        s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"], s.get("SQ_WAIT_INST_LDS", 0) / s["SQ_WAVE_CYCLES"],
        (s["SQ_INSTS_VALU"] - s["SQ_INSTS_MFMA"]) / s["SQ_INSTS_MFMA"], s.get("SQ_INSTS_LDS", 0) / s["SQ_INSTS_MFMA"],
        s.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s.get("SQ_LDS_IDX_ACTIVE", 1))))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
