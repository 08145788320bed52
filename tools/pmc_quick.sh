#!/bin/bash
# Quick PMC look at the MLP kernel of one bench frame (clock, matrix-pipe busy, wait split).  Honors OBJNERF_LIB.
# Usage (GPU box): bash tools/pmc_quick.sh            PMC_KERNEL=ray_bias_kernel bash tools/pmc_quick.sh  (any kernel of the frame)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pq; mkdir -p /tmp/pq
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pq/pass$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-rays 0 --train-steps 0 --pmc off > /tmp/pq/log$i 2>&1
done
python - <<'PY'
import csv, glob, os
KERNEL = os.environ.get("PMC_KERNEL", "mlp_kernel")
s, wall, n = {}, 0, 0
for f in sorted(glob.glob("/tmp/pq/pass*/*counter_collection.csv")):
    seen = {}
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            s[r["Counter_Name"]] = s.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            seen[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if seen and not wall:
        wall, n = sum(seen.values()) * 1e-9, len(seen)
cyc = s["GRBM_GUI_ACTIVE"] / 8
print("launches %d, kernel wall %.1f ms, clock %.2f GHz" % (n, wall * 1e3, cyc / wall / 1e9))
print("MFMA instr %.3e, busy cycles / MFMA %.1f, matrix pipe busy %.3f" % (s["SQ_INSTS_MFMA"], s["SQ_VALU_MFMA_BUSY_CYCLES"] / s["SQ_INSTS_MFMA"], s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
print("wave cycles: wait_any %.3f  wait_inst_any %.3f  active %.3f ; non-MFMA VALU per MFMA %.2f" % (
    s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"],
    (s["SQ_INSTS_VALU"] - s["SQ_INSTS_MFMA"]) / s["SQ_INSTS_MFMA"]))
if "SQ_INSTS_LDS" in s:
    print("LDS: instr / MFMA %.2f, bank-conflict cycles / LDS-active cycles %.3f, wait_inst_lds / wave cycles %.3f; VMEM rd %.3e wr %.3e, SALU / MFMA %.2f, VMEM-active / wave cycles %.3f" % (
        s["SQ_INSTS_LDS"] / s["SQ_INSTS_MFMA"], s.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s.get("SQ_LDS_IDX_ACTIVE", 1)), s.get("SQ_WAIT_INST_LDS", 0) / s["SQ_WAVE_CYCLES"],
        s.get("SQ_INSTS_VMEM_RD", 0), s.get("SQ_INSTS_VMEM_WR", 0), s.get("SQ_INSTS_SALU", 0) / s["SQ_INSTS_MFMA"], s.get("SQ_ACTIVE_INST_VMEM", 0) / s["SQ_WAVE_CYCLES"]))
PY
