#!/usr/bin/env python
"""Summarises the rocprofv3 PMC passes of tools/pmc_run.sh for the fused MLP kernel into profiles/<tag>_pmc.json (read by
bench.py for roofline.traffic).  One pass per counter group, one 640x480 frame = two launches of the kernel per pass.
Corrections (MI355X_MICROARCH.md, "HBM [CDNA4]"): FETCH_SIZE / WRITE_SIZE are in KB and come from the L2's memory-side
request counters (Infinity-Cache hits included); on gfx950 FETCH_SIZE reports exactly half of the bytes of wide
(16 B/lane) coalesced reads -- which is what this kernel's weight DMA and feature-row gathers are -- so it is doubled;
WRITE_SIZE is taken as reported.  FETCH_SIZE and WRITE_SIZE need separate passes (TCC counter slots).
Usage: python tools/pmc_summary.py <pmc dir> <out json> [evals per frame = 58982400]"""
import csv
import glob
import json
import os
import sys

KERNEL = "mlp_kernel"


def main(pmc_dir, out_json, evals=58982400, mfma_per_32=13536):
    """mfma_per_32: v_mfma_f32_32x32x2_f32 per 32 points with both branches: 13,876 contracted per sample point, 13,536 with
    the per-ray constant terms hoisted (round 3 default; OBJNERF_HOIST=0 -> pass 13876)"""
    sums, wall = {}, None
    for f in sorted(glob.glob(os.path.join(pmc_dir, "pass*", "*counter_collection.csv"))):
        per_dispatch = {}
        for row in csv.DictReader(open(f)):
            if KERNEL not in row["Kernel_Name"]:
                continue
            sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            per_dispatch[row["Dispatch_Id"]] = (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
        if per_dispatch and wall is None:
            wall = sum(e - s for s, e in per_dispatch.values()) * 1e-9
            launches = len(per_dispatch)
    g = sums.get
    evals = evals * launches / 2.0          # `evals` is per frame = two launches; a pass with a warm-up frame has four
    mfma = g("SQ_INSTS_MFMA", 0.0)
    d = {"launches_per_pass": launches, "mfma_instructions": mfma, "mfma_instructions_expected": evals / 32.0 * mfma_per_32,
         "kernel_wall_s_all_launches": wall}
    if mfma:
        d["mfma_busy_cycles_per_instruction"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / mfma
        d["non_mfma_valu_per_mfma"] = (g("SQ_INSTS_VALU", 0.0) - mfma) / mfma
    if g("GRBM_GUI_ACTIVE") and wall:
        cyc = g("GRBM_GUI_ACTIVE") / 8.0                       # 8 XCDs
        d["shader_cycles_per_xcd"] = cyc
        d["effective_clock_GHz"] = cyc / wall / 1e9
        d["mfma_pipe_busy_fraction"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)   # 1024 SIMDs
    if g("SQ_WAVE_CYCLES"):
        for k, name in (("SQ_WAIT_ANY", "wave_wait_any_fraction"), ("SQ_WAIT_INST_ANY", "wave_wait_inst_any_fraction"),
                        ("SQ_ACTIVE_INST_ANY", "wave_active_inst_fraction")):
            if g(k) is not None:
                d[name] = g(k) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_BANK_CONFLICT") is not None:
        d["lds_bank_conflict_cycles"] = g("SQ_LDS_BANK_CONFLICT")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["fetch_bytes_per_launch_corrected_x2"] = 2.0 * g("FETCH_SIZE") * 1024.0 / launches
        d["write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0 / launches
        d["hbm_traffic_bytes_per_launch"] = d["fetch_bytes_per_launch_corrected_x2"] + d["write_bytes_per_launch"]
        d["hbm_traffic_bytes_per_eval"] = d["hbm_traffic_bytes_per_launch"] * launches / evals
    json.dump({"workload": "python bench.py --steps 1 --warmup 1 --cpu-rays 0 (640x480 frames, 64 + 64, scene + object, voxel; counters summed over "
                           "the kernel's launches of a pass)",
               "raw_sum_over_launches": sums, "derived": d}, open(out_json, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *[int(x) for x in sys.argv[3:5]])
