#!/usr/bin/env python
"""Copies the artefacts of tools/gpu_session_r03z.sh (gpurun_out/r03z/) into profiles/ under their round-3 names and writes the
markdown wrappers (kernel tables with their one-paragraph readings).  usage: python tools/collect_profiles.py [gpurun_out/r03z]"""
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03z")
P = os.path.join(R, "profiles")


def line(f):
    return json.loads([ln for ln in open(f) if ln.startswith("{")][-1])


def head(path, n):
    return "".join(open(path).readlines()[:n])


def row(path, name):
    for ln in open(path):
        if name in ln:
            c = [x.strip() for x in ln.split("|")]
            return dict(calls=int(c[2]), total=float(c[3]), avg=float(c[4]))
    return None


for src, dst in [("bench.json", "r03_bench.json"), ("bench_c0.json", "r03_bench_c0.json"), ("bench_c2.json", "r03_bench_c2.json"),
                 ("bench_c4.json", "r03_bench_c4.json"), ("bench_c3_dist.json", "r03_bench_c3_dist.json"), ("r03_pmc.json", "r03_pmc.json"),
                 ("r03_parity.md", "r03_parity.md"), ("r03_frame_parity.md", "r03_frame_parity.md"), ("big_frame.md", "r03_big_frame.md"),
                 ("small_batch.md", "r03_small_batch.md"), ("r03_band_replay.md", "r03_band_replay.md")]:
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))

b = line(os.path.join(O, "bench.json"))
roof = b["roofline"]
ks = os.path.join(O, "trace_kernel_stats.md")
mlp = row(ks, "mlp_kernel<true, true, true, true")
flop_launch = 307200 * 192 * 1776128 / 2.0
open(os.path.join(P, "r03_kernel_stats.md"), "w").write(
    "# Round 3 — rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off` "
    "(1x MI355X, tools/gpu_session_r03z.sh)\n\n"
    "bench.py line of the same session (`profiles/r03_bench.json`): %.2f M ray-samples/s, %.1f ms per step; HIP-event avg MLP launch in the JSON's "
    "roofline block %.2f ms (%.3f of the fp32-MFMA peak on ALGORITHMIC FLOP); rocprof avg of the same kernel below: %.1f ms -> %.2f TFLOP / %.4f s = "
    "%.1f TFLOP/s = %.3f.\nRound 3 changes visible here: the kernel is the HOIST instantiation (last template flag; 13,536 MFMAs per 32 points instead of "
    "13,876) fed by `ray_bias_weights_kernel` + `ray_bias_kernel`; `composite_kernel` is gone from the frame (the compositing starts in the MLP kernel's "
    "epilogue, `composite_finish_kernel` is its per-ray second half).\n\n"
    % (b["value"] / 1e6, b["ms_per_step"], roof["avg_launch_ms"], roof["frac"], mlp["avg"], flop_launch / 1e12, mlp["avg"] / 1e3,
       flop_launch / (mlp["avg"] / 1e3) / 1e12, flop_launch / (mlp["avg"] / 1e3) / 157.3e12)
    + head(ks, 14) + "\n## HBM-bound stages of the same trace (`tools/hbm_rates.py`)\n\n" + open(os.path.join(O, "hbm_rates.md")).read())

open(os.path.join(P, "r03_kernel_stats_config4.md"), "w").write(
    "# Round 3 — kernel table of the editing demo (`bench.py --config 4`, `render_rays_multi` in one enqueue; tools/gpu_session_r03z.sh)\n\n"
    "Per frame (3 steps traced): scene-branch MLP launches (background set) and object-branch launches (two object sets, culled rays skipped), both "
    "HOIST instantiations, `ray_bias` kernels per ray set and pass, joint compositing `composite_multi_kernel`.\n\n" + head(os.path.join(O, "trace_c4_kernel_stats.md"), 14))

def budget(path, steps):
    """per-step kernel time by group, from the kernel table"""
    rows = []
    for ln in open(path):
        c = [x.strip() for x in ln.split("|")]
        if len(c) > 5 and c[2].isdigit():
            rows.append((c[1], float(c[3])))
    tot = sum(v for _, v in rows)

    def grp(*keys):
        return sum(v for n, v in rows if any(k in n for k in keys)) / steps
    g = [("fused forward with saved activations (`mlp_kernel<..., SAVE>`)", grp("mlp_kernel")), ("dgrad chain (`mlp_bwd_kernel`)", grp("mlp_bwd_kernel")),
         ("weight gradients, 256 x 256 tiles (`wgrad_big_kernel`)", grp("wgrad_big")), ("weight gradients, 128 x 128 tiles (`wgrad_units_kernel<false>`)", grp("wgrad_units_kernel<false>")),
         ("weight gradients, ragged tiles (`wgrad_units_kernel<true>`)", grp("wgrad_units_kernel<true>")), ("weight gradients, fix-up + heads", grp("fixup", "heads_wgrad")),
         ("gradients w.r.t. the embeddings (`gemm_kernel<true, false, *>`)", grp("gemm_kernel")), ("voxel embedding forward + table scatter", grp("voxel_embed"))]
    g.append(("everything else (compositing / sampling forward + backward, sigmoid, per-ray sums, Adam, fills, torch element-wise)", tot / steps - sum(v for _, v in g)))
    out = "Kernel time per step by group (the table below, / %d steps):\n\n| group | ms per step |\n|---|---|\n" % steps
    out += "".join("| %s | %.2f |\n" % x for x in g) + "| sum of kernel time | %.2f |\n\n" % (tot / steps)
    return out


# training artefacts: the closing session after the weight-gradient rewrite (tools/gpu_session_r03zz.sh) when it exists
OT = O
for cand in ("r03zz", "r03zn"):          # the latest one that exists (r03zn: after the 256 x 256 tiles, tools/gpu_session_r03n.sh)
    if os.path.exists(os.path.join(os.path.dirname(O), cand, "trace_train_kernel_stats.md")):
        OT = os.path.join(os.path.dirname(O), cand)
tk = os.path.join(OT, "trace_train_kernel_stats.md")
tlines = [l for l in open(os.path.join(OT, "train_bench.txt")).read().strip().splitlines() if l.startswith(("train step", "steady state"))]
tb = "\n\n".join(tlines)
extra = [l for l in open(os.path.join(P, "r03_train_bench.txt")).read().splitlines() if l.startswith("OBJNERF_MFMA=bf16x3")] if os.path.exists(os.path.join(P, "r03_train_bench.txt")) else []
open(os.path.join(P, "r03_train_bench.txt"), "w").write("\n".join(tlines + extra) + "\n")   # (the split-bf16 line is added by hand from its own session)
wf, wt, fx, hw, hf = (row(tk, n) for n in ("wgrad_units_kernel<false>", "wgrad_units_kernel<true>", "wgrad_fixup_kernel", "heads_wgrad_kernel", "heads_fixup_kernel"))
wb = row(tk, "wgrad_big_kernel") or dict(calls=0, total=0.0, avg=0.0)
steps = wf["calls"] / 2.0        # two passes (coarse, fine) per step
open(os.path.join(P, "r03_train_kernel_stats.md"), "w").write(
    "# Round 3 — training step (row f1): rocprofv3 --kernel-trace --stats of `python tools/train_bench.py` (1x MI355X; %d steps: 6 with a host "
    "synchronisation after every phase + 56 back to back; 2048 rays x (64 + 128), scene + object, voxel embedding, perturb / noise on, Adam)\n\n" % steps +
    "Wall clock of the same library, un-profiled (`profiles/r03_train_bench.txt`): " + tb + "\n\n"
    "Round 3: all weight-gradient products of a backward pass in `wgrad_units_kernel<false|true>` (full / ragged tiles; work unit = tile x ~1950-point slice, "
    "ordered product / slice / tile) + `wgrad_fixup_kernel` (ordered sum of the slices: bit-reproducible) + `heads_wgrad_kernel` / `heads_fixup_kernel` "
    "(1- and 3-row heads on the VALU) instead of ~35 atomic split-K `gemm_kernel<false, false, *>` launches per pass (round 2: 8.2 + 1.2 ms per step; "
    "first grouped version of this round: 6.2 + 0.7 + 0.17 + 0.68 = 7.7 ms; after the full-tile loop rewrite and the heads rewrite, "
    "`profiles/r03_wgrad_ablations.md`: 5.9 + 0.7 + 0.18 + 0.29 = 7.1 ms; with the 256 x 256 tiles of `wgrad_big_kernel`: %.1f big tiles + %.1f other full tiles "
    "+ %.1f ragged tiles + %.2f fix-up + %.2f heads = %.1f ms); the remaining "
    "`gemm_kernel<true, false, *>` launches are the gradients w.r.t. the embeddings (1.9 ms per step as in round 2).\n\n"
    % (wb["total"] / steps, wf["total"] / steps, wt["total"] / steps, fx["total"] / steps, (hw["total"] + hf["total"]) / steps,
       (wb["total"] + wf["total"] + wt["total"] + fx["total"] + hw["total"] + hf["total"]) / steps) + budget(tk, steps) + head(tk, 26))

pm = json.load(open(os.path.join(O, "r03_pmc.json")))
d = pm["derived"]
open(os.path.join(P, "r03_pmc.md"), "w").write(
    "# Round 3 — PMC counters of the MLP kernel, fp32-MFMA mode, HOIST instantiation, compositing in the epilogue (`tools/pmc_run.sh`, one rocprofv3 pass "
    "per counter group, `tools/pmc_summary.py`)\n\n```\n" + json.dumps(d, indent=1) + "\n```\n\n"
    "* `mfma_instructions` = exactly 13,536 x evals / 32: the 340 MFMAs per 32 points of the hoisted terms are gone, nothing else; matrix pipe busy %.3f at %.2f GHz.\n"
    "* `write_bytes_per_launch` 944 MB (round 2) -> %.0f MB: sigma / rgb of both branches are no longer written (compositing in the epilogue); what is left are "
    "the local weights (4 B per sample) and the segment records.\n"
    "* `fetch_bytes_per_launch_corrected_x2`: the x2 correction of MI355X_MICROARCH.md is for the wide coalesced reads the raw counter under-reports (weight DMA, "
    "table rows).  Attribution (`tools/gpu_session_r03y.sh`: one FETCH_SIZE pass each with OBJNERF_HOIST=1 / 0, first version of the per-ray-vector kernel): raw "
    "1,640 MB vs 749 MB per launch -- the per-ray vectors (550 MB, each read by S / 32 waves) add raw fetch that is NOT under-reported, so the corrected figure "
    "over-counts them; best estimate of this run: 2 x 749 + (%.0f - 749) + %.0f (writes) MB = %.2f GB per launch against ~0.94 GB compulsory (per sample point: z 4, "
    "local weights 4, records 2, ray vectors 19, rays / codes 3 B).  At ~7 GB/s it costs no time (the kernel is MFMA-bound).\n"
    % (d["mfma_pipe_busy_fraction"], d["effective_clock_GHz"], d["write_bytes_per_launch"] / 1e6, d["fetch_bytes_per_launch_corrected_x2"] / 2e6,
       d["write_bytes_per_launch"] / 1e6, (2 * 749e6 + (d["fetch_bytes_per_launch_corrected_x2"] / 2 - 749e6) + d["write_bytes_per_launch"]) / 1e9))
print("headline %.2f M, %.2f ms, frac %.4f; mlp avg %.2f ms" % (b["value"] / 1e6, b["ms_per_step"], roof["frac"], mlp["avg"]))
