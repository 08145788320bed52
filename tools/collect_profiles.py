#!/usr/bin/env python
"""Copies the artefacts of one `tools/gpu_session.sh TAG ...` call (gpurun_out/TAG/) into profiles/ under round names and writes
the markdown wrappers (kernel tables with a one-paragraph reading each).
usage: python tools/collect_profiles.py gpurun_out/r04d r04"""
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tools"))
O = os.path.join(R, sys.argv[1])
TAG = sys.argv[2]
P = os.path.join(R, "profiles")
PEAK = 157.3e12


def line(f):
    return json.loads([ln for ln in open(f) if ln.startswith("{")][-1])


def rows(path):
    out = []
    for ln in open(path):
        c = [x.strip() for x in ln.split("|")]
        if len(c) > 6 and c[2].isdigit():
            out.append(dict(name=c[1], calls=int(c[2]), total=float(c[3]), avg=float(c[4])))
    return out


def head(path, n):
    return "".join(open(path).readlines()[:n])


def have(name):
    return os.path.exists(os.path.join(O, name))


if have("bench.json"):
    shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, TAG + "_bench.json"))
    b = line(os.path.join(O, "bench.json"))
    print("headline %.2f M ray-samples/s, %.1f ms, frac %.4f; train_step %.2f ms" % (
        b["value"] / 1e6, b["ms_per_step"], b["roofline"]["frac"], b.get("train_step", {}).get("ms_per_step", float("nan"))))
for src, dst in (("bench_two_ranks_one_gpu.json", "_bench_two_ranks_one_gpu.json"), ("bench_eight_ranks_one_gpu.json", "_bench_eight_ranks_one_gpu.json"),
                 ("train_timeline.md", "_train_timeline.md"), ("arch_bench.md", "_arch_bench.md"), ("parity.md", "_parity.md"), ("small_batch.md", "_small_batch.md"),
                 ("mesh_query.md", "_mesh_query.md")):
    if have(src):
        shutil.copy(os.path.join(O, src), os.path.join(P, TAG + dst))

if have("trace_bench_kernel_stats.md") and have("bench.json"):
    ks = os.path.join(O, "trace_bench_kernel_stats.md")
    rr = rows(ks)
    mlp = next(r for r in rr if "mlp_kernel<true, true, true, true" in r["name"])
    n, S, I = 307200, 64, 64
    flop_launch = n * (S + S + I) * 1776128 / 2.0
    hbm = ["| kernel | calls | avg us | algorithmic MB / launch | GB/s | of 8 TB/s | note |", "|---|---|---|---|---|---|---|"]
    for key, bytes_, note in (
            ("composite_finish", n * (10 * (S + S + I) / 2.0 + 40), "mean of the coarse (S) and fine (S + I) launch"),
            ("ray_bias_kernel", n * (268 + 1792), "round 6: the 448 x 91 product per ray on the matrix pipe, whole-line stores through an LDS patch; products 144 us and stores 138 us overlap by a third (r06_ray_bias_probe.txt)"),
            ("sample_coarse", n * (32 + 4 * S), ""),
            ("sample_pdf_merge", n * (8 * S + 4 * (S + I)), "round 6: persistent waves, one element per lane, branch-free fixed-trip searches, next ray prefetched; float64 prefix scan of the cdf")):
        r = next((x for x in rr if key in x["name"]), None)
        if r:
            rate = bytes_ / (r["avg"] * 1e-3)
            hbm.append("| %s | %d | %.1f | %.1f | %.0f | %.2f | %s |" % (r["name"].split("(")[0].replace("objnerf::", ""), r["calls"], r["avg"] * 1e3,
                                                                      bytes_ / 1e6, rate / 1e9, rate / 8e12, note))
    roof = b["roofline"]
    open(os.path.join(P, TAG + "_kernel_stats.md"), "w").write(
        "# %s -- rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --warmup 1 --cpu-rays 0 --pmc off "
        "--train-steps 0` (1x MI355X, `tools/gpu_session.sh %s trace`)\n\n"
        "bench.py line of the same session (`profiles/%s_bench.json`): %.2f M ray-samples/s, %.1f ms per step; HIP-event average of an MLP launch "
        "in the JSON's roofline block %.2f ms (%.3f of the fp32-MFMA peak on ALGORITHMIC FLOP); rocprof average of the same kernel below: %.2f ms -> "
        "%.2f TFLOP / %.4f s = %.1f TFLOP/s = %.3f.  Per frame: 2 launches of the persistent MLP kernel (HOIST instantiation, compositing in the "
        "epilogue), `ray_bias_kernel` x 2 (its A-operand stream is gathered by `pack_all_kernel`, the one launch that re-gathers both models' "
        "weight streams at every call since round 6), `composite_finish_block_kernel` x 2, `sample_pdf_merge64_kernel`, `sample_coarse4_kernel`.\n\n"
        % (TAG, os.path.basename(O), TAG, b["value"] / 1e6, b["ms_per_step"], roof["avg_launch_ms"], roof["frac"], mlp["avg"], flop_launch / 1e12,
           mlp["avg"] / 1e3, flop_launch / (mlp["avg"] / 1e3) / 1e12, flop_launch / (mlp["avg"] / 1e3) / PEAK)
        + head(ks, 16) + "\n## HBM-bound stages of the same trace (algorithmic bytes of SURVEY.md 8d / time)\n\n" + "\n".join(hbm) + "\n")

if have("trace_train_kernel_stats.md"):
    tk = os.path.join(O, "trace_train_kernel_stats.md")
    rr = rows(tk)
    steps = next(r for r in rr if "mlp_bwd_kernel" in r["name"])["calls"] / 2.0
    tot = sum(r["total"] for r in rr)

    def grp(*keys):
        return sum(r["total"] for r in rr if any(k in r["name"] for k in keys)) / steps
    g = [("fused forward with saved activations (`mlp_kernel<..., SAVE>`)", grp("mlp_kernel")), ("dgrad chain (`mlp_bwd_kernel`)", grp("mlp_bwd_kernel")),
         ("weight gradients, 256 x 256 tiles (`wgrad_big_kernel`)", grp("wgrad_big")), ("weight gradients, 128 x 128 tiles (`wgrad_units_kernel<false>`)", grp("wgrad_units_kernel<false>")),
         ("weight gradients, ragged tiles (`wgrad_units_kernel<true>`)", grp("wgrad_units_kernel<true>")), ("weight gradients, fix-up + heads", grp("fixup", "heads_wgrad")),
         ("per-ray code gradient (`gemm_kernel`; the embedding gradients are formed inside the chain since round 6)", grp("gemm_kernel")), ("voxel embedding forward + table scatter", grp("voxel_embed"))]
    g.append(("everything else (compositing / sampling forward + backward, sigmoid, per-ray sums, Adam, fills, the loss's torch element-wise kernels)", tot / steps - sum(v for _, v in g)))
    txt = "Kernel time per step by group (the table below, / %d steps):\n\n| group | ms per step |\n|---|---|\n" % steps
    txt += "".join("| %s | %.2f |\n" % x for x in g) + "| sum of kernel time | %.2f |\n\n" % (tot / steps)
    ts = b.get("train_step") if have("bench.json") else None
    open(os.path.join(P, TAG + "_train_kernel_stats.md"), "w").write(
        "# %s -- training step (row f1): rocprofv3 --kernel-trace --stats of `python tools/train_bench.py` (1x MI355X; %d steps; 2048 rays x (64 + 64), "
        "scene + object branches, voxel embedding, perturb / noise on, occlusion mask, Adam)\n\n" % (TAG, steps)
        + ("The driver-visible figure of the same library: `train_step` block of `profiles/%s_bench.json`: %.2f ms per free-running step = %.1f TFLOP/s on "
           "3 x forward FLOP = %.3f of the fp32-MFMA peak.\n\n" % (TAG, ts["ms_per_step"], ts["roofline"]["achieved"], ts["roofline"]["frac"]) if ts else "")
        + txt + head(tk, 28))

if have("pass1"):
    import pmc_summary
    pmc_summary.main(O, os.path.join(P, TAG + "_pmc.json"))
    d = json.load(open(os.path.join(P, TAG + "_pmc.json")))["derived"]
    open(os.path.join(P, TAG + "_pmc.md"), "w").write(
        "# %s -- PMC counters of the MLP kernel, fp32-MFMA mode, HOIST instantiation, compositing in the epilogue (`tools/pmc_run.sh`: one rocprofv3 pass "
        "per counter group, no trace domains; `tools/pmc_summary.py`)\n\n```\n" % TAG + json.dumps(d, indent=1) + "\n```\n\n"
        "* `mfma_instructions` against `mfma_instructions_expected` (13,536 x evals / 32): the kernel issues exactly the hoisted instruction count; matrix pipe busy "
        "%.3f at %.2f GHz.\n* HBM-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 corrections of MI355X_MICROARCH.md): %.2f GB against ~0.94 GB of "
        "algorithmic bytes (per sample point: z 4, local weights 4, records 2, ray vectors 19, rays / codes 3 B).  The x2 correction over-counts the per-ray vectors "
        "(550 MB of ordinary 64-byte reads); the rest of the excess is voxel-table rows fetched by more than one XCD's L2.  At ~10 GB/s it costs no time (the kernel is "
        "MFMA-bound); it is reported because the tier asks for it.\n"
        % (d.get("mfma_pipe_busy_fraction", float("nan")), d.get("effective_clock_GHz", float("nan")), d.get("hbm_traffic_bytes_per_launch", float("nan")) / 1e9))
