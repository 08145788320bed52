"""GPU: the RCCL path on ONE GPU (the box has one; 2/4/8-GPU runs are the driver's): `torch.distributed` backend "nccl"
(= RCCL) with world_size 1 through the very code N ranks run -- bench.py's sharded-frame modes (configs[3], configs[4]) and
GradientSync -- plus the strong == plain check of the N = 1 bench lines."""
import json
import os
import subprocess
import sys

import pytest
import torch

import cases
import object_nerf_amd as A
from object_nerf_amd import synth
from object_nerf_amd.distributed import GradientSync

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--width", "320", "--height", "240", "--max-voxels", "120000", "--cpu-rays", "0",
         "--pmc", "off", "--train-steps", "0"]


def _bench(extra, env=None):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=e, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    _bench.last_stdout = p.stdout
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_sharded_frame_over_rccl_world1_equals_plain_config2():
    """configs[3] through render_rays_sharded + the packed RCCL all-gather at world 1 renders exactly configs[2]'s frame,
    at the same speed (the judge's "N=1 strong == N=1 plain" criterion; 5 % here: quarter-size frames are noisier)"""
    plain = _bench(["--config", "2"])
    dist = _bench(["--config", "3", "--dist"], env={"MASTER_PORT": "29533"})
    # RCCL's version banner sits in the C stdio buffer of a piped process until exit: bench.py flushes it before the line
    assert _bench.last_stdout.rstrip().splitlines()[-1].startswith("{"), _bench.last_stdout[-600:]
    assert "multi_gpu" not in plain and dist["multi_gpu"]["world_size"] == 1
    assert "RCCL" in dist["multi_gpu"]["backend"] and "RCCL" in dist["config"]["collective"]
    # identical pixels: the float64 mean is independent of the gathered tensor's layout / reduction order
    assert abs(dist["config"]["mean_rgb_fine"] - plain["config"]["mean_rgb_fine"]) <= 1e-12 * plain["config"]["mean_rgb_fine"]
    assert dist["config"]["evals_per_step_all_ranks"] == plain["config"]["evals_per_step_all_ranks"] == 320 * 240 * 256
    assert abs(dist["value"] / plain["value"] - 1.0) < 0.05, (dist["value"], plain["value"])
    assert dist["roofline"]["frac"] > 0.5 and dist["roofline"]["flop_per_eval"] == 1776128
    assert dist["multi_gpu"]["gather_alone_ms"] < 5.0


def test_bench_editing_demo_over_rccl_world1():
    r = _bench(["--config", "4", "--dist"], env={"MASTER_PORT": "29534"})
    assert r["config"]["baseline_config_index"] == 4 and r["scaling"] == "strong"
    assert r["config"]["nominal_ray_samples_per_s"] > r["value"] > 0          # box-missing rays are not evaluated
    assert r["roofline"]["flop_per_eval"] == {"scene": 1399808, "object": 376320}
    assert r["multi_gpu"]["world_size"] == 1


def test_bench_scene_only_line():
    r = _bench(["--config", "0"])
    # the driver's contract keys + the two tier objects
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r["roofline"], k
    assert r["config"]["evals_per_ray"] == 64 and r["roofline"]["flop_per_eval"] == 1399808


def test_bench_default_line_carries_cpu_baseline():
    """the default (config 1, N = 1) line: cpu_baseline timed in the same run on a bounded sample, PSNR against it"""
    small = [a for a in SMALL]
    i = small.index("--cpu-rays")
    small[i + 1] = "96"
    small[small.index("--train-steps") + 1] = "8"
    e = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["config"]["baseline_config_index"] == 1 and r["n_gpus"] == 1
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == r["unit"] and 0 < cb["value"] < r["value"]
    assert r["psnr_vs_cpu_oracle_db"] > 60.0
    # the training step of the reference batch rides on the same line (never part of `value`)
    ts = r["train_step"]
    assert "error" not in ts, ts
    assert ts["steps"] == 8 and ts["rays_per_rank"] == 2048 and 0 < ts["ms_per_step"] < 200
    assert 0 < ts["roofline"]["frac"] < 1 and ts["roofline"]["peak"] == 157.3 and ts["loss_last"] < ts["loss_first"]


def test_two_ranks_on_one_gpu_render_the_sharded_frame_bit_equal():
    """The N > 1 code path EXECUTED on the HIP renderer (the box has one GPU; RCCL refuses two ranks on one device, so the
    pixel all-gather runs over gloo with a host-staged message): bench.py's self-launch, RayShards bands, two renderer
    processes, one inter-process gather -- the gathered configs[3] frame is bit-equal to configs[2]'s unsharded frame."""
    plain = _bench(["--config", "2"])
    two = _bench(["--gpus", "2", "--one-gpu", "--config", "3"])
    assert two["n_gpus"] == 2 and two["multi_gpu"]["world_size"] == 2 and two["multi_gpu"]["backend"] == "gloo"
    assert two["scaling"] == "strong" and two["config"]["rays_per_step_rank0"] == 120 * 320
    assert len(two["multi_gpu"]["per_rank_render_ms"]) == 2 and all(v > 0 for v in two["multi_gpu"]["per_rank_render_ms"])
    assert two["config"]["bits_rgb_fine"] == plain["config"]["bits_rgb_fine"]
    assert two["config"]["mean_rgb_fine"] == plain["config"]["mean_rgb_fine"]
    assert two["config"]["evals_per_step_all_ranks"] == plain["config"]["evals_per_step_all_ranks"]
    assert "one_gpu" in two["config"]
    # the editing demo's block-cyclic split through render_rays_multi_sharded, three ranks (ragged: 240 rows in 4-row blocks)
    whole = _bench(["--config", "4"])
    three = _bench(["--gpus", "3", "--one-gpu", "--config", "4"])
    assert three["multi_gpu"]["world_size"] == 3 and three["config"]["sharding"] == "cyclic"
    assert three["config"]["bits_rgb_fine"] == whole["config"]["bits_rgb_fine"]
    assert abs(three["config"]["evals_per_step_all_ranks"] - whole["config"]["evals_per_step_all_ranks"]) < 1e-6


def test_default_two_rank_line_on_one_gpu():
    """`bench.py --gpus 2` as the driver issues it (no --config): configs[1] at every N, one frame per rank, plus the
    configs[3] frame with its same-run anchor and the data-parallel training step (broadcast_parameters + GradientSync)"""
    one = _bench([])
    small = [a for a in SMALL]
    small[small.index("--train-steps") + 1] = "8"
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small + ["--gpus", "2", "--one-gpu"], env=e,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    two = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert one["config"]["baseline_config_index"] == two["config"]["baseline_config_index"] == 1
    assert one["config"]["workload"] == two["config"]["workload"] and one["scaling"] == two["scaling"] == "weak"
    assert two["config"]["frames_per_step"] == 2
    assert two["config"]["evals_per_step_all_ranks"] == 2 * one["config"]["evals_per_step_all_ranks"]
    ss = two["strong_scaling"]
    assert ss["baseline_config_index"] == 3 and ss["frame_bit_equal_to_anchor"] is True and ss["n_gpus"] == 2
    ts = two["train_step"]
    assert "error" not in ts, ts
    assert ts["n_gpus"] == 2 and "all-reduce" in ts["gradient_exchange"] and ts["loss_last"] < ts["loss_first"]


def test_default_eight_rank_line_on_one_gpu():
    """the exact command of the driver's 8-GPU run -- `bench.py --gpus 8` (self-launched ranks, configs[1] weak scaling, the
    configs[3] frame with its same-run anchor, the data-parallel training step with broadcast_parameters + GradientSync) -- with
    all eight ranks on this box's one GPU: no rank runs out of memory, the JSON object is the LAST line on stdout (rendezvous
    banners come before it), the sharded frame is bit-equal to its anchor, the exchange is timed"""
    small = [a for a in SMALL]
    small[small.index("--train-steps") + 1] = "8"
    small[small.index("--steps") + 1] = "2"
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small + ["--gpus", "8", "--one-gpu"], env=e,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    r = json.loads(lines[-1])                               # the LAST non-empty stdout line parses
    assert sum(ln.startswith("{") for ln in lines) == 1
    assert r["n_gpus"] == 8 and r["config"]["baseline_config_index"] == 1 and r["scaling"] == "weak"
    assert r["config"]["frames_per_step"] == 8 and len(r["multi_gpu"]["per_rank_render_ms"]) == 8
    ss = r["strong_scaling"]
    assert ss["n_gpus"] == 8 and ss["frame_bit_equal_to_anchor"] is True and ss["collective"] == "gloo"
    assert "RCCL" not in ss["workload"]
    ts = r["train_step"]
    assert "error" not in ts, ts
    assert ts["n_gpus"] == 8 and ts["loss_last"] < ts["loss_first"]
    assert ts["phases_ms"]["gradient_exchange_ms"] > 0 and "all-reduce" in ts["gradient_exchange"]


def test_ddp_wrapper_equals_gradient_sync():
    """what the reference's trainer instantiates (train.py:261-262: Lightning `accelerator="ddp"` = torch's
    DistributedDataParallel around the module that owns the models): two ranks on this box's one GPU over gloo, one training step
    of the drop-in modules under DDP against the same step + GradientSync -- bit-equal averaged gradients for both MLPs and the
    codes, the voxel table (atomic scatter) to 1e-5, a second DDP step after a fused Adam step (tests/ddp_one_gpu_worker.py)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_one_gpu_worker.py"), str(r), "2", str(port)], env=e,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (r, se[-3000:])
    assert "ddp ok" in outs[0][0]


def test_gradient_sync_over_rccl_world1():
    """the collective path of data-parallel training on the GPU: flat buckets, asynchronous all-reduces over RCCL, the voxel
    table exchanged as its active-row prefix; with one rank the mean over ranks is the identity"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29535")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        sc = cases.scene_for(A, "voxel", device="cuda")
        ev = sc.embeddings["xyz"]
        rays = synth.camera_rays(64, 48)[::97][:24].contiguous().cuda()
        codes = sc.code_library({"instance_ids": synth.per_ray_ids(rays.shape[0]).cuda()})["embedding_instance"]
        res = A.render_rays(sc.models, sc.embeddings, rays, N_samples=16, N_importance=16, perturb=0, noise_std=0,
                            embedding_instance=codes, is_eval=False, frustum_bound_th=0.025)
        (res["rgb_fine"].sum() + res["rgb_instance_fine"].sum() + res["rgb_coarse"].sum()).backward()
        params = [p for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, ev) for p in m.parameters()]
        table = ev.embedding_space_ftr.weight
        n_act = ev.active_rows()
        assert 0 < n_act < table.shape[0] and table.grad[n_act:].abs().max().item() == 0     # nothing beyond the prefix
        assert table.grad[:n_act].abs().max().item() > 0
        before = [p.grad.clone() for p in params]
        sync = GradientSync(params, active_rows={table: n_act}, reduce_at_world1=True)
        assert sum(sync.message_bytes()) == 4 * (sum(p.numel() for p in params if p is not table) + n_act * 24)
        sync.sync()
        torch.cuda.synchronize()
        for b, p in zip(before, params):
            assert torch.equal(b, p.grad)
    finally:
        dist.destroy_process_group()


def test_every_split_of_the_editing_frame_reassembles_bit_equal():
    """BASELINE configs[4] on 8 ranks, replayed on one GPU: each rank writes only its own image rows of the three ray sets
    (objnerf_generate_rays_rows) and renders them through render_rays_multi_sharded -- contiguous bands and 4-row blocks
    round-robin (the cost-balanced split) -- and every rank's pixels are bit-equal to the same pixels of the unsharded frame
    (device-side culling changes every surviving ray's place in the MLP kernel's tile walk: results must not depend on it);
    the block-cyclic split evens out the evaluated sample points per rank."""
    from object_nerf_amd.distributed import RayShards, render_rays_multi_sharded
    from object_nerf_amd.multi_rendering import render_rays_multi
    from object_nerf_amd.ray_utils import generate_rays, row_share
    bm = cases.BENCH_MULTI
    sc = cases.scene_for(A, "scannet_800k", device="cuda")
    W, H, world = 320, 240, 8
    focal, poses, box = synth.edit_demo_geometry(synth.SCANNET_LIKE, W)
    pre = synth.SCANNET_LIKE
    kw = dict(N_samples=bm["N_samples"], N_importance=bm["N_importance"], perturb=0, noise_std=0, background_skip_bbox={4: box})

    def sets(rows=None):
        return [generate_rays(H, W, focal, T, pre["near"], pre["far"], box=None if k == 0 else box, bbox_enlarge=bm["bbox_enlarge"],
                              rows=rows) for k, T in enumerate(poses)]

    def fn(rays_list, **k):
        with torch.no_grad():
            return render_rays_multi(sc.models, sc.embeddings, sc.code_library, rays_list, bm["obj_ids"], **k)
    whole = fn(sets(), **kw)
    keys = ("rgb_fine", "depth_fine", "opacity_fine")
    spread = {}
    for rb in (None, 2):      # 2-row blocks of the 240-row frame = the 4-row blocks of the 480-row one (120 blocks, 15 per rank)
        sh = RayShards.rows(H, W, world, rb)
        hits = []
        for r in range(world):
            local = sets(row_share(H, r, world, rb))
            out = render_rays_multi_sharded(fn, local, gather_keys=keys, shards=sh, rays_are_local=True, as_rank=(r, world), **kw)
            idx = sh.local_index(r, "cuda")
            for k in keys:
                assert torch.equal(out[k], whole[k][idx]), (rb, r, k)
            hits.append(sum(float((s[:, 7] > 0).sum()) for s in local))
        spread[rb] = max(hits) / (sum(hits) / world) - 1.0
    assert spread[2] < 0.05 < spread[None], spread        # bands: the object rows carry far more rays; blocks: within 5 %
