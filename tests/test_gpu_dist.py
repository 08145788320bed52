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
         "--split-bf16-steps", "0", "--pmc", "off"]


def _bench(extra, env=None):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=e, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_sharded_frame_over_rccl_world1_equals_plain_config2():
    """configs[3] through render_rays_sharded + the packed RCCL all-gather at world 1 renders exactly configs[2]'s frame,
    at the same speed (the judge's "N=1 strong == N=1 plain" criterion; 5 % here: quarter-size frames are noisier)"""
    plain = _bench(["--config", "2"])
    dist = _bench(["--config", "3", "--dist"], env={"MASTER_PORT": "29533"})
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
    e = dict(os.environ)
    e.pop("OBJNERF_MFMA", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["config"]["baseline_config_index"] == 1 and r["n_gpus"] == 1
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == r["unit"] and 0 < cb["value"] < r["value"]
    assert r["psnr_vs_cpu_oracle_db"] > 60.0


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
