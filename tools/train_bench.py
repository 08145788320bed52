#!/usr/bin/env python
"""Times one training step of the reference's batch shape (train.py:147-180: 2048 rays, 64 coarse + 64 fine,
perturb = 1, noise_std = 1, scene + object branches, voxel embedding) on the HIP training path:
render_rays forward + backward (+ Adam step).  Not the headline metric; recorded in DESIGN.md.
Under torch.distributed.run (one rank per GPU, RCCL) every rank trains on its own ray batch and GradientSync averages
the gradients before the optimizer step (data parallel, as the reference's Lightning DDP)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402


def main(n_rays=2048, steps=5):
    import torch.distributed as dist
    from object_nerf_amd.distributed import GradientSync
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = "cuda"
    sc = synth.build_scene(A, True, preset=synth.SCANNET_LIKE, max_voxels=800_000, device=dev)
    rays_all = synth.camera_rays(640, 480).to(dev)
    params = [p for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"]) for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, fused=os.environ.get("OBJNERF_BENCH_ADAM", "fused") == "fused")
    sync = GradientSync(params)
    g = torch.Generator(device=dev).manual_seed(rank)
    target = torch.rand(n_rays, 3, device=dev, generator=g)
    zeros = torch.zeros(n_rays, device=dev)

    def step(timed=True):
        idx = torch.randint(0, rays_all.shape[0], (n_rays,), device=dev, generator=g)
        rays = rays_all[idx].contiguous()
        ids = synth.per_ray_ids(n_rays).to(dev)
        opt.zero_grad(set_to_none=True)
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        r = A.render_rays(sc.models, sc.embeddings, rays, N_samples=64, N_importance=64, perturb=1.0, noise_std=1.0,
                          embedding_instance=codes, frustum_bound_th=0.025, pass_through_mask=(ids == 1).view(-1, 1))
        mse = torch.nn.functional.mse_loss      # the reference's loss terms are nn.MSELoss (models/losses.py)
        loss = sum(mse(r["rgb_%s" % t], target) + mse(r["rgb_instance_%s" % t], target)
                   + 0.1 * mse(r["depth_%s" % t], zeros) + mse(r["opacity_instance_%s" % t], zeros) for t in ("coarse", "fine"))
        if not timed:                       # a training loop as it runs: nothing waits for the device inside a step
            loss.backward()
            sync.sync()
            opt.step()
            return loss
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        sync.sync()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        return loss.item(), t1, t2, t3

    step()
    rows = []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, t1, t2, t3 = step()
        rows.append((t1 - t0, t2 - t1, t3 - t2, loss))
    fw, bw, op = (sorted(r[i] for r in rows)[len(rows) // 2] for i in range(3))
    # steady state: 8 steps in flight at most (the launch queue is bounded by a synchronisation every 8 steps), as a loop
    # that reads its loss for logging every few steps does; the three per-phase figures above each end in a host
    # synchronisation and an idle device, which costs the step ~2-3 ms of launch latency and clock ramp
    n_free = int(os.environ.get("TRAIN_BENCH_FREE_STEPS", "48"))      # 0: skip (counter-collection runs serialise every kernel)
    for _ in range(8 if n_free else 0):
        step(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n_free):
        step(False)
        if i % 8 == 7:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / max(n_free, 1)
    evals = n_rays * 192 * world
    flop = evals * 1776128 * 3.0          # forward + dgrad + wgrad
    if rank == 0:
        print("train step %d rays/rank x %d ranks: forward %.1f ms, backward %.1f ms, Adam %.1f ms -> %.2f M ray-samples/s, "
              "%.1f TFLOP/s (3x forward FLOP), loss %.4f -> %.4f"
              % (n_rays, world, fw * 1e3, bw * 1e3, op * 1e3, evals / (fw + bw + op) / 1e6, flop / (fw + bw) / 1e12,
                 rows[0][3], rows[-1][3]))
        if n_free:
            print("steady state (no host synchronisation inside a step, %d steps): %.2f ms per step incl. Adam -> %.2f M ray-samples/s, "
              "%.1f TFLOP/s (3x forward FLOP over the whole step)" % (n_free, free * 1e3, evals / free / 1e6, flop / free / 1e12))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
