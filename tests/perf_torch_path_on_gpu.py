#!/usr/bin/env python
"""Same-hardware comparison (not a pytest file; run by hand on the GPU box): the reference's PyTorch path -- as
restated by the oracle, which is bit-exact with the reference on CPU (tests/test_oracle_vs_reference.py) -- executed
by PyTorch-ROCm on the MI355X itself (fp32 rocBLAS/hipBLASLt GEMMs + ATen element-wise kernels, chunk = 32768 as in
models/rendering.py:246), next to the HIP path, on the bench workload (inference) and on the reference's training
batch (forward + backward).  The oracle stays a checker here: it lives under tests/ and nothing in the product
imports it.

    python tests/perf_torch_path_on_gpu.py [n_rays_inference=32768] [n_rays_train=2048]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402
from oracle import objnerf_oracle as O  # noqa: E402


def oracle_state(sc, dev):
    ev = sc.embeddings["xyz"]
    grid = dict(voxel_idx_map=ev.voxel_idx_map.to(dev), table=ev.embedding_space_ftr.weight.detach().to(dev),
                voxel_offset=ev.voxel_offset.to(dev), voxel_size=ev.voxel_size.to(dev), voxel_shape=ev.voxel_shape.to(dev))
    pc = {k: v.detach().to(dev) for k, v in sc.models["coarse"].state_dict().items()}
    pf = {k: v.detach().to(dev) for k, v in sc.models["fine"].state_dict().items()}
    return grid, pc, pf


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main(n_inf=32768, n_train=2048):
    dev = "cuda"

    def on_gpu(*a, **k):                   # the oracle's factory calls (linspace, tensor) must land on the GPU
        with torch.device(dev):
            return O.render_rays(*a, **k)
    S, I = 64, 64
    evals = S + (S + I)

    # ---- inference: bench.py's workload on a slab of the frame ----
    sc = synth.build_scene(A, True, preset=synth.TOYDESK_LIKE, device=dev)
    rays = synth.camera_rays(640, 480, near=0.05, far=1.5).to(dev)
    rays = rays[:: max(1, rays.shape[0] // n_inf)][:n_inf].contiguous()
    ids = synth.per_ray_ids(rays.shape[0]).to(dev)
    codes = sc.code_library({"instance_ids": ids})["embedding_instance"].detach()
    grid, pc, pf = oracle_state(sc, dev)
    kw = dict(N_samples=S, N_importance=I, perturb=0, noise_std=0, embedding_instance=codes, frustum_bound_th=0.025, is_eval=True)
    with torch.no_grad():
        t_hip = timed(lambda: A.render_rays(sc.models, sc.embeddings, rays, chunk=1 << 30, **kw))
        t_ref = timed(lambda: on_gpu(pc, pf, grid, rays, **kw))
        a = A.render_rays(sc.models, sc.embeddings, rays, **kw)["rgb_fine"]
        b = on_gpu(pc, pf, grid, rays, **kw)["rgb_fine"]
    mse = ((a.double() - b.double()) ** 2).mean().item()
    n = rays.shape[0]
    print("inference, %d rays x (%d + %d) samples, both branches, voxel embedding, fp32:" % (n, S, S + I))
    print("  HIP path            %8.2f ms  %7.2f M ray-samples/s" % (t_hip * 1e3, n * evals / t_hip / 1e6))
    print("  PyTorch-ROCm path   %8.2f ms  %7.2f M ray-samples/s   (x%.1f)" % (t_ref * 1e3, n * evals / t_ref / 1e6, t_ref / t_hip))
    print("  PSNR(HIP, PyTorch-ROCm) rgb_fine = %.1f dB" % (-10 * torch.log10(torch.tensor(max(mse, 1e-30))).item()))

    # ---- training step: train.py:147-180 batch shape ----
    sc = synth.build_scene(A, True, preset=synth.SCANNET_LIKE, max_voxels=800_000, device=dev)
    rays_all = synth.camera_rays(640, 480).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    rays = rays_all[torch.randint(0, rays_all.shape[0], (n_train,), device=dev, generator=g)].contiguous()
    ids = synth.per_ray_ids(n_train).to(dev)
    target = torch.rand(n_train, 3, device=dev, generator=g)
    ptm = (ids == 1).view(-1, 1)
    rnd = {"perturb_rand": torch.rand(n_train, S, device=dev), "u_rand": torch.rand(n_train, I, device=dev),
           "noise": [torch.randn(n_train, S, device=dev), torch.randn(n_train, S, device=dev),
                     torch.randn(n_train, S + I, device=dev), torch.randn(n_train, S + I, device=dev)]}

    def loss_of(r):
        return sum(((r["rgb_%s" % t] - target) ** 2).mean() + ((r["rgb_instance_%s" % t] - target) ** 2).mean()
                   + 0.1 * (r["depth_%s" % t] ** 2).mean() + (r["opacity_instance_%s" % t] ** 2).mean() for t in ("coarse", "fine"))

    params = [p for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"]) for p in m.parameters()]

    def hip_step():
        for p in params:
            p.grad = None
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        r = A.render_rays(sc.models, sc.embeddings, rays, N_samples=S, N_importance=I, perturb=1.0, noise_std=1.0,
                          embedding_instance=codes, frustum_bound_th=0.025, pass_through_mask=ptm)
        loss_of(r).backward()

    grid, pc, pf = oracle_state(sc, dev)
    table = grid["table"].clone().requires_grad_(True)
    grid["table"] = table
    pc = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
    code_table = sc.code_library.embedding_instance.weight.detach().clone().requires_grad_(True)
    leaves = [table, code_table] + list(pc.values()) + list(pf.values())

    def ref_step():
        for p in leaves:
            p.grad = None
        codes = code_table[ids.squeeze()]
        r = on_gpu(pc, pf, grid, rays, N_samples=S, N_importance=I, perturb=1.0, noise_std=1.0,
                          embedding_instance=codes, frustum_bound_th=0.025, pass_through_mask=ptm, randoms=rnd)
        loss_of(r).backward()

    t_hip = timed(hip_step, 5)
    t_ref = timed(ref_step, 5)
    print("training step (forward + backward), %d rays x (%d + %d), perturb/noise on:" % (n_train, S, S + I))
    print("  HIP path            %8.2f ms  %7.2f M ray-samples/s" % (t_hip * 1e3, n_train * evals / t_hip / 1e6))
    print("  PyTorch-ROCm path   %8.2f ms  %7.2f M ray-samples/s   (x%.1f)" % (t_ref * 1e3, n_train * evals / t_ref / 1e6, t_ref / t_hip))


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:3]])
