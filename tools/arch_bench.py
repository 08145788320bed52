#!/usr/bin/env python
"""Throughput of the layer-wise path (object_nerf_amd/generic.py, csrc/generic.hip: any config.model architecture, stage by stage
through the C ABI with the intermediate tensors in memory) next to the fused kernels, on a 320x240 frame (64 + 64, scene +
object): the default architecture through both paths (OBJNERF_PATH=layerwise), and the two non-default shapes of tests/cases.py.
usage: python tools/arch_bench.py [out.md]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402
import cases  # noqa: E402

DEV = "cuda"


def rate(sc, env):
    old = os.environ.get("OBJNERF_PATH")
    if env:
        os.environ["OBJNERF_PATH"] = env
    try:
        rays = synth.camera_rays(320, 240).to(DEV)
        n = rays.shape[0]
        with torch.no_grad():
            codes = sc.code_library({"instance_ids": synth.per_ray_ids(n).to(DEV)})["embedding_instance"]
            kw = dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, embedding_instance=codes, is_eval=True)
            A.render_rays(sc.models, sc.embeddings, rays, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                A.render_rays(sc.models, sc.embeddings, rays, **kw)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        return n * 192 / dt / 1e6, dt * 1e3
    finally:
        if env:
            os.environ.pop("OBJNERF_PATH")
            if old is not None:
                os.environ["OBJNERF_PATH"] = old


def train_ms(sc, n_rays=2048, steps=6):
    """one training step of the reference's batch shape (2048 rays, 64 + 64, perturb / noise on) on a scene's architecture: forward +
    backward (no optimizer), ms"""
    rays_all = synth.camera_rays(320, 240).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    params = [p for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"]) for p in m.parameters()]
    target = torch.rand(n_rays, 3, device=DEV, generator=g)
    ids = synth.per_ray_ids(n_rays).to(DEV)

    def step():
        for p in params:
            p.grad = None
        rays = rays_all[torch.randint(0, rays_all.shape[0], (n_rays,), device=DEV, generator=g)].contiguous()
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        r = A.render_rays(sc.models, sc.embeddings, rays, N_samples=64, N_importance=64, perturb=1.0, noise_std=1.0,
                          embedding_instance=codes, frustum_bound_th=0.025)
        mse = torch.nn.functional.mse_loss
        sum(mse(r["rgb_%s" % t], target) + mse(r["rgb_instance_%s" % t], target) for t in ("coarse", "fine")).backward()
    step(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main(out=None):
    lines = ["| architecture | path | ms per 320x240 frame | M ray-samples/s |", "|---|---|---|---|"]
    sc = cases.scene_for(A, "voxel", device=DEV)
    for name, env in (("fused kernels", None), ("layer-wise", "layerwise")):
        r, ms = rate(sc, env)
        lines.append("| default (D 8, W 256, 4 x 128 object branch, voxel 16+8) | %s | %.1f | %.1f |" % (name, ms, r))
    for a in sorted(cases.ARCH_SCENES):
        sc = cases.scene_for(A, a, device=DEV)
        r, ms = rate(sc, None)
        lines.append("| %s %s | layer-wise | %.1f | %.1f |" % (a, cases.ARCH_SCENES[a][2], ms, r))
    if os.environ.get("ARCH_BENCH_TRAIN", "1") != "0":
        lines += ["", "| architecture | training step (2048 rays x (64 + 64), forward + backward), ms |", "|---|---|"]
        for a in sorted(cases.ARCH_SCENES):
            sc = cases.scene_for(A, a, device=DEV)
            lines.append("| %s | %.2f |" % (a, train_ms(sc)))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write("# the layer-wise path for non-default architectures (tools/arch_bench.py)\n\n" + txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
