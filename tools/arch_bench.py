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
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write("# the layer-wise path for non-default architectures (tools/arch_bench.py)\n\n" + txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
