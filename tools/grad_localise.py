#!/usr/bin/env python
"""Diagnostic (GPU box): which rays carry the gradient discrepancy between the HIP training path and the fp32 oracle.
TEST INFRASTRUCTURE"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, helpers as H  # noqa: E402
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402
from oracle import objnerf_oracle as O  # noqa: E402
import test_gpu_train as T  # noqa: E402

DEV = "cuda"
KEY = "xyz_encoding_1.0.weight"


def grads(sc, rays, ids, kw):
    n = rays.shape[0]
    for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"]):
        m.zero_grad()
    codes = sc.code_library({"instance_ids": ids.to(DEV)})["embedding_instance"]
    res = A.render_rays(sc.models, sc.embeddings, rays.to(DEV), embedding_instance=codes, **kw)
    T._loss(res).backward()
    zf = res["z_vals_fine"].detach().cpu()
    pc = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in sc.models["coarse"].state_dict().items()}
    pf = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in sc.models["fine"].state_dict().items()}
    ctab = sc.code_library.embedding_instance.weight.detach().cpu().clone()
    grid = H.oracle_grid(sc.embeddings["xyz"])
    ro = O.render_rays(pc, pf, grid, rays, embedding_instance=ctab[ids], z_fine_override=zf, **kw)
    T._loss(ro).backward()
    hip = dict(sc.models["fine"].named_parameters())[KEY].grad.detach().cpu().clone()
    return hip, pf[KEY].grad.clone(), res, ro


def main():
    sc = cases.scene_for(A, "voxel", device=DEV)
    S, I, n = 64, 64, 2048
    kw = dict(N_samples=S, N_importance=I, perturb=0.0, noise_std=0.0, is_eval=True, frustum_bound_th=-1.0)
    rays = H.test_rays(n, w=256, h=192, stride=23)
    ids = synth.per_ray_ids(n, seed=5)
    hip, orc, _, _ = grads(sc, rays, ids, kw)
    print("all %d rays: fine.%s rel L2 %.3e  (|g| %.3e)" % (n, KEY, H.rel_l2(hip, orc), orc.norm().item()))
    worst = (0.0, 0)
    for g0 in range(0, n, 128):
        sl = slice(g0, g0 + 128)
        hip, orc, _, _ = grads(sc, rays[sl], ids[sl], kw)
        e = (hip.double() - orc.double()).norm().item()
        print("  rays %4d..%4d  abs err %.3e  rel %.3e  |g| %.3e" % (g0, g0 + 127, e, H.rel_l2(hip, orc), orc.norm().item()))
        if e > worst[0]:
            worst = (e, g0)
    g0 = worst[1]
    print("worst group starts at", g0)
    per = []
    for r in range(g0, g0 + 128):
        hip, orc, res, ro = grads(sc, rays[r:r + 1], ids[r:r + 1], kw)
        per.append(((hip.double() - orc.double()).norm().item(), r, orc.norm().item()))
    per.sort(reverse=True)
    for e, r, gn in per[:5]:
        print("  ray %d abs err %.3e |g| %.3e" % (r, e, gn))
    r = per[0][1]
    hip, orc, res, ro = grads(sc, rays[r:r + 1], ids[r:r + 1], kw)
    w = ro["weights_fine"][0].detach()
    print("ray", r, "weights_fine top5:", torch.topk(w, 5))
    print("  hip weights at those:", res["weights_fine"][0].detach().cpu()[torch.topk(w, 5)[1]])
    # sigma pre-activation of the fine pass from the oracle's own evaluation
    grid = H.oracle_grid(sc.embeddings["xyz"])
    pf = H.state(sc.models["fine"])
    z = ro["z_vals_fine"].detach()
    xyz = rays[r:r + 1, None, 0:3] + rays[r:r + 1, None, 3:6] * z[..., None]
    ctab = sc.code_library.embedding_instance.weight.detach().cpu()
    sg, c, isg, ic = O.eval_points(pf, grid, xyz, rays[r:r + 1, 3:6], ctab[ids[r:r + 1]], True, True, 32768)
    print("  min |sigma_pre| scene %.3e at %d, inst %.3e" % (sg.abs().min().item(), int(sg.abs().argmin()), isg.abs().min().item()))
    idx = torch.topk(w, 5)[1]
    print("  sigma_pre at top weights:", sg[0][idx])


if __name__ == "__main__":
    main()
