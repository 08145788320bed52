#!/usr/bin/env python
"""Diagnostic (GPU box): parameter-gradient error of the HIP training path against the oracle's autograd in float64, next to
the float32 oracle's own distance from float64 (the fp32 noise floor of each gradient), as the batch grows.
usage: python tools/grad_floor.py [n_rays ...]      TEST INFRASTRUCTURE"""
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


def main(sizes):
    sc = cases.scene_for(A, "voxel", device=DEV)
    S, I = 64, 64
    kw = dict(N_samples=S, N_importance=I, perturb=1.0, noise_std=1.0, is_eval=False, frustum_bound_th=0.025)
    if os.environ.get("GRAD_FLOOR_SMOOTH"):     # no occlusion mask, no density noise: the only decisions left are the (Leaky)ReLUs
        kw.update(noise_std=0.0, frustum_bound_th=-1.0)
    for n in sizes:
        rays = H.test_rays(n, w=256, h=192, stride=23)
        ids = synth.per_ray_ids(n, seed=5)
        ptm = (torch.arange(n) % 3 == 0).view(n, 1)
        g = torch.Generator().manual_seed(2)
        rnd = dict(perturb_rand=torch.rand(n, S, generator=g), u_rand=torch.rand(n, I, generator=g),
                   noise=[torch.randn(n, S, generator=g), torch.randn(n, S, generator=g),
                          torch.randn(n, S + I, generator=g), torch.randn(n, S + I, generator=g)])
        for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"]):
            m.zero_grad()
        codes = sc.code_library({"instance_ids": ids.to(DEV)})["embedding_instance"]
        rd = dict(perturb_rand=rnd["perturb_rand"].to(DEV), u_rand=rnd["u_rand"].to(DEV), noise=[t.to(DEV) for t in rnd["noise"]])
        res = A.render_rays(sc.models, sc.embeddings, rays.to(DEV), embedding_instance=codes, pass_through_mask=ptm.to(DEV), _randoms=rd, **kw)
        T._loss(res).backward()
        zf = res["z_vals_fine"].detach().cpu()

        def oracle(dt):
            cv = lambda t: t.detach().cpu().to(dt) if t.is_floating_point() else t.detach().cpu()   # noqa: E731
            pc = {k: cv(v).clone().requires_grad_(v.is_floating_point()) for k, v in sc.models["coarse"].state_dict().items()}
            pf = {k: cv(v).clone().requires_grad_(v.is_floating_point()) for k, v in sc.models["fine"].state_dict().items()}
            ctab = cv(sc.code_library.embedding_instance.weight).clone().requires_grad_(True)
            grid = {k: cv(v) for k, v in H.oracle_grid(sc.embeddings["xyz"]).items()}
            grid["table"] = grid["table"].clone().requires_grad_(True)
            r = {k: ([x.to(dt) for x in v] if isinstance(v, list) else v.to(dt)) for k, v in rnd.items()}
            old = torch.get_default_dtype()
            torch.set_default_dtype(dt)
            try:
                ro = O.render_rays(pc, pf, grid, rays.to(dt), embedding_instance=ctab[ids], pass_through_mask=ptm, randoms=r,
                                   z_fine_override=zf.to(dt), **kw)
                tot = 0.0
                gg = torch.Generator().manual_seed(0)
                for k in sorted(ro):
                    if k.startswith(("weights_", "z_vals_")):
                        continue
                    tot = tot + (ro[k] * torch.randn(ro[k].shape, generator=gg, dtype=torch.float32).to(dt)).sum()
                tot.backward()
            finally:
                torch.set_default_dtype(old)
            out = {"coarse." + k: v.grad for k, v in pc.items() if v.grad is not None}
            out.update({"fine." + k: v.grad for k, v in pf.items() if v.grad is not None})
            out["codes"], out["table"] = ctab.grad, grid["table"].grad
            return out
        o32 = oracle(torch.float32)
        o64 = oracle(torch.float64) if not os.environ.get("GRAD_FLOOR_NO64") else o32
        mine = {"coarse." + k: p.grad for k, p in sc.models["coarse"].named_parameters()}
        mine.update({"fine." + k: p.grad for k, p in sc.models["fine"].named_parameters()})
        mine["codes"], mine["table"] = sc.code_library.embedding_instance.weight.grad, sc.embeddings["xyz"].embedding_space_ftr.weight.grad
        print("== n = %d rays (%d points)" % (n, n * (2 * S + I)))
        rows = []
        for k in o64:
            rows.append((k, H.rel_l2(mine[k], o64[k]), H.rel_l2(o32[k], o64[k]), H.rel_l2(mine[k], o32[k])))
        rows.sort(key=lambda r: -r[1])
        for k, a, b, c in rows[:10]:
            print("  %-42s hip-vs-f64 %.2e   oracle32-vs-f64 %.2e   hip-vs-oracle32 %.2e" % (k, a, b, c))
        print("  worst ratio (hip-vs-f64) / (oracle32-vs-f64): %.2f" % max(a / max(b, 1e-30) for _, a, b, _ in rows))


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [24, 512, 2048])
