#!/usr/bin/env python
"""Diagnostic (GPU box): per-key errors of render_rays_multi on the bench edit-demo golden and the neighbourhood of the
worst weights_fine entry.  TEST INFRASTRUCTURE."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, helpers as H
import object_nerf_amd as A
from object_nerf_amd.multi_rendering import render_rays_multi
DEV = "cuda"
for mode in ("f32", "bf16x3"):
    os.environ["OBJNERF_MFMA"] = mode
    g = cases.load_golden("multi_bench_edit_demo")
    bm = cases.BENCH_MULTI
    sc = cases.scene_for(A, "scannet_800k", device=DEV)
    _, _, box = cases.bench_multi_geometry()
    with torch.no_grad():
        r = render_rays_multi(sc.models, sc.embeddings, sc.code_library, [g["_rays_%d" % k].to(DEV) for k in range(3)],
                              bm["obj_ids"], N_samples=bm["N_samples"], N_importance=bm["N_importance"], perturb=0,
                              noise_std=0, background_skip_bbox={4: box})
    print("==", mode)
    for k in g:
        if k.startswith("_") or k == "obj_ids_coarse":
            continue
        print("  %-16s %.3e" % (k, H.normwise(r[k], g[k])))
    d = (r["weights_fine"].cpu() - g["weights_fine"]).abs()
    ray = int(d.max(-1)[0].argmax()); j = int(d[ray].argmax())
    lo, hi = max(j - 4, 0), j + 5
    print("  worst ray", ray, "pos", j, "of", d.shape[1])
    print("   z  ours", r["z_vals_fine"][ray, lo:hi].cpu().tolist())
    print("   z  ref ", g["z_vals_fine"][ray, lo:hi].tolist())
    print("   w  ours", r["weights_fine"][ray, lo:hi].cpu().tolist())
    print("   w  ref ", g["weights_fine"][ray, lo:hi].tolist())
    so, sr = torch.sort(r["weights_fine"].cpu(), -1)[0], torch.sort(g["weights_fine"], -1)[0]
    print("  sorted-weights multiset error %.3e" % ((so - sr).abs().max() / sr.abs().max()).item())
    print("  rays with |dw|>1e-3*max:", int((d.max(-1)[0] > 1e-3 * g["weights_fine"].abs().max()).sum()), "of", d.shape[0])
