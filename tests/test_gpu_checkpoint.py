"""GPU: a checkpoint written by the REFERENCE's own module types renders through the packed weight stream (SURVEY.md §8
row f3; the use case is render_tools/editable_renderer.py:75-79: `load_from_checkpoint(...).cuda().eval()` then render).

tests/golden/reference_small.ckpt is `{"state_dict": ObjectNeRFSystem.state_dict(), ...}` of the real train.py system
built from models/* (oracle/ref_callers.py `checkpoint`: 300-point cloud, 6,500-row table, seeded W1 weights), and
tests/golden/ckpt_render.npz what the reference rendered from that system -- `render_rays` (64 + 64, eval) and
`render_rays_multi` (ids [0, 4, 4], removed-object box).  Here: torch.load -> checkpoint.build_from_state_dict -> .cuda() ->
the HIP entry points on the same rays, graded like the other reference goldens; then export -> strict reload equality and a
bit-equal re-render from the exported dict."""
import os

import pytest
import torch

import cases
import helpers as H
import object_nerf_amd as A
from object_nerf_amd import checkpoint
from object_nerf_amd.multi_rendering import render_rays_multi

pytestmark = pytest.mark.gpu
DEV = "cuda"
CKPT = os.path.join(cases.GOLDEN_DIR, "reference_small.ckpt")
KW = dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, is_eval=True)


def load():
    ckpt = torch.load(CKPT, map_location="cpu", weights_only=True)      # tensors and plain containers only: no pickle code runs
    return ckpt, checkpoint.build_from_state_dict(ckpt, device=DEV)


def render_single(sc):
    rays, ids, _, _ = cases.render_inputs("voxel_eval")
    with torch.no_grad():
        codes = sc.code_library({"instance_ids": ids.to(DEV)})["embedding_instance"]
        return rays, codes, A.render_rays(sc.models, sc.embeddings, rays.to(DEV), embedding_instance=codes, chunk=32768, **KW)


def test_reference_checkpoint_renders_like_the_reference():
    ckpt, sc = load()
    assert all(p.is_cuda for m in sc.models.values() for p in m.parameters()) and sc.embeddings["xyz"].voxel_idx_map.is_cuda
    g = {k[len("single_"):]: v for k, v in cases.load_golden("ckpt_render").items() if k.startswith("single_")}
    rays, codes, out = render_single(sc)
    assert sorted(out) == sorted(g)
    f64 = H.oracle_f64(sc, True, rays, codes.cpu(), None, None, {k: v for k, v in KW.items() if k not in ("perturb", "noise_std")})
    for k in g:
        err, floor = H.normwise(out[k], g[k]), H.normwise(g[k], f64[k])
        tol = max(H.FLOOR_FACTOR * floor, 2e-5) if k.endswith("fine") else 1e-4
        assert err <= tol, "%s: normwise %.3e > %.3e (fp64 floor %.3e)" % (k, err, tol, floor)
    moved = int(H.moved_rays(out["z_vals_fine"], g["z_vals_fine"], g["z_vals_coarse"]).sum())
    moved64 = int(H.moved_rays(f64["z_vals_fine"], g["z_vals_fine"], g["z_vals_coarse"]).sum())
    assert moved <= moved64 + 1
    assert H.psnr(out["rgb_fine"], g["rgb_fine"]) >= 60.0


def test_reference_checkpoint_renders_multi_like_the_reference():
    _, sc = load()
    g = {k[len("multi_"):]: v for k, v in cases.load_golden("ckpt_render").items() if k.startswith("multi_")}
    sets, boxes = cases.multi_inputs()
    m = cases.MULTI
    with torch.no_grad():
        r = render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.to(DEV) for s in sets], m["obj_ids"],
                              N_samples=m["N_samples"], N_importance=m["N_importance"], perturb=0, noise_std=0,
                              background_skip_bbox={4: boxes[0]})
    assert sorted(r) == sorted(g)
    f64 = H.oracle_multi_f64(sc, sets, m["obj_ids"], boxes=[boxes[0]], N_samples=m["N_samples"], N_importance=m["N_importance"])
    H.grade_multi(r, g, "checkpoint / multi", f64, sets)
    assert H.psnr(r["rgb_fine"], g["rgb_fine"]) >= 60.0


def test_export_reloads_strictly_and_renders_bit_equal():
    ckpt, sc = load()
    sd = ckpt["state_dict"]
    exported = checkpoint.export_state_dict(sc)
    assert list(exported) == list(sd)                      # same keys in the LightningModule's order
    for k in sd:
        assert exported[k].dtype == sd[k].dtype and torch.equal(exported[k], sd[k]), k
    sc2 = checkpoint.build_from_state_dict({"state_dict": exported}, device=DEV)
    _, _, a = render_single(sc)
    _, _, b = render_single(sc2)
    for k in a:
        assert torch.equal(a[k], b[k]), k
