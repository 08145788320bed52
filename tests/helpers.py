"""Shared test plumbing: scene construction for the drop-in types and the oracle's view of it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402


def state(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def oracle_grid(ev, keep_graph=False):
    """plain-tensor view of an EmbeddingVoxel for oracle.objnerf_oracle (keep_graph: the table stays the Parameter, so
    that autograd through the oracle reaches it)"""
    table = ev.embedding_space_ftr.weight if keep_graph else ev.embedding_space_ftr.weight.detach().cpu()
    return dict(voxel_idx_map=ev.voxel_idx_map.cpu(), table=table,
                voxel_offset=ev.voxel_offset.cpu(), voxel_size=ev.voxel_size.cpu(), voxel_shape=ev.voxel_shape.cpu())


def box_dict_from_helper(b):
    """synth.oriented_box-style dict of a reference BBoxRayHelper-shaped object (utils/bbox_utils.py:9-34 attributes)"""
    import numpy as np
    pa, aa = np.asarray(b.pose_avg, dtype=np.float64).squeeze(), np.asarray(b.axis_align_mat, dtype=np.float64)
    return dict(scale_factor=float(b.scale_factor), R_avg=pa[:3, :3], t_avg=pa[:3, 3], R_box=aa[:3, :3], t_box=aa[:3, 3],
                bmin=np.asarray(b.bbox_bounds[0], dtype=np.float64), bmax=np.asarray(b.bbox_bounds[1], dtype=np.float64))


def test_rays(n=96, w=64, h=48, stride=29, **kw):
    rays_all = synth.camera_rays(w, h, **kw)
    idx = torch.arange(0, rays_all.shape[0], stride)[:n]
    return rays_all[idx].contiguous()


def scene(use_voxel=True, device="cpu", max_voxels=120000, n_points=200_000, preset=synth.SCANNET_LIKE):
    return synth.build_scene(A, use_voxel, preset=preset, max_voxels=max_voxels, n_points=n_points, device=device)


def normwise(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def cdf_position(bins, weights, z, eps=1e-5):
    """F(z): position of samples z (N,I) on the float64 piecewise-linear CDF that sample_pdf inverts
    (bins (N,nb), weights (N,nb-1)); used to grade samplers in the well-conditioned domain."""
    bins, w, z = bins.double(), weights.double() + eps, z.double()
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    idx = (torch.searchsorted(bins.contiguous(), z.contiguous(), right=True) - 1).clamp(0, bins.shape[1] - 2)
    b0, b1 = torch.gather(bins, 1, idx), torch.gather(bins, 1, idx + 1)
    c0, c1 = torch.gather(cdf, 1, idx), torch.gather(cdf, 1, idx + 1)
    t = ((z - b0) / (b1 - b0).clamp_min(1e-300)).clamp(0, 1)
    return c0 + t * (c1 - c0)


def sampler_residual(bins, weights, u, z, eps=1e-5):
    """Per-sample grade of an inverse-CDF sampler, robust to both kinds of ill-conditioning:
    flat CDF (tiny pdf: z is sensitive, F(z) is not) and steep CDF (narrow heavy bin: F(z) is
    sensitive to one ulp of z, z is not).  Returns min(|F64(z) - u|, |z - z64| / max|bins|)."""
    from oracle import objnerf_oracle as O
    q = cdf_position(bins, weights, z, eps)
    z64 = O.sample_pdf(bins.double(), weights.double(), u.shape[1], det=False, eps=eps, u=u.double())
    rz = (z.double() - z64).abs() / bins.double().abs().max()
    return torch.minimum((q - u.double()).abs(), rz)


def fine_pass_on_reference_depths(sc, case, g, device="cuda"):
    """Teacher forcing: the fine MLP + compositing (stage entry points of the C ABI) evaluated on the REFERENCE's fine
    depths g["z_vals_fine"] of render case `case`.  Returns {golden key: tensor} for the seven fine-pass maps."""
    import ctypes as C
    import cases
    from object_nerf_amd import _lib
    from object_nerf_amd.rendering import mfma_mode
    c = cases.RENDER_CASES[case]
    use_voxel = cases.SCENES[c["scene"]][0]
    rays, ids, ptm, _ = cases.render_inputs(case)
    kw = c["kw"]
    n, S = g["z_vals_fine"].shape
    l = _lib.lib()
    with torch.no_grad():
        codes = sc.code_library({"instance_ids": ids.to(device)})["embedding_instance"].contiguous()
    z = g["z_vals_fine"].to(device).contiguous()
    rays_d = rays.to(device)
    b3 = mfma_mode() == "bf16x3"
    blob, aux = sc.models["fine"].packed(split_bf16=b3)
    buf = {k: torch.empty(n, S, *sh, device=device) for k, sh in dict(sigma=(), rgb=(3,), isig=(), irgb=(3,)).items()}
    a = _lib.MlpArgs()
    a.use_voxel, a.do_scene, a.do_object, a.mfma_bf16x3 = int(use_voxel), 1, 1, int(b3)
    a.blob, a.aux = blob.data_ptr(), aux.data_ptr()
    a.rays, a.z_vals, a.n_rays, a.S = rays_d.data_ptr(), z.data_ptr(), n, S
    a.codes, a.code_stride = codes.data_ptr(), 64
    if use_voxel:
        a.grid = sc.embeddings["xyz"].grid_struct()
    a.sigma, a.rgb, a.inst_sigma, a.inst_rgb = (buf[k].data_ptr() for k in ("sigma", "rgb", "isig", "irgb"))
    _lib.check(l.objnerf_mlp_eval(C.byref(a), _lib.stream_ptr()), "mlp_eval")
    out = {k: torch.empty(n, *sh, device=device) for k, sh in dict(weights=(S,), opacity=(), rgb_map=(3,), depth=(),
                                                                    rgb_inst=(3,), depth_inst=(), opacity_inst=()).items()}
    ca = _lib.CompositeArgs()
    ca.n_rays, ca.S, ca.z_vals = n, S, z.data_ptr()
    ca.sigma, ca.rgb, ca.inst_sigma, ca.inst_rgb = (buf[k].data_ptr() for k in ("sigma", "rgb", "isig", "irgb"))
    ca.white_back = int(kw.get("white_back", False))
    ca.occlusion = int((not kw.get("is_eval", False)) and kw.get("frustum_bound_th", 0) > 0)
    ca.frustum_bound_th = kw.get("frustum_bound_th", 0.0)
    ptm8 = ptm.reshape(-1).to(torch.uint8).to(device) if ptm is not None else None
    if ptm8 is not None:
        ca.pass_through_mask = ptm8.data_ptr()
    ca.rays_in_bbox = int(kw.get("rays_in_bbox", False))
    for k, t in out.items():
        setattr(ca, k, t.data_ptr())
    _lib.check(l.objnerf_composite(C.byref(ca), _lib.stream_ptr()), "composite")
    torch.cuda.synchronize()
    names = dict(weights="weights_fine", opacity="opacity_fine", rgb_map="rgb_fine", depth="depth_fine",
                 rgb_inst="rgb_instance_fine", depth_inst="depth_instance_fine", opacity_inst="opacity_instance_fine")
    return {gk: out[k] for k, gk in names.items()}


def moved_rays(z_ours, z_ref, z_coarse, frac=0.25):
    """Per-ray mask: some fine depth differs from the reference's by more than `frac` of the ray's smallest coarse
    spacing -- an importance sample that landed in a different place of its bin or in another bin (the discontinuities
    of rendering.py:43-60: the searchsorted decision, the denom < eps branch, and u = 1 against cdf[-1]).
    Everything on such a ray that is indexed by sample position (weights_, z_vals_) is then shifted, not perturbed."""
    zo, zr, zc = z_ours.detach().cpu().double(), z_ref.detach().cpu().double(), z_coarse.detach().cpu().double()
    gap = (zc[:, 1:] - zc[:, :-1]).abs()
    gap = torch.where(gap > 0, gap, torch.full_like(gap, float("inf"))).min(-1)[0]
    gap = torch.where(torch.isfinite(gap), gap, torch.ones_like(gap))
    return (zo - zr).abs().max(-1)[0] > frac * gap


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


import pytest  # noqa: E402

MFMA_MODES = ["f32", "bf16x3"]


@pytest.fixture(autouse=True, params=MFMA_MODES)
def mfma_mode(request, monkeypatch):
    """GPU test modules import this autouse fixture: every test in them runs once per arithmetic mode of the fused MLP
    kernels (OBJNERF_MFMA: fp32 MFMA, and the opt-in split-bf16 mode), same tolerances.  Tests that never reach those
    kernels are marked `single_mode` and run once."""
    if request.node.get_closest_marker("single_mode") is not None and request.param != "f32":
        pytest.skip("mode-independent test")
    monkeypatch.setenv("OBJNERF_MFMA", request.param)
    return request.param
