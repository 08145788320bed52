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
    blob, aux = sc.models["fine"].packed()
    buf = {k: torch.empty(n, S, *sh, device=device) for k, sh in dict(sigma=(), rgb=(3,), isig=(), irgb=(3,)).items()}
    a = _lib.MlpArgs()
    a.use_voxel, a.do_scene, a.do_object = int(use_voxel), 1, 1
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


def occlusion_flipped_rays(depth_ours, z_ours, depth_ref, z_ref, th):
    """Per-ray mask: the training-mode occlusion decision `depth + frustum_bound_th < z` (rendering.py:192-202) comes out
    differently for some sample of the ray under OUR depth map than under the reference's.  The depth maps themselves are graded
    against their own tolerance; where two maps inside that tolerance straddle a sample depth, that sample's instance weight is
    switched on in one render and off in the other -- a whole-sample discontinuity of the reference's algorithm (like a moved
    importance sample, helpers.moved_rays), not an arithmetic error of either.  Such rays are left out of the instance keys of that
    pass (round 6: the regenerated caller goldens put a sample of one of 40 rays 7.6e-5 from its threshold)."""
    do, zo = depth_ours.detach().cpu().double().reshape(-1, 1), z_ours.detach().cpu().double()
    dr, zr = depth_ref.detach().cpu().double().reshape(-1, 1), z_ref.detach().cpu().double()
    return (((do + th) < zo) != ((dr + th) < zr)).any(-1)


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# ---- image-scale cases (cases.FRAME_CASES): the HIP path on a whole 160x120 frame against the reference's frame ----------
def render_frame_case(case, sc, device="cuda", device_rays=False):
    """-> dict of result tensors of the product on the frame case's inputs (single: render_rays, multi: render_rays_multi
    on the ray sets of the oracle's generate_rays -- bit-equal to the reference's ray / box code, asserted when the golden
    was made -- or, device_rays=True, on the sets written by objnerf_generate_rays)"""
    import cases
    from object_nerf_amd.multi_rendering import render_rays_multi
    from oracle import objnerf_oracle as O
    fc = cases.FRAME_CASES[case]
    with torch.no_grad():
        if fc["kind"] == "single":
            rays, ids, kw, _ = cases.frame_inputs(case)
            codes = sc.code_library({"instance_ids": ids.to(device)})["embedding_instance"]
            return A.render_rays(sc.models, sc.embeddings, rays.to(device), embedding_instance=codes, chunk=32768, **kw)
        if device_rays:
            from object_nerf_amd.ray_utils import generate_rays
            sets = cases.frame_multi_sets(generate_rays, case)
        else:
            # the very sets the reference rendered the golden frame from (regenerating them through the reference's fp32 matmul +
            # norm on THIS host may give directions an ulp apart: they round differently on different CPUs)
            sets = [s.to(device) for s in cases.golden_multi_sets(case)]
        bm = cases.BENCH_MULTI
        box = cases.frame_inputs(case)[2]
        r = render_rays_multi(sc.models, sc.embeddings, sc.code_library, sets, bm["obj_ids"], N_samples=bm["N_samples"],
                              N_importance=bm["N_importance"], perturb=0, noise_std=0, background_skip_bbox={4: box})
        r["_sets"] = sets
        return r


def psnr(a, b):
    return (-10.0 * torch.log10(((a.detach().cpu().double() - b.detach().cpu().double()) ** 2).mean().clamp_min(1e-30))).item()


def frame_report(case, out, g):
    """Per-map distances of a rendered frame from the reference's frame: {key: (err, floor, err_l2, floor_l2)}, the
    moved-ray count on the stored subset, the reference-vs-float64 count, PSNR(ours, reference) and the PSNR difference
    against a fixed synthetic target (utils/metrics.py:5-15, over the whole frame)."""
    import cases
    keys = sorted(k for k in g if not k.startswith("_"))
    rows = {k: (normwise(out[k], g[k]), float(g["_floor_" + k]), rel_l2(out[k], g[k]), float(g["_floor_l2_" + k])) for k in keys}
    sub = g["_sub"]
    zc = out["z_vals_coarse"][sub.to(out["z_vals_coarse"].device)]
    if cases.FRAME_CASES[case]["kind"] == "multi":
        # the joint (K*S) coarse depths interleave the sets: the spacing that matters is a set's own, take the background's
        zc = out["_sets"][0][sub.to(zc.device), 6:8]
        zc = zc[:, :1] + (zc[:, 1:] - zc[:, :1]) * torch.linspace(0, 1, cases.BENCH_MULTI["N_samples"], device=zc.device)
    moved = int(moved_rays(out["z_vals_fine"][sub.to(zc.device)], g["_z_vals_fine_sub"], zc).sum())
    target = torch.rand(g["rgb_fine"].shape, generator=torch.Generator().manual_seed(3))
    return dict(rows=rows, moved=moved, moved64=int(g["_moved64"]), n_sub=int(sub.numel()),
                psnr=psnr(out["rgb_fine"], g["rgb_fine"]), psnr64=float(g["_psnr64"]),
                dpsnr=abs(psnr(out["rgb_fine"], target) - psnr(g["rgb_fine"], target)))


FLOOR_FACTOR = 3.0      # end-to-end fine-pass tolerance in units of the reference's own fp32-vs-fp64 distance


def oracle_multi_f64(sc, sets, obj_ids, boxes=None, randoms=None, arch=None, **kw):
    """oracle.render_rays_multi in float64 on the same inputs (the reference's arithmetic carried out exactly enough):
    its distance from the reference's fp32 result is the yardstick of the fine-pass keys"""
    from oracle import objnerf_oracle as O
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        dbl = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}   # noqa: E731
        rnd = None
        if randoms:
            rnd = dict(u_rand=[t.double() for t in randoms["u_rand"]], noise=[t.double() for t in randoms["noise"]])
        with torch.no_grad():
            return O.render_rays_multi(dbl(state(sc.models["coarse"])), dbl(state(sc.models["fine"])),
                                       dbl(oracle_grid(sc.embeddings["xyz"])),
                                       sc.code_library.embedding_instance.weight.detach().cpu().double(),
                                       [s.detach().cpu().double() for s in sets], list(obj_ids), skip_boxes=boxes, randoms=rnd, arch=arch, **kw)
    finally:
        torch.set_default_dtype(old)


def grade_multi(r, g, what, f64=None, sets=None, n_samples=64):
    """render_rays_multi against the reference, graded like the single-ray-set path (round 4; until round 3: flat 5e-3 on
    "settled" rays, 2e-2 on pixels, 10 % unsettled rays allowed).  Coarse keys: 1e-4.  Fine keys: FLOOR_FACTOR x the
    reference's own fp32-vs-fp64 distance on that key (f64 = oracle_multi_f64 on the same inputs; never below 2e-5), and no
    more rays with moved importance samples than under the float64 oracle, + 1 (moved = a fine depth further than a
    quarter of the ray's smallest own-set coarse spacing from the reference's: sets = the ray sets, for those spacings)."""
    assert f64 is not None, "grade_multi needs the float64 oracle's result (helpers.oracle_multi_f64)"
    report = []
    for k in g:
        if k.startswith("_"):
            continue
        if k == "obj_ids_coarse":
            nz = g["z_vals_coarse"] != 0          # tie order at z == 0 is unspecified in the reference
            assert torch.equal(r[k].cpu()[nz], g[k][nz])
            continue
        err = normwise(r[k], g[k])
        floor = normwise(g[k], f64[k])
        if k.endswith("coarse"):      # the BASELINE contract's 1e-4, or the reference's own floor where that is larger (a sample
            # on a face of the removed object's box flips between its sigma and -1e5 within fp32 roundoff, multi_rendering.py:239-241)
            assert err <= max(1e-4, FLOOR_FACTOR * floor), "%s/%s %.3e (fp64 floor %.3e)" % (what, k, err, floor)
            continue
        tol = max(FLOOR_FACTOR * floor, 2e-5)
        report.append("%s %.1e (floor %.1e)" % (k, err, floor))
        assert err <= tol, "%s/%s: normwise %.3e > %.1f x fp64 floor %.3e" % (what, k, err, FLOOR_FACTOR, floor)
    if "z_vals_fine" in g and sets is not None:
        # smallest coarse spacing over the sets that hit (rays that missed a box have near = far = 0)
        gaps = []
        for s_ in sets:
            s_ = s_.detach().cpu().double()
            gap = (s_[:, 7] - s_[:, 6]) / (n_samples - 1)
            gaps.append(torch.where(gap > 0, gap, torch.full_like(gap, float("inf"))))
        gap = torch.stack(gaps).min(0)[0]
        gap = torch.where(torch.isfinite(gap), gap, torch.ones_like(gap))

        def moved(z):
            return int(((z.detach().cpu().double() - g["z_vals_fine"].double()).abs().max(-1)[0] > 0.25 * gap).sum())
        mo, m64 = moved(r["z_vals_fine"]), moved(f64["z_vals_fine"])
        report.append("moved %d (fp64 oracle %d)" % (mo, m64))
        assert mo <= m64 + 1, "%s: %d rays' importance samples moved (float64 oracle: %d)" % (what, mo, m64)
    print(what, "; ".join(report))


def oracle_f64(sc, use_voxel, rays, codes, ptm, randoms, kw, arch=None):
    """the oracle in float64 on the same inputs -> fp32-vs-fp64 noise floor per key"""
    from oracle import objnerf_oracle as O
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        dbl = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}   # noqa: E731
        grid = dbl(oracle_grid(sc.embeddings["xyz"])) if use_voxel else None
        rnd = None
        if randoms:
            rnd = dict(perturb_rand=randoms["perturb_rand"].double(), u_rand=randoms["u_rand"].double(),
                       noise=[t.double() for t in randoms["noise"]])
        with torch.no_grad():
            return O.render_rays(dbl(state(sc.models["coarse"])), dbl(state(sc.models["fine"])), grid, rays.double(),
                                 embedding_instance=codes.double(), pass_through_mask=ptm, randoms=rnd, arch=arch, **kw)
    finally:
        torch.set_default_dtype(old)
