"""GPU: every HIP stage, teacher-forced (identical inputs), through the C ABI, against the committed
reference outputs (tests/golden/) and the oracle.  Contract (BASELINE.json): <= 1e-4 relative fp32
per stage; the assertions below are tighter (what the kernels actually achieve) so regressions show."""
import ctypes as C

import pytest
import torch

import cases
import helpers as H
import object_nerf_amd as A
from object_nerf_amd import _lib, synth
from object_nerf_amd.bbox import check_in_any_boxes
from oracle import objnerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
_scenes = {}


def scene(name):
    if name not in _scenes:
        _scenes[name] = cases.scene_for(A, name, device=DEV)
    return _scenes[name]


def check(a, b, tol, what):
    err = H.normwise(a, b)
    assert err <= tol, "%s: normwise error %.3e > %.1e" % (what, err, tol)
    return err


def test_pos_encode_matches_reference():
    g = cases.load_golden("stage_pe")
    for k, (x, nf) in cases.pe_inputs().items():
        with torch.no_grad():
            out = A.Embedding(x.shape[1], nf)(x.to(DEV))
        assert out.shape == g[k].shape
        # sin/cos of arguments up to 2^9 * 4.5: abs error ~1e-7 (output bounded by 1)
        assert (out.cpu() - g[k]).abs().max().item() < 2e-6, k


def test_sincos_large_and_special_arguments():
    x = torch.tensor([[0.0, -0.0, 1e-30], [3.14159265, -3.14159265, 1.5707963], [1000.5, -2047.75, 30000.0],
                      [65535.0, 1e6, -3.3e8]])
    with torch.no_grad():
        out = A.Embedding(3, 2)(x.to(DEV)).cpu()
    ref = O.pos_encode(x.double(), 2).float()
    # Cody-Waite path below 65536, OCML Payne-Hanek path above: f32-roundoff class everywhere
    assert (out - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("sname", ["voxel", "sparse"])
def test_voxel_embed_matches_reference(sname):
    g = cases.load_golden("stage_voxel_embed_" + sname)
    ev = scene(sname).embeddings["xyz"]
    with torch.no_grad():
        s, o = ev(cases.voxel_points().to(DEV))
    # raw trilinear features (first 16 / 8 columns) and xyz are exact-order restatements
    check(s[:, :16], g["scene_ftr"][:, :16], 2e-6, "voxel raw scene")
    check(o[:, :8], g["obj_ftr"][:, :8], 2e-6, "voxel raw obj")
    # positional encodings: absolute error (values bounded by 1; 2^5 band amplifies input ulps)
    assert (s.cpu() - g["scene_ftr"])[:, 16:208].abs().max().item() < 2e-4
    assert (o.cpu() - g["obj_ftr"])[:, 8:].abs().max().item() < 2e-4
    # xyz encoding: |x| up to 1e3 at the 2^9 band (arguments 5e5) stays f32-roundoff class
    assert (s.cpu() - g["scene_ftr"])[:, 208:].abs().max().item() < 2e-6
    # out-of-grid points have all-zero voxel features
    assert s[4, :16].abs().max().item() == 0 and o[5, :8].abs().max().item() == 0


@pytest.mark.parametrize("sname", ["voxel", "plain"])
def test_mlp_branches_match_reference(sname):
    """ObjectNeRF.forward / forward_instance on identical pre-embedded inputs (memory-form kernel)"""
    g = cases.load_golden("stage_mlp_" + sname)
    m = scene(sname).models["coarse"]
    i = {k: (v.to(DEV) if v is not None else None) for k, v in cases.mlp_inputs(sname == "voxel").items()}
    with torch.no_grad():
        o = m({"emb_xyz": i["emb_xyz"], "emb_dir": i["emb_dir"]})
        oi = m.forward_instance(i)
        so = m({"emb_xyz": i["emb_xyz"]}, sigma_only=True)
        soi = m.forward_instance(i, sigma_only=True)
    # sigma_only launches the density-only kernel variant (skips final/dir/rgb layers): same values, bit for bit
    assert list(so) == ["sigma"] and list(soi) == ["inst_sigma"]
    assert torch.equal(so["sigma"], o["sigma"]) and torch.equal(soi["inst_sigma"], oi["inst_sigma"])
    assert o["sigma"].shape == (200, 1) and o["rgb"].shape == (200, 3)
    for a, k in ((o["sigma"], "sigma"), (o["rgb"], "rgb"), (oi["inst_sigma"], "inst_sigma"), (oi["inst_rgb"], "inst_rgb")):
        check(a, g[k], 1e-5, "mlp/%s/%s" % (sname, k))


@pytest.mark.parametrize("sname", ["voxel", "plain"])
def test_sigma_query_matches_the_reference_mesh_tool(sname):
    """SURVEY.md section 8 row f4 as ONE kernel: the density-grid query of tools/extract_mesh.py:62-113 on the 32^3 golden
    lattice (scene, and object 4 with its code) -- points embedded inside the MLP kernel, nothing but sigma written --
    against what the reference's embedding_xyz -> forward(sigma_only=True) chunk loop returned; and against the drop-in's
    own memory form of the same loop (embedding materialised, then the sigma-only kernel)."""
    g = cases.load_golden("stage_sigma_grid")
    sc = scene(sname)
    emb, fine = sc.embeddings["xyz"], sc.models["fine"]
    oid = cases.SIGMA_GRID["obj_id"]
    x, y, z = cases.sigma_grid_axes()
    with torch.no_grad():
        code = sc.code_library.embedding_instance.weight[oid]
        s0 = fine.query_sigma(emb, lattice=(x, y, z))
        s4 = fine.query_sigma(emb, lattice=(x, y, z), obj_code=code)
        assert s0.shape == s4.shape == (32 ** 3, 1)
        e0 = check(s0[:, 0], g["%s_obj0" % sname], 2e-5, "sigma query / scene / " + sname)
        e4 = check(s4[:, 0], g["%s_obj%d" % (sname, oid)], 2e-5, "sigma query / object / " + sname)
        print("sigma query %s: normwise error scene %.2e, object %.2e" % (sname, e0, e4))
        # the explicit-points form of the same lattice: bit-equal (same kernel, same tile walk)
        import numpy as np
        pts = torch.FloatTensor(np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3)).to(DEV)
        assert torch.equal(fine.query_sigma(emb, xyz=pts), s0)
        assert torch.equal(fine.query_sigma(emb, xyz=pts, obj_code=code.view(1, 64)), s4)
        # the script's own form on the drop-in types: embed, then the memory-form sigma-only kernel
        if sname == "voxel":
            ex, ov = emb(pts)
            m0 = fine({"emb_xyz": ex, "obj_voxel": ov}, sigma_only=True)["sigma"]
            m4 = fine.forward_instance({"emb_xyz": ex, "obj_voxel": ov, "obj_code": code.expand(pts.shape[0], 64).contiguous()},
                                       sigma_only=True)["inst_sigma"]
        else:
            ex = emb(pts)
            m0 = fine({"emb_xyz": ex}, sigma_only=True)["sigma"]
            m4 = fine.forward_instance({"emb_xyz": ex, "obj_code": code.expand(pts.shape[0], 64).contiguous()}, sigma_only=True)["inst_sigma"]
        check(s0, m0, 2e-5, "fused vs memory form / scene")
        check(s4, m4, 2e-5, "fused vs memory form / object")
    # the object query hoists its ONE code (a constant of the whole query) like the render path hoists per-ray terms:
    # same sums in another association; OBJNERF_HOIST=0 contracts the code per point
    import os
    old = os.environ.get("OBJNERF_HOIST")
    os.environ["OBJNERF_HOIST"] = "0"
    try:
        with torch.no_grad():
            s4_plain = fine.query_sigma(emb, lattice=(x, y, z), obj_code=code)
    finally:
        if old is None:
            os.environ.pop("OBJNERF_HOIST")
        else:
            os.environ["OBJNERF_HOIST"] = old
    assert not torch.equal(s4_plain, s4) and H.normwise(s4, s4_plain) < 2e-6
    check(s4_plain[:, 0], g["%s_obj%d" % (sname, oid)], 2e-5, "sigma query / object, code contracted per point / " + sname)


def test_sigma_query_lattice_order_tails_and_errors():
    """np.meshgrid's 'xy' order with nx != ny != nz, point counts that are not multiples of 128 (incl. 1 and 0), the oracle
    on the same lattice, and the argument checks of the new form"""
    sc = scene("voxel")
    emb, fine = sc.embeddings["xyz"], sc.models["fine"]
    import numpy as np
    x, y, z = cases.sigma_grid_axes((7, 5, 11))
    with torch.no_grad():
        s = fine.query_sigma(emb, lattice=(x, y, z))
        pts = torch.FloatTensor(np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3))
        assert s.shape == (7 * 5 * 11, 1) and torch.equal(fine.query_sigma(emb, xyz=pts.to(DEV)), s)
        ref = O.sigma_grid(H.state(fine), H.oracle_grid(emb), x, y, z)
        check(s, ref, 2e-5, "7 x 5 x 11 lattice vs oracle")
        for n in (1, 31, 129):
            assert torch.equal(fine.query_sigma(emb, xyz=pts[:n].to(DEV)), s[:n])
        assert fine.query_sigma(emb, xyz=pts[:0].to(DEV)).shape == (0, 1)
        with pytest.raises(ValueError):
            fine.query_sigma(emb)
        with pytest.raises(RuntimeError, match="ONE 64-d code"):
            fine.query_sigma(emb, xyz=pts.to(DEV), obj_code=torch.zeros(2, 64))
    # the C ABI refuses inconsistent lattice sizes and two branches at once
    a = _lib.MlpArgs()
    blob, aux = fine.packed()
    a.use_voxel, a.sigma_only, a.do_scene = 1, 1, 1
    a.blob, a.aux, a.grid = blob.data_ptr(), aux.data_ptr(), emb.grid_struct()
    ax = torch.zeros(4, device=DEV)
    out = torch.zeros(64, device=DEV)
    a.lat_x = a.lat_y = a.lat_z = ax.data_ptr()
    a.lat_n[0], a.lat_n[1], a.lat_n[2] = 4, 4, 4
    a.n_points, a.sigma = 63, out.data_ptr()
    assert _lib.lib().objnerf_mlp_eval(C.byref(a), _lib.stream_ptr()) != 0
    assert b"lat_n" in _lib.lib().objnerf_last_error()
    a.n_points, a.do_object, a.inst_sigma = 64, 1, out.data_ptr()
    assert _lib.lib().objnerf_mlp_eval(C.byref(a), _lib.stream_ptr()) != 0
    assert b"one branch" in _lib.lib().objnerf_last_error()


def test_embedding_with_linear_frequency_bands():
    """Embedding(logscale=False): bands torch.linspace(1, 2^(F-1), F) (embedding_helper.py:54-55) against the formula"""
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(300, 3, generator=g) * 4 - 2)
    for nf in (4, 10):
        e = A.Embedding(3, nf, logscale=False)
        assert e.out_channels == 3 * (2 * nf + 1) and torch.equal(e.freq_bands, torch.linspace(1, 2 ** (nf - 1), nf))
        with torch.no_grad():
            got = e(x.to(DEV)).cpu()
        want = torch.cat([x] + [f(fr * x) for fr in torch.linspace(1, 2 ** (nf - 1), nf) for f in (torch.sin, torch.cos)], -1)
        assert got.shape == want.shape and (got - want).abs().max().item() < 2e-6


def test_mlp_ragged_point_counts():
    """tile tails: point counts that are not multiples of 32 / 128, including 1"""
    m = scene("plain").models["coarse"]
    P = H.state(m)
    g = torch.Generator().manual_seed(5)
    for n in (1, 31, 33, 127, 129, 1000):
        ex, ed = torch.randn(n, 63, generator=g), torch.randn(n, 27, generator=g)
        with torch.no_grad():
            o = m({"emb_xyz": ex.to(DEV), "emb_dir": ed.to(DEV)})
        sg, c = O.mlp_scene(P, ex, ed)
        check(o["sigma"], sg, 1e-5, "n=%d sigma" % n)
        check(o["rgb"], c, 1e-5, "n=%d rgb" % n)
    with torch.no_grad():
        o = m({"emb_xyz": torch.zeros(0, 63, device=DEV), "emb_dir": torch.zeros(0, 27, device=DEV)})
    assert o["sigma"].shape == (0, 1)


def test_weight_repack_on_parameter_change():
    m = cases.scene_for(A, "plain", device=DEV).models["coarse"]
    i = cases.mlp_inputs(False)
    args = {"emb_xyz": i["emb_xyz"].to(DEV), "emb_dir": i["emb_dir"].to(DEV)}
    with torch.no_grad():
        a = m(args)["sigma"].clone()
        m.sigma.bias.add_(1.0)                       # in-place update bumps ._version -> repack
        b = m(args)["sigma"]
    assert torch.allclose(b, a + 1.0, atol=1e-5)


def test_sample_pdf_matches_reference():
    g = cases.load_golden("stage_sample_pdf")
    bins, w, u = cases.pdf_inputs()
    det = A.sample_pdf(bins.to(DEV), w.to(DEV), 64, det=True)
    rnd = A.sample_pdf(bins.to(DEV), w.to(DEV), u.shape[1], det=False, u=u.to(DEV))
    # Inverse-CDF sampling is ill-conditioned in z where the pdf is tiny (dz = dcdf / density) and in
    # F(z) where a narrow bin is heavy, so each sample is graded in whichever domain is well conditioned
    # (helpers.sampler_residual).  The reference's own fp32 samples satisfy the same bound.
    udet = torch.linspace(0, 1, 64).expand(bins.shape[0], 64)
    # Bound: eps = 1e-5.  The algorithm itself is discontinuous where a bin's pdf crosses eps
    # (`denom[denom < eps] = 1`, rendering.py:53-54): two correct fp32 evaluations can take different
    # sides, which moves F(z) by at most the bin's mass (< eps).  Away from that, residuals are ~1e-7.
    for ours, want, uu in ((det, g["det"], udet), (rnd, g["rnd"], u)):
        r_ours, r_ref = H.sampler_residual(bins, w, uu, ours.cpu()), H.sampler_residual(bins, w, uu, want)
        assert r_ours.max().item() < 1.5e-5 and r_ref.max().item() < 1.5e-5
        assert (r_ours > 1e-6).float().mean().item() < 0.01      # and almost all samples are roundoff-exact
        # z-domain agreement with the reference for the bulk of the samples
        dz = (ours.cpu() - want).abs() / bins.abs().max()
        assert dz.median().item() < 1e-6 and (dz > 1e-4).float().mean().item() < 0.02
    # det samples are non-decreasing, start exactly at bins[0] and end at bins[-1] (the last
    # one is exact only when the fp32 cdf does not overshoot 1.0 by an ulp, in the reference as well)
    assert (det[:, 1:] >= det[:, :-1]).all()
    assert torch.equal(det[:, 0].cpu(), bins[:, 0])
    assert (det[:, -1].cpu() <= bins[:, -1]).all() and (bins[:, -1] - det[:, -1].cpu()).max().item() < 1e-3


def _composite_case(n=50, S=64, seed=7):
    g = torch.Generator().manual_seed(seed)
    z = torch.sort(torch.rand(n, S, generator=g) * 3 + 0.1, -1)[0]
    return dict(z=z, sigma=torch.randn(n, S, generator=g) * 8, rgb=torch.rand(n, S, 3, generator=g),
                isig=torch.randn(n, S, generator=g) * 8, irgb=torch.rand(n, S, 3, generator=g),
                noise=torch.randn(n, S, generator=g), noise_i=torch.randn(n, S, generator=g),
                ptm=(torch.arange(n) % 4 == 0))


def _run_composite(d, **kw):
    n, S = d["z"].shape
    dv = {k: v.to(DEV).contiguous() for k, v in d.items()}
    out = {k: torch.empty(n, *s, device=DEV) for k, s in dict(weights=(S,), opacity=(), rgb_map=(3,), depth=(),
                                                               rgb_inst=(3,), depth_inst=(), opacity_inst=()).items()}
    a = _lib.CompositeArgs()
    a.n_rays, a.S = n, S
    a.z_vals, a.sigma, a.rgb = dv["z"].data_ptr(), dv["sigma"].data_ptr(), dv["rgb"].data_ptr()
    if kw.get("inst", True):
        a.inst_sigma, a.inst_rgb = dv["isig"].data_ptr(), dv["irgb"].data_ptr()
    a.noise_std = kw.get("noise_std", 0.0)
    if a.noise_std:
        a.noise, a.noise_inst = dv["noise"].data_ptr(), dv["noise_i"].data_ptr()
    a.white_back = int(kw.get("white_back", False))
    a.use_zero_as_last_delta = int(kw.get("zero_last", False))
    a.occlusion = int(kw.get("occlusion", False))
    a.frustum_bound_th = kw.get("th", 0.0)
    ptm8 = dv["ptm"].to(torch.uint8)
    if kw.get("ptm", False):
        a.pass_through_mask = ptm8.data_ptr()
    a.rays_in_bbox = int(kw.get("rays_in_bbox", False))
    for k, t in out.items():
        setattr(a, k, t.data_ptr())
    _lib.check(_lib.lib().objnerf_composite(C.byref(a), _lib.stream_ptr()), "composite")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("kw", [dict(), dict(white_back=True, zero_last=True),
                                dict(occlusion=True, th=0.025, ptm=True, rays_in_bbox=True),
                                dict(noise_std=1.0, occlusion=True, th=0.3), dict(inst=False)])
@pytest.mark.parametrize("S", [64, 128, 100, 192])
def test_composite_matches_oracle(kw, S):
    d = _composite_case(S=S)
    out = _run_composite(d, **kw)
    inst = kw.get("inst", True)
    ref = O.composite(d["z"], d["sigma"], d["rgb"], d["isig"] if inst else None, d["irgb"] if inst else None,
                      d["noise"], d["noise_i"], kw.get("noise_std", 0.0), kw.get("white_back", False),
                      kw.get("zero_last", False), kw.get("occlusion", False), kw.get("th", 0.0),
                      d["ptm"] if kw.get("ptm") else None, kw.get("rays_in_bbox", False))
    names = dict(weights="weights", opacity="opacity", rgb_map="rgb", depth="depth")
    if inst:
        names.update(rgb_inst="rgb_instance", depth_inst="depth_instance", opacity_inst="opacity_instance")
    for k, rk in names.items():
        check(out[k], ref[rk], 1e-5, "composite %s %r" % (k, kw))


def test_points_in_boxes_matches_reference():
    g = cases.load_golden("stage_points_in_boxes")
    _, boxes = cases.multi_inputs()
    m = check_in_any_boxes({4: boxes[0]}, cases.voxel_points(600).view(20, 30, 3).to(DEV))
    assert m.shape == (20, 30) and torch.equal(m.cpu(), g["inside"].bool())
    empty = check_in_any_boxes({}, cases.voxel_points(10).to(DEV))
    assert not empty.any()


def test_sample_coarse_bitwise():
    """coarse depths reproduce the reference's fp32 expression order bit for bit"""
    g = cases.load_golden("render_voxel_eval")
    g2 = cases.load_golden("render_voxel_disp_zero")
    rays = g["_rays"].to(DEV)
    l = _lib.lib()
    for use_disp, want in ((0, g["z_vals_coarse"]), (1, g2["z_vals_coarse"])):
        z = torch.empty(rays.shape[0], 64, device=DEV)
        steps = torch.linspace(0, 1, 64).to(DEV)
        _lib.check(l.objnerf_sample_coarse(_lib.ptr(rays), _lib.ptr(steps), None, 0.0, use_disp, rays.shape[0], 64,
                                           _lib.ptr(z), _lib.stream_ptr()), "sample_coarse")
        if use_disp:
            check(z, want, 2e-7, "z disp")       # two IEEE divisions: identical up to the last ulp of 1/x
        else:
            assert torch.equal(z.cpu(), want)


def test_generate_rays_matches_reference():
    """on-device ray generation + ray/OBB near-far against the reference's CPU path"""
    from object_nerf_amd.ray_utils import generate_rays
    g = cases.load_golden("stage_generate_rays")
    h, w, focal, Toc, box = cases.raygen_inputs()
    rg = cases.RAYGEN
    bg = generate_rays(h, w, focal, Toc, rg["near"], rg["far"])
    obj = generate_rays(h, w, focal, Toc, box=box, bbox_enlarge=rg["bbox_enlarge"])
    assert bg.shape == (h * w, 8)
    assert torch.equal(bg[:, :3].cpu(), g["background"][:, :3]) and torch.equal(bg[:, 6:].cpu(), g["background"][:, 6:])
    assert (bg[:, 3:6].cpu() - g["background"][:, 3:6]).abs().max().item() < 3e-7      # rotate + normalise: last-ulp class
    hit = obj[:, 7] > 0
    assert torch.equal(hit.cpu(), g["hit"].bool())                                     # same rays hit / miss
    assert (obj[~hit][:, 6:] == 0).all()
    check(obj[:, 6:], g["object"][:, 6:], 2e-6, "box near/far")


# ---- ray culling on the device + ray-subset evaluation (the pieces behind objnerf_render_rays_multi) ----
@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 5000])
def test_compact_rays_matches_nonzero(n):
    g = torch.Generator().manual_seed(n)
    S = 7
    z = torch.rand(n, S, generator=g) + 0.1
    dead = torch.rand(n, generator=g) < 0.4
    z[dead, -1] = 0.0
    if n == 63:
        z[:, -1] = 0.0                      # nothing survives
    zd = z.to(DEV)
    idx = torch.full((n,), -7, dtype=torch.int32, device=DEV)
    cnt = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    l = _lib.lib()
    scratch = torch.empty(l.objnerf_compact_scratch_ints(n), dtype=torch.int32, device=DEV)
    _lib.check(l.objnerf_compact_rays(_lib.ptr(zd), n, S, _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(scratch), _lib.stream_ptr()),
               "compact_rays")
    want = (z[:, -1] != 0).nonzero().squeeze(1).to(torch.int32)
    assert int(cnt.item()) == want.numel()
    assert torch.equal(idx.cpu()[:want.numel()], want)          # ascending
    assert (idx.cpu()[want.numel():] == -7).all()               # nothing written past the count


@pytest.mark.parametrize("branch", ["scene", "object"])
def test_mlp_eval_on_a_ray_subset(branch):
    """objnerf_mlp_args.ray_index / n_active: the listed rays get exactly the values of a full evaluation, the other
    rays' outputs are not touched; an empty list launches and writes nothing"""
    sc = scene("voxel")
    rays = H.test_rays(97, stride=31).to(DEV)
    n, S = rays.shape[0], 24
    z = (rays[:, 6:7] + (rays[:, 7:8] - rays[:, 6:7]) * torch.linspace(0, 1, S, device=DEV)).contiguous()
    blob, aux = sc.models["coarse"].packed()
    code = sc.code_library.embedding_instance.weight.detach()[4].contiguous()
    l = _lib.lib()

    def run(index, count, sigma, rgb):
        a = _lib.MlpArgs()
        a.use_voxel = 1
        a.blob, a.aux = blob.data_ptr(), aux.data_ptr()
        a.rays, a.z_vals, a.n_rays, a.S = rays.data_ptr(), z.data_ptr(), n, S
        a.grid = sc.embeddings["xyz"].grid_struct()
        if branch == "object":
            a.do_object, a.codes, a.code_stride = 1, code.data_ptr(), 0
            a.inst_sigma, a.inst_rgb = sigma.data_ptr(), rgb.data_ptr()
        else:
            a.do_scene, a.sigma, a.rgb = 1, sigma.data_ptr(), rgb.data_ptr()
        if index is not None:
            a.ray_index, a.n_active = index.data_ptr(), count.data_ptr()
        _lib.check(l.objnerf_mlp_eval(C.byref(a), _lib.stream_ptr()), "mlp_eval")
        torch.cuda.synchronize()

    full_s, full_c = torch.empty(n, S, device=DEV), torch.empty(n, S, 3, device=DEV)
    run(None, None, full_s, full_c)
    pick = torch.tensor([5, 0, 96, 17, 18, 40, 41, 42, 43, 64], dtype=torch.int32, device=DEV)      # not ascending: allowed
    padded = torch.cat([pick, torch.full((n - pick.numel(),), 3, dtype=torch.int32, device=DEV)])  # entries past the count
    for count in (pick.numel(), 0):
        sub_s, sub_c = torch.full((n, S), 123.0, device=DEV), torch.full((n, S, 3), 123.0, device=DEV)
        run(padded, torch.tensor([count], dtype=torch.int32, device=DEV), sub_s, sub_c)
        on = torch.zeros(n, dtype=torch.bool, device=DEV)
        on[pick[:count].long()] = True
        assert torch.equal(sub_s[on], full_s[on]) and torch.equal(sub_c[on], full_c[on])
        assert (sub_s[~on] == 123.0).all() and (sub_c[~on] == 123.0).all()


@pytest.mark.parametrize("K,S,where", [(20, 128, "LDS beyond the default 64 KiB"), (3, 2100, "global staging"), (40, 150, "global staging, 40 sets")])
def test_composite_multi_has_no_sample_limit(K, S, where):
    """the reference sorts any K*S (multi_rendering.py:112); until round 3 the kernel refused K*S > 2340 and K > 16.  Now:
    LDS staging up to 152 KiB (raised per-kernel limit), a scratch slice per workgroup beyond; up to 64 sets."""
    g = torch.Generator().manual_seed(8)
    n, M = 1100 if "global" in where else 9, K * S           # > 1024 rays: workgroups of the global variant take several rays
    zs = [torch.sort(torch.rand(n, S, generator=g) * 3.0, -1)[0] for _ in range(K)]
    zs[K - 1][::4] = 0.0
    sg = [torch.randn(n, S, generator=g) * 3.0 for _ in range(K)]
    cs = [torch.rand(n, S, 3, generator=g) for _ in range(K)]
    ref = O.composite_multi([z.clone() for z in zs], cs, sg, noise_std=0.0, white_back=False)
    dz, dsg, dcs = [z.to(DEV).contiguous() for z in zs], [t.to(DEV) for t in sg], [t.to(DEV) for t in cs]
    out = {k: torch.empty(n, *sh, device=DEV) for k, sh in dict(z=(M,), w=(M,), ids=(M,), opacity=(), rgb=(3,), depth=()).items()}
    own = [torch.empty(n, S, device=DEV) for _ in range(K)]
    a = _lib.CompositeMultiArgs()
    a.n_rays, a.K, a.S = n, K, S
    arr = C.c_void_p * K
    hz, hs, hr, ho = (arr(*[t.data_ptr() for t in ts]) for ts in (dz, dsg, dcs, own))
    a.h_z, a.h_sigma, a.h_rgb, a.h_own_weights = hz, hs, hr, ho
    a.z_sorted, a.weights, a.obj_ids = out["z"].data_ptr(), out["w"].data_ptr(), out["ids"].data_ptr()
    a.opacity, a.rgb_map, a.depth = out["opacity"].data_ptr(), out["rgb"].data_ptr(), out["depth"].data_ptr()
    nbytes = _lib.lib().objnerf_composite_multi_scratch_bytes(K, S)
    assert (nbytes > 0) == ("global" in where)
    if nbytes:
        assert _lib.lib().objnerf_composite_multi(C.byref(a), _lib.stream_ptr()) != 0       # refused without its scratch
        assert b"scratch" in _lib.lib().objnerf_last_error()
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        a.scratch = scratch.data_ptr()
    _lib.check(_lib.lib().objnerf_composite_multi(C.byref(a), _lib.stream_ptr()), "composite_multi")
    torch.cuda.synchronize()
    nz = ref["z_vals"] != 0          # cross-set ties at z == 0: the oracle's stable sort and the kernel agree by construction
    assert torch.equal(out["z"].cpu(), ref["z_vals"]) and torch.equal(out["ids"].cpu()[nz], ref["obj_ids"][nz])
    for k, rk in (("w", "weights"), ("opacity", "opacity"), ("rgb", "rgb"), ("depth", "depth")):
        check(out[k], ref[rk], 5e-5, "composite_multi %s / %s" % (where, rk))
    for i in (0, K - 1):
        check(own[i], ref["weights"][ref["obj_ids"] == i].view(n, S), 5e-5, "own weights %d" % i)


@pytest.mark.parametrize("variant", ["ascending_with_ties", "one_set_descending", "noise_white"])
def test_composite_multi_matches_oracle(variant):
    """objnerf_composite_multi (joint stable depth sort + compositing, multi_rendering.py:96-157) against the oracle:
    exact cross-set and in-set ties (stable order = set order, then sample order), a non-ascending set (the general
    rank-sort path instead of the K-run merge), noise and white background."""
    g = torch.Generator().manual_seed(5)
    n, K, S = 37, 3, 20
    zs = [torch.sort(torch.rand(n, S, generator=g) * 3.0, -1)[0] for _ in range(K)]
    zs[1][:, 3:6] = zs[0][:, 3:6]                 # exact ties across sets
    zs[1] = torch.sort(zs[1], -1)[0]
    zs[2][:, 7] = zs[2][:, 8]                     # a tie inside a set
    zs[2][::5] = 0.0                              # rays that missed their box
    if variant == "one_set_descending":
        zs[1] = torch.flip(zs[1], [-1]).contiguous()
    sg = [torch.randn(n, S, generator=g) * 3.0 for _ in range(K)]
    cs = [torch.rand(n, S, 3, generator=g) for _ in range(K)]
    noise = torch.randn(n, K * S, generator=g) if variant == "noise_white" else None
    kw = dict(noise_std=0.7, white_back=True) if variant == "noise_white" else dict(noise_std=0.0, white_back=False)
    ref = O.composite_multi([z.clone() for z in zs], cs, sg, noise=noise, **kw)
    M = K * S
    dz, dsg, dcs = [z.to(DEV).contiguous() for z in zs], [t.to(DEV) for t in sg], [t.to(DEV) for t in cs]
    out = {k: torch.empty(n, *sh, device=DEV) for k, sh in dict(z=(M,), w=(M,), ids=(M,), opacity=(), rgb=(3,), depth=()).items()}
    own = [torch.empty(n, S, device=DEV) for _ in range(K)]
    a = _lib.CompositeMultiArgs()
    a.n_rays, a.K, a.S = n, K, S
    arr = C.c_void_p * K
    hz, hs, hr, ho = (arr(*[t.data_ptr() for t in ts]) for ts in (dz, dsg, dcs, own))
    a.h_z, a.h_sigma, a.h_rgb, a.h_own_weights = hz, hs, hr, ho
    nd = noise.to(DEV) if noise is not None else None
    a.noise = nd.data_ptr() if nd is not None else None
    a.noise_std, a.white_back = kw["noise_std"], int(kw["white_back"])
    a.z_sorted, a.weights, a.obj_ids = out["z"].data_ptr(), out["w"].data_ptr(), out["ids"].data_ptr()
    a.opacity, a.rgb_map, a.depth = out["opacity"].data_ptr(), out["rgb"].data_ptr(), out["depth"].data_ptr()
    _lib.check(_lib.lib().objnerf_composite_multi(C.byref(a), _lib.stream_ptr()), "composite_multi")
    torch.cuda.synchronize()
    assert torch.equal(out["z"].cpu(), ref["z_vals"]) and torch.equal(out["ids"].cpu(), ref["obj_ids"])    # same stable order
    for k, rk in (("w", "weights"), ("opacity", "opacity"), ("rgb", "rgb"), ("depth", "depth")):
        check(out[k], ref[rk], 2e-5, "composite_multi %s/%s" % (variant, rk))
    for i in range(K):       # weights[obj_ids == i] in the set's own sample order (multi_rendering.py:269-271)
        want = ref["weights"][ref["obj_ids"] == i].view(n, S)
        if variant == "one_set_descending" and i == 1:
            want = torch.flip(want, [-1])          # sorted order of a descending set is its reverse
        check(own[i], want, 2e-5, "own weights %d" % i)


@pytest.mark.parametrize("sname", ["voxel", "plain"])
def test_hoisted_per_ray_terms_match_the_per_sample_contraction(sname):
    """objnerf_ray_bias + objnerf_mlp_args.ray_bias (the object code's and the direction embedding's share of four layers
    computed once per ray, their k-steps skipped in the MLP kernel: 2.45 % fewer MFMAs) against the same kernel contracting
    every term per sample point -- same sums in another association: sigma / rgb of both branches within 2e-6 (normwise), on a
    batch whose rays straddle waves (S = 40) and on one with S = 64; per-ray codes."""
    sc = cases.scene_for(A, sname, device=DEV)
    use_voxel = cases.SCENES[sname][0]
    l = _lib.lib()
    for S, n in ((40, 97), (64, 300)):
        rays = H.test_rays(n, w=64, h=48, stride=7).to(DEV)
        n = rays.shape[0]
        z = (rays[:, 6:7] + (rays[:, 7:8] - rays[:, 6:7]) * torch.linspace(0, 1, S, device=DEV)).contiguous()
        codes = sc.code_library({"instance_ids": synth.per_ray_ids(n, seed=9).to(DEV)})["embedding_instance"].detach().contiguous()
        blob, aux = sc.models["coarse"].packed()
        outs = []
        for hoist in (False, True):
            buf = {k: torch.empty(n, S, *sh, device=DEV) for k, sh in dict(sigma=(), rgb=(3,), isig=(), irgb=(3,)).items()}
            a = _lib.MlpArgs()
            a.use_voxel, a.do_scene, a.do_object = int(use_voxel), 1, 1
            a.blob, a.aux = blob.data_ptr(), aux.data_ptr()
            a.rays, a.z_vals, a.n_rays, a.S = rays.data_ptr(), z.data_ptr(), n, S
            a.codes, a.code_stride = codes.data_ptr(), 64
            if use_voxel:
                a.grid = sc.embeddings["xyz"].grid_struct()
            a.sigma, a.rgb, a.inst_sigma, a.inst_rgb = (buf[k].data_ptr() for k in ("sigma", "rgb", "isig", "irgb"))
            if hoist:
                rb = torch.empty(l.objnerf_ray_bias_floats(n), device=DEV)
                _lib.check(l.objnerf_ray_bias(C.byref(a), _lib.ptr(rb), _lib.stream_ptr()), "ray_bias")
                a.ray_bias = rb.data_ptr()
            _lib.check(l.objnerf_mlp_eval(C.byref(a), _lib.stream_ptr()), "mlp_eval")
            torch.cuda.synchronize()
            outs.append(buf)
        for k in outs[0]:
            assert torch.isfinite(outs[1][k]).all()
            assert H.normwise(outs[1][k], outs[0][k]) < 2e-6, (sname, S, k, H.normwise(outs[1][k], outs[0][k]))
