"""render_rays / render_rays_multi / the MLP forwards for architectures OTHER than the shipped default.

The reference builds any `config.model` shape (models/nerf_model.py:18-95: D, W, skips, inst_D, inst_W, inst_skips, N_freq_*,
voxel channel counts, code length; `Embedding(logscale=False)`, embedding_helper.py:53-56).  The persistent MFMA kernel
behind `render_rays` is specialised for the architecture every shipped reference config uses; everything else runs here:
the same pipeline, stage by stage through the C ABI -- coarse depths, sample points, embeddings written column block by
column block, the layer-wise MLP of csrc/generic.hip (fp32 MFMA GEMMs with bias / LeakyReLU / sigmoid epilogues),
compositing, inverse-CDF sampling -- with the intermediate tensors in memory.  Python only allocates and enqueues; there is
still no CPU or PyTorch arithmetic on the path.  Inference only: training a non-default shape raises.
"""
import ctypes as C

import torch

from . import _lib

RAY_CHUNK_POINTS = 1 << 19      # sample points per MLP call (activations: ~3.5 * W floats per point)


def arch_of(model):
    a = _lib.Arch()
    a.D, a.W = int(model.D), int(model.W)
    a.inst_D, a.inst_W = int(model.inst_D), int(model.inst_W)
    sk, isk = [int(i) for i in model.skips if 0 < int(i) < a.D], [int(i) for i in model.inst_skips if 0 < int(i) < a.inst_D]
    if len(sk) > 8 or len(isk) > 8:
        raise NotImplementedError("object_nerf_amd: at most 8 skip layers per branch")
    a.n_skips, a.n_inst_skips = len(sk), len(isk)
    for i, v in enumerate(sk):
        a.skips[i] = v
    for i, v in enumerate(isk):
        a.inst_skips[i] = v
    a.in_xyz, a.in_dir = int(model.in_channels_xyz), int(model.in_channels_dir)
    a.code_c = int(model.N_obj_code_length)
    a.obj_voxel_c = int(model.inst_channel_in) - a.in_xyz - a.code_c
    return a


def param_table(model):
    """(ctypes array of device pointers, tensors kept alive) in objnerf_arch_num_param_ptrs() order"""
    mods = dict(model.named_modules())
    names = ["xyz_encoding_%d.0" % (i + 1) for i in range(model.D)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"]
    names += ["instance_encoding_%d.0" % (i + 1) for i in range(model.inst_D)] + [
        "instance_encoding_final.0", "inst_dir_encoding.0", "instance_sigma", "inst_rgb.0"]
    ts = []
    for n in names:
        m = mods[n]
        ts += [_lib.as_f32(m.weight.detach()), _lib.as_f32(m.bias.detach())]
    _lib.require_cuda(ts[0], "ObjectNeRF parameters")
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), ts


def mlp(model, emb_xyz, emb_dir, obj_voxel, obj_code, scene, obj, sigma_only=False):
    """the layer-wise MLP on pre-embedded rows -> (sigma, rgb, inst_sigma, inst_rgb), each None when its branch is off"""
    l = _lib.lib()
    dev = emb_xyz.device
    n = emb_xyz.shape[0]
    g = _lib.MlpGenericArgs()
    g.arch = arch_of(model)
    tab, keep = param_table(model)
    g.h_params = tab
    g.do_scene, g.do_object, g.sigma_only, g.n_points = int(scene), int(obj), int(bool(sigma_only)), n
    exyz = _lib.as_f32(emb_xyz)
    if exyz.shape[1] != g.arch.in_xyz:
        raise RuntimeError("emb_xyz has %d channels, expected %d" % (exyz.shape[1], g.arch.in_xyz))
    g.emb_xyz = exyz.data_ptr()
    keep.append(exyz)
    if emb_dir is not None:
        ed = _lib.as_f32(emb_dir)
        if ed.shape[1] != g.arch.in_dir:
            raise RuntimeError("emb_dir has %d channels, expected %d" % (ed.shape[1], g.arch.in_dir))
        g.emb_dir = ed.data_ptr()
        keep.append(ed)
    out = [None, None, None, None]
    if scene:
        out[0] = torch.empty(n, 1, dtype=torch.float32, device=dev)
        g.sigma = out[0].data_ptr()
        if not sigma_only:
            out[1] = torch.empty(n, 3, dtype=torch.float32, device=dev)
            g.rgb = out[1].data_ptr()
    if obj:
        out[2] = torch.empty(n, 1, dtype=torch.float32, device=dev)
        g.inst_sigma = out[2].data_ptr()
        if not sigma_only:
            out[3] = torch.empty(n, 3, dtype=torch.float32, device=dev)
            g.inst_rgb = out[3].data_ptr()
        if g.arch.code_c > 0:
            oc = _lib.as_f32(obj_code)
            if oc.shape[1] != g.arch.code_c:
                raise RuntimeError("obj_code has %d channels, expected %d" % (oc.shape[1], g.arch.code_c))
            g.obj_code = oc.data_ptr()
            keep.append(oc)
        if g.arch.obj_voxel_c > 0:
            ov = _lib.as_f32(obj_voxel)
            if ov.shape[1] != g.arch.obj_voxel_c:
                raise RuntimeError("obj_voxel has %d channels, expected %d" % (ov.shape[1], g.arch.obj_voxel_c))
            g.obj_voxel = ov.data_ptr()
            keep.append(ov)
    if n == 0:
        return out
    ws = torch.empty(l.objnerf_mlp_generic_workspace_floats(C.byref(g.arch), n), dtype=torch.float32, device=dev)
    g.workspace = ws.data_ptr()
    _lib.check(l.objnerf_mlp_generic(C.byref(g), _lib.stream_ptr()), "mlp_generic")
    return out


def _repeat(src, repeat):
    """(n, C) per-ray rows -> (n * repeat, C): the `repeat` of rendering.py:89-94, written by a kernel"""
    src = _lib.as_f32(src)
    n, c = src.shape
    out = torch.empty(n * repeat, c, dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().objnerf_repeat_rows(_lib.ptr(src), c, n * repeat, c, repeat, _lib.ptr(out), c, _lib.stream_ptr()), "repeat_rows")
    return out


def eval_points(model, embeddings, rays, z, codes, scene, obj):
    """the MLP chunk loop of inference_model (rendering.py:86-137) for one (n, S) depth array -> sigma (n,S), rgb (n,S,3),
    inst_sigma, inst_rgb (None when the branch is off)"""
    l = _lib.lib()
    n, S = z.shape
    dev = z.device
    xyz = torch.empty(n * S, 3, dtype=torch.float32, device=dev)
    _lib.check(l.objnerf_sample_points(_lib.ptr(rays), _lib.ptr(z), n, S, _lib.ptr(xyz), _lib.stream_ptr()), "sample_points")
    emb = embeddings["xyz"](xyz)
    e_xyz, e_obj = emb if isinstance(emb, tuple) else (emb, None)
    e_dir = _repeat(embeddings["dir"](rays[:, 3:6].contiguous()), S)
    code_rep = _repeat(codes, S) if (obj and codes is not None) else None
    sg, c, isg, ic = mlp(model, e_xyz, e_dir, e_obj, code_rep, scene, obj)
    v = lambda t, *s: None if t is None else t.view(n, S, *s)      # noqa: E731
    return v(sg), v(c, 3), v(isg), v(ic, 3)


def _composite(z, sg, c, isg, ic, flags, noise, noise_inst, ptm, out):
    n, S = z.shape
    ca = _lib.CompositeArgs()
    ca.n_rays, ca.S, ca.z_vals = n, S, z.data_ptr()
    ca.sigma, ca.rgb = sg.data_ptr(), c.data_ptr()
    if isg is not None:
        ca.inst_sigma, ca.inst_rgb = isg.data_ptr(), ic.data_ptr()
    if flags["noise_std"] != 0:
        ca.noise = noise.data_ptr()
        if isg is not None:
            ca.noise_inst = noise_inst.data_ptr()
    ca.noise_std = float(flags["noise_std"])
    ca.white_back, ca.use_zero_as_last_delta = int(flags["white_back"]), int(flags["use_zero_as_last_delta"])
    ca.occlusion = int((not flags["is_eval"]) and flags["frustum_bound_th"] > 0)          # rendering.py:192
    ca.frustum_bound_th = float(flags["frustum_bound_th"])
    if ptm is not None:
        ca.pass_through_mask = ptm.data_ptr()
    ca.rays_in_bbox = int(flags["rays_in_bbox"] and isg is not None)
    ca.weights, ca.opacity, ca.rgb_map, ca.depth = (out[k].data_ptr() for k in ("weights", "opacity", "rgb", "depth"))
    if isg is not None:
        ca.rgb_inst, ca.depth_inst, ca.opacity_inst = (out[k].data_ptr() for k in ("rgb_instance", "depth_instance", "opacity_instance"))
    _lib.check(_lib.lib().objnerf_composite(C.byref(ca), _lib.stream_ptr()), "composite")


def render_rays(models, embeddings, rays, codes, S, I, flags, randoms, z_steps, u_det, alloc_out):
    """models/rendering.py:233-337 stage by stage (see the module docstring).  rays (n, 8) fp32, codes (n, code_c);
    flags: use_disp, perturb, noise_std, white_back, forward_instance, is_eval, use_zero_as_last_delta, frustum_bound_th,
    rays_in_bbox, pass_through_mask; randoms: pre-drawn tensors or None; alloc_out(n, s) -> result tensors of one pass."""
    l = _lib.lib()
    n = rays.shape[0]
    dev = rays.device
    fi = bool(flags["forward_instance"])
    oc = alloc_out(n, S)
    of = alloc_out(n, S + I) if I > 0 else None
    ptm = flags.get("pass_through_mask")
    rnd = randoms or {}
    step = max(1, RAY_CHUNK_POINTS // max(S + I, 1))
    for lo in range(0, n, step):
        hi = min(lo + step, n)
        m = hi - lo
        r_ = rays[lo:hi]
        c_ = codes[lo:hi]
        sl = lambda o: {k: t[lo:hi] for k, t in o.items()}      # noqa: E731  (row slices of contiguous outputs are contiguous)
        pr = rnd.get("perturb_rand")
        _lib.check(l.objnerf_sample_coarse(_lib.ptr(r_), _lib.ptr(z_steps), _lib.ptr(pr[lo:hi].contiguous()) if pr is not None else None,
                                           float(flags["perturb"]), int(flags["use_disp"]), m, S, _lib.ptr(oc["z_vals"][lo:hi]),
                                           _lib.stream_ptr()), "sample_coarse")
        nz = rnd.get("noise", [None] * 4)
        sg, c, isg, ic = eval_points(models["coarse"], embeddings, r_, oc["z_vals"][lo:hi], c_, True, fi)
        _composite(oc["z_vals"][lo:hi], sg, c, isg, ic, flags, nz[0][lo:hi].contiguous() if nz[0] is not None else None,
                   nz[1][lo:hi].contiguous() if nz[1] is not None else None, ptm[lo:hi] if ptm is not None else None, sl(oc))
        if I <= 0:
            continue
        det = flags["perturb"] == 0
        u = u_det if det else rnd["u_rand"][lo:hi].contiguous()
        _lib.check(l.objnerf_sample_pdf_merge(_lib.ptr(oc["z_vals"][lo:hi]), _lib.ptr(oc["weights"][lo:hi]), _lib.ptr(u), 0 if det else I,
                                              m, S, I, 1e-5, None, _lib.ptr(of["z_vals"][lo:hi]), _lib.stream_ptr()), "sample_pdf_merge")
        sg, c, isg, ic = eval_points(models["fine"], embeddings, r_, of["z_vals"][lo:hi], c_, True, fi)
        _composite(of["z_vals"][lo:hi], sg, c, isg, ic, flags, nz[2][lo:hi].contiguous() if nz[2] is not None else None,
                   nz[3][lo:hi].contiguous() if nz[3] is not None else None, ptm[lo:hi] if ptm is not None else None, sl(of))
    return oc, of


def render_rays_multi(models, embeddings, table, rays_c, clips, ids, S, I, use_disp, perturb, noise_std, white_back, boxes,
                      z_steps, u_det, u_rand, noise):
    """render_tools/multi_rendering.py:160-325 stage by stage for a non-default architecture: per ray set coarse depths ->
    one branch of the layer-wise MLP -> sigma masks -> joint compositing -> per-set importance sampling -> the same again."""
    l = _lib.lib()
    K = len(rays_c)
    n = rays_c[0].shape[0]
    dev = rays_c[0].device

    def alloc(M, want_ids):
        o = {"z_vals": torch.empty(n, M, dtype=torch.float32, device=dev), "weights": torch.empty(n, M, dtype=torch.float32, device=dev),
             "opacity": torch.empty(n, dtype=torch.float32, device=dev), "depth": torch.empty(n, dtype=torch.float32, device=dev),
             "rgb": torch.empty(n, 3, dtype=torch.float32, device=dev)}
        if want_ids:
            o["obj_ids"] = torch.empty(n, M, dtype=torch.float32, device=dev)
        return o

    def one_pass(model, zs, out, own, nz):
        Sp = zs[0].shape[1]
        sgs, cs = [], []
        for k in range(K):
            code = table[ids[k]].reshape(1, -1).expand(n, -1).contiguous() if ids[k] > 0 else None
            sg, c, isg, ic = eval_points(model, embeddings, rays_c[k], zs[k], code, ids[k] == 0, ids[k] > 0)
            sg, c = (sg, c) if ids[k] == 0 else (isg, ic)
            sg, c = sg.contiguous(), c.contiguous()
            use_boxes = ids[k] == 0 and boxes is not None and boxes.shape[0] > 0
            _lib.check(l.objnerf_mask_sigma_rgb(_lib.ptr(sg), _lib.ptr(c), _lib.ptr(rays_c[k]), _lib.ptr(zs[k]), n, Sp,
                                                _lib.ptr(boxes) if use_boxes else None, boxes.shape[0] if use_boxes else 0,
                                                _lib.stream_ptr()), "mask_sigma_rgb")
            sgs.append(sg)
            cs.append(c)
        a = _lib.CompositeMultiArgs()
        a.n_rays, a.K, a.S = n, K, Sp
        arr = C.c_void_p * K
        hz, hs, hr = arr(*[t.data_ptr() for t in zs]), arr(*[t.data_ptr() for t in sgs]), arr(*[t.data_ptr() for t in cs])
        a.h_z, a.h_sigma, a.h_rgb = hz, hs, hr
        if own is not None:
            ho = arr(*[t.data_ptr() for t in own])
            a.h_own_weights = ho
        if noise_std != 0:
            a.noise = nz.data_ptr()
        a.noise_std, a.white_back = float(noise_std), int(bool(white_back))
        a.z_sorted, a.weights = out["z_vals"].data_ptr(), out["weights"].data_ptr()
        a.obj_ids = out["obj_ids"].data_ptr() if "obj_ids" in out else None
        a.opacity, a.rgb_map, a.depth = out["opacity"].data_ptr(), out["rgb"].data_ptr(), out["depth"].data_ptr()
        nb = l.objnerf_composite_multi_scratch_bytes(K, Sp)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev) if nb > 0 else None
        if scratch is not None:
            a.scratch = scratch.data_ptr()
        _lib.check(l.objnerf_composite_multi(C.byref(a), _lib.stream_ptr()), "composite_multi")

    zc = []
    for k in range(K):
        z = torch.empty(n, S, dtype=torch.float32, device=dev)
        _lib.check(l.objnerf_sample_coarse(_lib.ptr(rays_c[k]), _lib.ptr(z_steps), None, 0.0, int(bool(use_disp)), n, S, _lib.ptr(z),
                                           _lib.stream_ptr()), "sample_coarse")
        zc.append(z)
    oc = alloc(K * S, True)
    own = [torch.empty(n, S, dtype=torch.float32, device=dev) for _ in range(K)] if I > 0 else None
    one_pass(models["coarse"], zc, oc, own, noise[0] if noise else None)
    of = None
    if I > 0:
        zf = []
        det = perturb == 0
        for k in range(K):
            z = torch.empty(n, S + I, dtype=torch.float32, device=dev)
            u = u_det if det else u_rand[k].contiguous()
            _lib.check(l.objnerf_sample_pdf_merge_clip(_lib.ptr(zc[k]), _lib.ptr(own[k]), _lib.ptr(u), 0 if det else I, n, S, I, 1e-5, None,
                                                       _lib.ptr(z), _lib.ptr(clips[k]) if clips[k] is not None else None,
                                                       _lib.stream_ptr()), "sample_pdf_merge_clip")
            zf.append(z)
        of = alloc(K * (S + I), False)
        one_pass(models["fine"], zf, of, None, noise[1] if noise else None)
    return oc, of
