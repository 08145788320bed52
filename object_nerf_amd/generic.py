"""render_rays / render_rays_multi / the MLP forwards for architectures OTHER than the shipped default.

The reference builds any `config.model` shape (models/nerf_model.py:18-95: D, W, skips, inst_D, inst_W, inst_skips, N_freq_*,
voxel channel counts, code length; `Embedding(logscale=False)`, embedding_helper.py:53-56).  The persistent MFMA kernel
behind `render_rays` is specialised for the architecture every shipped reference config uses; everything else runs here:
the same pipeline, stage by stage through the C ABI -- coarse depths, sample points, embeddings written column block by
column block, the layer-wise MLP of csrc/generic.hip (fp32 MFMA GEMMs with bias / LeakyReLU / sigmoid epilogues),
compositing, inverse-CDF sampling -- with the intermediate tensors in memory.  Python only allocates and enqueues; there is
still no CPU or PyTorch arithmetic on the path.  With autograd recording `render_rays` takes `RenderRaysGenericFn` below: the
same stages with every layer's activations kept, and their backward.
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


def eval_points_slabbed(model, embeddings, rays, z, codes, scene, obj):
    """eval_points over ray slabs of RAY_CHUNK_POINTS sample points: the transient embeddings, repeated direction / code rows and
    the MLP workspace (~3.5 W + in_xyz + in_obj floats per point) are bounded by the slab, not by the batch -- the editor's
    32k-ray chunks x (64 + 64) samples would otherwise hold ~23 GB per ray set at W = 256, more for wider models."""
    n, S = z.shape
    step = max(1, RAY_CHUNK_POINTS // max(S, 1))
    if n <= step:
        return eval_points(model, embeddings, rays, z, codes, scene, obj)
    dev = z.device
    new = lambda *sh: torch.empty(n, S, *sh, dtype=torch.float32, device=dev)      # noqa: E731
    sg, c = (new(), new(3)) if scene else (None, None)
    isg, ic = (new(), new(3)) if obj else (None, None)
    for lo in range(0, n, step):
        hi = min(lo + step, n)
        parts = eval_points(model, embeddings, rays[lo:hi], z[lo:hi], codes[lo:hi].contiguous() if codes is not None else None, scene, obj)
        for dst, src in zip((sg, c, isg, ic), parts):
            if dst is not None:
                dst[lo:hi].copy_(src)
    return sg, c, isg, ic


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
            sg, c, isg, ic = eval_points_slabbed(model, embeddings, rays_c[k], zs[k], code, ids[k] == 0, ids[k] > 0)
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


# ---------------------------------------------------------------------------------------------------------------------
# training of non-default architectures: a differentiable render_rays on the layer-wise path
# ---------------------------------------------------------------------------------------------------------------------
def _freqs_dev(emb, dev):
    """device table of an Embedding's bands when they are not 2^k (logscale=False), else None"""
    if getattr(emb, "logscale", True):
        return None
    if getattr(emb, "_bands_dev", None) is None or emb._bands_dev.device != dev:
        emb._bands_dev = emb.freq_bands.to(torch.float32).to(dev).contiguous()
    return emb._bands_dev


class _GPass:
    __slots__ = ("z", "S", "xyz", "raw", "emb_xyz", "obj_voxel", "emb_dir", "code_pts", "sigma", "rgb", "isig", "irgb", "ws", "noise", "noise_i")


class RenderRaysGenericFn(torch.autograd.Function):
    """`render_rays` with autograd recording for ANY config.model architecture (the default one trains on the fused kernels,
    object_nerf_amd/autograd.py): forward = the stages of `generic.render_rays` with every layer's activations kept
    (objnerf_mlp_generic_train_forward), backward = compositing backward -> layer-wise MLP backward
    (objnerf_mlp_generic_train_backward) -> positional-encoding / trilinear backward into the voxel table -> per-ray sums
    of the code gradients.  Gradients: every ObjectNeRF parameter of both models, the voxel feature table, the codes; none
    through the sampler (rendering.py:307 detaches) or the occlusion mask."""

    @staticmethod
    def forward(ctx, meta, rays, codes, table, *params):
        from .autograd import _composite_args, _empty, _ptr_table
        l = _lib.lib()
        st = _lib.stream_ptr()
        dev = rays.device
        n, S, I = rays.shape[0], meta["S"], meta["I"]
        fi = meta["forward_instance"]
        mc, mf = meta["models"]
        exyz, edir = meta["emb_xyz"], meta["emb_dir"]
        vox = meta["use_voxel"]
        npar = len(params) // (2 if I > 0 else 1)
        p_c = [_lib.as_f32(p.detach()) for p in params[:npar]]
        p_f = [_lib.as_f32(p.detach()) for p in params[npar:2 * npar]] if I > 0 else None
        rays_c, codes_c = _lib.as_f32(rays.detach()), _lib.as_f32(codes.detach())
        rnd = meta["randoms"]
        arch = arch_of(mc)
        in_dir = arch.in_dir
        emb_dir_ray = _empty(n, in_dir, dev=dev)
        dirs = rays_c[:, 3:6].contiguous()
        _lib.check(l.objnerf_pos_encode_block(_lib.ptr(dirs), 3, n, 3, edir.N_freqs, _lib.ptr(_freqs_dev(edir, dev)), _lib.ptr(emb_dir_ray),
                                              in_dir, st), "pos_encode_block")

        def run_pass(model, z, pp, noise, noise_i):
            ps = _GPass()
            Sx = z.shape[1]
            P = n * Sx
            ps.S, ps.z = Sx, z
            ps.xyz = _empty(P, 3, dev=dev)
            _lib.check(l.objnerf_sample_points(_lib.ptr(rays_c), _lib.ptr(z), n, Sx, _lib.ptr(ps.xyz), st), "sample_points")
            ps.emb_xyz = _empty(P, arch.in_xyz, dev=dev)
            ps.raw = ps.obj_voxel = None
            if vox:
                C_, F = exyz.channels, exyz.N_freqs
                cs, co = C_ - exyz.instance_ftr_C, exyz.instance_ftr_C
                g = meta["grid"]
                ps.raw = _empty(P, C_, dev=dev)
                ps.obj_voxel = _empty(P, co * (2 * F + 1), dev=dev)
                _lib.check(l.objnerf_voxel_features(C.byref(g), C_, _lib.ptr(ps.xyz), P, _lib.ptr(ps.raw), C_, st), "voxel_features")
                _lib.check(l.objnerf_pos_encode_block(C.c_void_p(ps.raw.data_ptr()), C_, P, cs, F, None, _lib.ptr(ps.emb_xyz), arch.in_xyz, st), "pe")
                _lib.check(l.objnerf_pos_encode_block(_lib.ptr(ps.xyz), 3, P, 3, 10, None,
                                                      C.c_void_p(ps.emb_xyz.data_ptr() + 4 * cs * (2 * F + 1)), arch.in_xyz, st), "pe")
                _lib.check(l.objnerf_pos_encode_block(C.c_void_p(ps.raw.data_ptr() + 4 * cs), C_, P, co, F, None, _lib.ptr(ps.obj_voxel),
                                                      ps.obj_voxel.shape[1], st), "pe")
            else:
                _lib.check(l.objnerf_pos_encode_block(_lib.ptr(ps.xyz), 3, P, 3, exyz.N_freqs, _lib.ptr(_freqs_dev(exyz, dev)),
                                                      _lib.ptr(ps.emb_xyz), arch.in_xyz, st), "pe")
            ps.emb_dir = _repeat(emb_dir_ray, Sx)
            ps.code_pts = _repeat(codes_c, Sx) if fi else None
            ps.sigma, ps.rgb = _empty(P, dev=dev), _empty(P, 3, dev=dev)
            ps.isig, ps.irgb = (_empty(P, dev=dev), _empty(P, 3, dev=dev)) if fi else (None, None)
            ps.ws = _empty(l.objnerf_mlp_generic_train_workspace_floats(C.byref(arch), P), dev=dev)
            ps.noise, ps.noise_i = noise, noise_i
            ga, keep = _generic_train_args(arch, ps, pp, fi)
            _lib.check(l.objnerf_mlp_generic_train_forward(C.byref(ga), st), "mlp_generic_train_forward")
            outs = {"weights": _empty(n, Sx, dev=dev), "opacity": _empty(n, dev=dev), "rgb": _empty(n, 3, dev=dev), "depth": _empty(n, dev=dev)}
            if fi:
                outs.update({"rgb_instance": _empty(n, 3, dev=dev), "depth_instance": _empty(n, dev=dev), "opacity_instance": _empty(n, dev=dev)})
            ca = _composite_args(meta, ps, outs)
            _lib.check(l.objnerf_composite(C.byref(ca), st), "composite")
            return ps, outs

        z_c = _empty(n, S, dev=dev)
        pr = rnd.get("perturb_rand") if meta["perturb"] > 0 else None
        _lib.check(l.objnerf_sample_coarse(_lib.ptr(rays_c), _lib.ptr(meta["z_steps"]), _lib.ptr(pr) if pr is not None else None,
                                           float(meta["perturb"]), int(meta["use_disp"]), n, S, _lib.ptr(z_c), st), "sample_coarse")
        nz = rnd.get("noise", [None] * 4)
        passes, results = [], {}
        ps, outs = run_pass(mc, z_c, p_c, nz[0], nz[1])
        passes.append(ps)
        results.update({"%s_coarse" % k: v for k, v in outs.items()})
        results["z_vals_coarse"] = z_c
        if I > 0:
            z_f = _empty(n, S + I, dev=dev)
            det = meta["perturb"] == 0
            u = meta["u_det"] if det else rnd["u_rand"]
            _lib.check(l.objnerf_sample_pdf_merge(_lib.ptr(z_c), _lib.ptr(outs["weights"]), _lib.ptr(u), 0 if det else I, n, S, I, 1e-5, None,
                                                  _lib.ptr(z_f), st), "sample_pdf_merge")
            ps, outs = run_pass(mf, z_f, p_f, nz[2], nz[3])
            passes.append(ps)
            results.update({"%s_fine" % k: v for k, v in outs.items()})
            results["z_vals_fine"] = z_f
        keys = sorted(results)
        ctx.meta, ctx.passes, ctx.keys, ctx.arch = meta, passes, keys, arch
        ctx.p_c, ctx.p_f, ctx.rays_c = p_c, p_f, rays_c
        ctx.table_shape = table.shape if table is not None else None
        ctx.n_params, ctx.code_c = len(params), codes_c.shape[1]
        out = tuple(results[k] for k in keys)
        ctx.mark_non_differentiable(*[results[k] for k in keys if k.startswith(("weights_", "z_vals_"))])
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, *grads):
        from .autograd import _MAPS, _composite_args, _empty, _ptr_table
        l = _lib.lib()
        st = _lib.stream_ptr()
        meta, arch = ctx.meta, ctx.arch
        fi, vox = meta["forward_instance"], meta["use_voxel"]
        exyz = meta["emb_xyz"]
        dev = ctx.rays_c.device
        n = ctx.rays_c.shape[0]
        g = dict(zip(ctx.keys, grads))
        d_table = torch.zeros(ctx.table_shape, dtype=torch.float32, device=dev) if vox else None
        d_codes = torch.zeros(n, ctx.code_c, dtype=torch.float32, device=dev) if fi else None
        all_grads = []
        for typ, ps, pp in zip(("coarse", "fine"), ctx.passes, (ctx.p_c, ctx.p_f)):
            gp = [torch.zeros_like(p) for p in pp]
            all_grads.append(gp)
            P, Sx = ps.emb_xyz.shape[0], ps.S
            gm_t = {k: (_lib.as_f32(g["%s_%s" % (k, typ)]) if g.get("%s_%s" % (k, typ)) is not None else None) for k in _MAPS}
            if all(v is None for v in gm_t.values()):
                continue
            d_sigma, d_rgb = _empty(P, dev=dev), _empty(P, 3, dev=dev)
            d_isig, d_irgb = (_empty(P, dev=dev), _empty(P, 3, dev=dev)) if fi else (None, None)
            ca = _composite_args(meta, ps)
            _lib.check(l.objnerf_composite_backward(
                C.byref(ca), _lib.ptr(gm_t["rgb"]), _lib.ptr(gm_t["depth"]), _lib.ptr(gm_t["opacity"]), _lib.ptr(gm_t["rgb_instance"]),
                _lib.ptr(gm_t["depth_instance"]), _lib.ptr(gm_t["opacity_instance"]), _lib.ptr(d_sigma), _lib.ptr(d_rgb), _lib.ptr(d_isig),
                _lib.ptr(d_irgb), st), "composite_backward")
            ga, keep = _generic_train_args(arch, ps, pp, fi)
            gtable = _ptr_table(gp)
            emb_cols = 0
            d_emb = d_ov = d_code = None
            if vox:
                F = exyz.N_freqs
                cs, co = exyz.channels - exyz.instance_ftr_C, exyz.instance_ftr_C
                emb_cols = cs * (2 * F + 1)
                d_emb = _empty(P, emb_cols, dev=dev)
            if fi and arch.obj_voxel_c > 0:
                d_ov = _empty(P, arch.obj_voxel_c, dev=dev)
            if fi and arch.code_c > 0:
                d_code = _empty(P, arch.code_c, dev=dev)
            scratch = _empty(l.objnerf_mlp_generic_train_scratch_floats(C.byref(arch), P), dev=dev)
            _lib.check(l.objnerf_mlp_generic_train_backward(C.byref(ga), _lib.ptr(d_sigma), _lib.ptr(d_rgb), _lib.ptr(d_isig), _lib.ptr(d_irgb),
                                                            gtable, _lib.ptr(d_emb), emb_cols, _lib.ptr(d_ov), _lib.ptr(d_code),
                                                            _lib.ptr(scratch), st), "mlp_generic_train_backward")
            if vox:
                C_ = exyz.channels
                d_raw = torch.zeros(P, C_, dtype=torch.float32, device=dev)
                _lib.check(l.objnerf_pos_encode_block_backward(C.c_void_p(ps.raw.data_ptr()), C_, P, cs, F, None, _lib.ptr(d_emb), emb_cols,
                                                               C.c_void_p(d_raw.data_ptr()), C_, st), "pe_backward")
                if d_ov is not None:
                    _lib.check(l.objnerf_pos_encode_block_backward(C.c_void_p(ps.raw.data_ptr() + 4 * cs), C_, P, co, F, None, _lib.ptr(d_ov),
                                                                   arch.obj_voxel_c, C.c_void_p(d_raw.data_ptr() + 4 * cs), C_, st), "pe_backward")
                _lib.check(l.objnerf_voxel_features_backward(C.byref(meta["grid"]), C_, _lib.ptr(ps.xyz), P, _lib.ptr(d_raw), C_,
                                                             _lib.ptr(d_table), st), "voxel_features_backward")
            if d_code is not None:
                _lib.check(l.objnerf_sum_over_samples(_lib.ptr(d_code), n, Sx, arch.code_c, _lib.ptr(d_codes), st), "sum_over_samples")
        flat = list(all_grads[0]) + (list(all_grads[1]) if len(all_grads) > 1 else [])
        flat += [None] * (ctx.n_params - len(flat))
        return (None, None, d_codes, d_table, *flat)


def _generic_train_args(arch, ps, params, fi):
    g = _lib.MlpGenericArgs()
    g.arch = arch
    tab = (C.c_void_p * len(params))(*[t.data_ptr() for t in params])
    g.h_params = tab
    g.do_scene, g.do_object, g.sigma_only, g.n_points = 1, int(fi), 0, ps.emb_xyz.shape[0]
    g.emb_xyz, g.emb_dir = ps.emb_xyz.data_ptr(), ps.emb_dir.data_ptr()
    g.sigma, g.rgb = ps.sigma.data_ptr(), ps.rgb.data_ptr()
    if fi:
        g.inst_sigma, g.inst_rgb = ps.isig.data_ptr(), ps.irgb.data_ptr()
        if arch.code_c > 0:
            g.obj_code = ps.code_pts.data_ptr()
        if arch.obj_voxel_c > 0:
            g.obj_voxel = ps.obj_voxel.data_ptr()
    g.workspace = ps.ws.data_ptr()
    return g, tab


def train_param_list(model):
    """the model's parameters in objnerf_arch_num_param_ptrs() order (weight, bias pairs)"""
    mods = dict(model.named_modules())
    names = ["xyz_encoding_%d.0" % (i + 1) for i in range(model.D)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"]
    names += ["instance_encoding_%d.0" % (i + 1) for i in range(model.inst_D)] + [
        "instance_encoding_final.0", "inst_dir_encoding.0", "instance_sigma", "inst_rgb.0"]
    out = []
    for nme in names:
        out += [mods[nme].weight, mods[nme].bias]
    return out
