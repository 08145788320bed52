"""Differentiable render_rays: what `training_step` (train.py:147-180) needs.

`render_rays` dispatches here whenever autograd is recording and something on the path requires a
gradient.  Round 6: the coarse and the fine pass are TWO `torch.autograd.Function` nodes (`render_rays_nodes`) -- their only
coupling is the detached `weights_coarse` -> sampler (rendering.py:305-310) -- so the fine model's parameter gradients
are final when the fine node's backward returns (it runs first: it was recorded last) and a data-parallel wrapper
(torch's DDP reducer, distributed.GradientSync.attach) starts exchanging them while the coarse node's backward still
runs.  OBJNERF_TRAIN_NODES=1 keeps the single node of rounds 1-5 (`RenderRaysFn`: same launches, one node; the two
forms' parameter gradients are bit-equal, tests/test_gpu_train.py).  Forward and backward of either form are sequences
of C-ABI kernel launches (include/objnerf_hip.h "training path"):

  forward   coarse depths -> sample points -> voxel / positional embedding (materialised: the weight-gradient products read it) ->
            fused persistent MLP forward that embeds in registers, hoists the per-ray constant terms, and keeps every layer's
            output and its LeakyReLU sign mask -> compositing -> sample_pdf + merge (no gradient, as in the reference:
            rendering.py:307 detaches) -> fine pass
  backward  compositing backward -> fused dgrad chain (fed by the masks) -> gradients w.r.t. the embeddings -> voxel-table scatter
            (LDS hash + atomics into the feature table) -> one grouped deterministic weight-gradient pass, which also leaves
            16-point column sums from which a small second pass forms the terms that are constant along a ray (direction / code
            weight columns, the code gradient) -> per-ray sum of the code gradients
  (OBJNERF_TRAIN_LAYERWISE=1: the same step layer by layer on the fp32 MFMA GEMM; non-default architectures: generic.py)

Gradients are produced for every ObjectNeRF parameter (coarse and fine), the voxel feature table and
`embedding_instance`; none for rays / depths (the reference has none either).  The Python here only
allocates tensors, repeats per-ray rows to per-point rows and orders the launches.
"""
import ctypes as C
import os

import torch

from . import _lib

_MAPS = ("rgb", "depth", "opacity", "rgb_instance", "depth_instance", "opacity_instance")


def _empty(*shape, dev):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class _Pass:
    """saved state of one (coarse / fine) pass"""
    __slots__ = ("z", "xyz", "emb_xyz", "obj_voxel", "emb_dir", "code_pts", "sigma", "rgb", "isig", "irgb", "ws",
                 "noise", "noise_i", "S", "per_ray")


def _ptr_table(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _composite_args(meta, ps, outs=None):
    a = _lib.CompositeArgs()
    n = ps.z.shape[0]
    a.n_rays, a.S = n, ps.S
    a.z_vals, a.sigma, a.rgb = ps.z.data_ptr(), ps.sigma.data_ptr(), ps.rgb.data_ptr()
    if meta["forward_instance"]:
        a.inst_sigma, a.inst_rgb = ps.isig.data_ptr(), ps.irgb.data_ptr()
    a.noise_std = float(meta["noise_std"])
    if meta["noise_std"] != 0:
        a.noise = ps.noise.data_ptr()
        if meta["forward_instance"]:
            a.noise_inst = ps.noise_i.data_ptr()
    a.white_back = int(meta["white_back"])
    a.use_zero_as_last_delta = int(meta["use_zero_as_last_delta"])
    a.occlusion = int((not meta["is_eval"]) and meta["frustum_bound_th"] > 0)
    a.frustum_bound_th = float(meta["frustum_bound_th"])
    if meta["ptm"] is not None:
        a.pass_through_mask = meta["ptm"].data_ptr()
    a.rays_in_bbox = int(meta["rays_in_bbox"] and meta["forward_instance"])
    if outs is not None:
        a.weights, a.opacity, a.rgb_map, a.depth = (outs[k].data_ptr() for k in ("weights", "opacity", "rgb", "depth"))
        if meta["forward_instance"]:
            a.rgb_inst, a.depth_inst, a.opacity_inst = (outs[k].data_ptr() for k in
                                                        ("rgb_instance", "depth_instance", "opacity_instance"))
    return a


def _train_args(meta, ps, params, packed=None, rays=None, codes=None, ray_bias_ws=None):
    a = _lib.TrainArgs()
    if packed is not None:
        blob, aux, blob_bwd = packed[:3]
        a.bwd_dx = int(bool(packed[3])) if len(packed) > 3 else 0
        a.aux = aux.data_ptr()
        if blob is not None:
            a.blob = blob.data_ptr()
        if blob_bwd is not None:
            a.blob_bwd = blob_bwd.data_ptr()
        if rays is not None and blob is not None and os.environ.get("OBJNERF_TRAIN_LAYERWISE") != "mem":
            # forward: embeddings recomputed in registers from the un-embedded inputs ("mem": read them back instead)
            a.rays, a.z_vals, a.n_rays, a.S = rays.data_ptr(), ps.z.data_ptr(), rays.shape[0], ps.S
            if codes is not None:
                a.codes, a.code_stride = codes.data_ptr(), codes.stride(0)
            if meta["use_voxel"]:
                a.grid = meta["grid"]
            if ray_bias_ws is not None:
                a.ray_bias_ws = ray_bias_ws.data_ptr()
    a.use_voxel, a.do_object = int(meta["use_voxel"]), int(meta["forward_instance"])
    a.n_points = ps.emb_xyz.shape[0]
    table = _ptr_table(params)
    a.h_params = table
    a.emb_xyz = ps.emb_xyz.data_ptr()
    if ps.emb_dir is not None:
        a.emb_dir = ps.emb_dir.data_ptr()
    if meta["forward_instance"]:
        if ps.code_pts is not None:
            a.obj_code = ps.code_pts.data_ptr()
        if meta["use_voxel"]:
            a.obj_voxel = ps.obj_voxel.data_ptr()
        a.inst_sigma, a.inst_rgb = ps.isig.data_ptr(), ps.irgb.data_ptr()
    a.sigma, a.rgb = ps.sigma.data_ptr(), ps.rgb.data_ptr()
    a.workspace = ps.ws.data_ptr()
    return a, table


def _dir_embedding(rays_c):
    """Embedding(3, 4) of the ray directions, one row per ray (pos_encode stage kernel)"""
    l = _lib.lib()
    n = rays_c.shape[0]
    dirs = rays_c[:, 3:6].contiguous()
    emb = _empty(n, 27, dev=rays_c.device)
    _lib.check(l.objnerf_pos_encode(_lib.ptr(dirs), n, 3, 4, _lib.ptr(emb), _lib.stream_ptr()), "pos_encode")
    return emb


def _run_pass(meta, rays_c, codes_c, emb_dir_ray, z, pp, noise, noise_i, packed):
    """forward of one pass at the depths z (n, Sx): -> (_Pass with everything the backward needs, dict of result tensors)"""
    l = _lib.lib()
    st = _lib.stream_ptr()
    dev = rays_c.device
    n = rays_c.shape[0]
    fi, vox = meta["forward_instance"], meta["use_voxel"]
    ps = _Pass()
    Sx = z.shape[1]
    P = n * Sx
    ps.S, ps.z = Sx, z
    ps.xyz = _empty(P, 3, dev=dev)
    _lib.check(l.objnerf_sample_points(_lib.ptr(rays_c), _lib.ptr(z), n, Sx, _lib.ptr(ps.xyz), st), "sample_points")
    if vox:
        ps.emb_xyz, ps.obj_voxel = _empty(P, 271, dev=dev), _empty(P, 104, dev=dev)
        _lib.check(l.objnerf_voxel_embed(C.byref(meta["grid"]), _lib.ptr(ps.xyz), P, _lib.ptr(ps.emb_xyz),
                                         _lib.ptr(ps.obj_voxel), st), "voxel_embed")
    else:
        ps.emb_xyz, ps.obj_voxel = _empty(P, 63, dev=dev), None
        _lib.check(l.objnerf_pos_encode(_lib.ptr(ps.xyz), P, 3, 10, _lib.ptr(ps.emb_xyz), st), "pos_encode")
    # per-ray form of the terms that are constant along a ray (include/objnerf_hip.h, objnerf_train_args.emb_dir_ray): the
    # fused kernels embed directions / read codes per ray in the forward, and the backward contracts the weight columns
    # they meet over 16-point segment sums -- the per-point copies are then never read.  ONE predicate, decided here and
    # carried to the backward in the pass state (Python and C used to decide it separately; the C side now gets it
    # as objnerf_train_args.emb_dir_ray != NULL and no longer re-reads the environment)
    ps.per_ray = (packed is not None and packed[0] is not None and packed[2] is not None and Sx % 16 == 0
                  and (not fi or codes_c.stride(0) == 64)
                  and os.environ.get("OBJNERF_TRAIN_LAYERWISE") != "mem" and _env_flag("OBJNERF_TRAIN_PER_RAY", True)
                  and os.environ.get("OBJNERF_WGRAD") != "atomic")
    ps.emb_dir = None if ps.per_ray else emb_dir_ray.repeat_interleave(Sx, 0)
    ps.code_pts = (None if ps.per_ray else codes_c.repeat_interleave(Sx, 0)) if fi else None
    ps.sigma, ps.rgb = _empty(P, dev=dev), _empty(P, 3, dev=dev)
    ps.isig, ps.irgb = (_empty(P, dev=dev), _empty(P, 3, dev=dev)) if fi else (None, None)
    ps.ws = _empty(l.objnerf_train_workspace_floats(int(fi), P), dev=dev)
    ps.noise, ps.noise_i = noise, noise_i
    # scratch of the hoisted per-ray terms (objnerf_train_args.ray_bias_ws): only the fused forward uses it
    rb = _empty(n, _lib.RAY_BIAS_FLOATS, dev=dev) if (packed is not None and packed[0] is not None) else None
    a, keep = _train_args(meta, ps, pp, packed, rays_c, codes_c if fi else None, rb)
    _lib.check(l.objnerf_mlp_train_forward(C.byref(a), st), "mlp_train_forward")
    outs = {"weights": _empty(n, Sx, dev=dev), "opacity": _empty(n, dev=dev), "rgb": _empty(n, 3, dev=dev),
            "depth": _empty(n, dev=dev)}
    if fi:
        outs.update({"rgb_instance": _empty(n, 3, dev=dev), "depth_instance": _empty(n, dev=dev),
                     "opacity_instance": _empty(n, dev=dev)})
    ca = _composite_args(meta, ps, outs)
    _lib.check(l.objnerf_composite(C.byref(ca), st), "composite")
    return ps, outs


def _env_flag(name, default):
    """'0' / 'false' / 'no' / 'off' (any case) = False, unset = default, anything else = True"""
    v = os.environ.get(name)
    if v is None:
        return default
    return v.strip().lower() not in ("0", "false", "no", "off", "")


def _pass_backward(meta, ps, pp, packed, gm_t, gp, rays_c, emb_dir_ray, codes_c, d_table, d_codes):
    """backward of one pass: the gradients of its pixel maps (gm_t: name -> tensor or None) into the parameter-gradient views gp
    (accumulated), the voxel-table gradient d_table (scattered into) and the per-ray code gradient d_codes (accumulated)"""
    l = _lib.lib()
    st = _lib.stream_ptr()
    fi, vox = meta["forward_instance"], meta["use_voxel"]
    dev = rays_c.device
    n = rays_c.shape[0]
    P, Sx = ps.emb_xyz.shape[0], ps.S
    d_sigma, d_rgb = _empty(P, dev=dev), _empty(P, 3, dev=dev)
    d_isig, d_irgb = (_empty(P, dev=dev), _empty(P, 3, dev=dev)) if fi else (None, None)
    ca = _composite_args(meta, ps)
    _lib.check(l.objnerf_composite_backward(
        C.byref(ca), _lib.ptr(gm_t["rgb"]), _lib.ptr(gm_t["depth"]), _lib.ptr(gm_t["opacity"]),
        _lib.ptr(gm_t["rgb_instance"]), _lib.ptr(gm_t["depth_instance"]), _lib.ptr(gm_t["opacity_instance"]),
        _lib.ptr(d_sigma), _lib.ptr(d_rgb), _lib.ptr(d_isig), _lib.ptr(d_irgb), st), "composite_backward")
    a, keep = _train_args(meta, ps, pp, packed)
    gtable = _ptr_table(gp)
    d_emb = _empty(P, ps.emb_xyz.shape[1], dev=dev)
    d_ov = _empty(P, 104, dev=dev) if (fi and vox) else None
    seg = 16 if ps.per_ray else 1                       # per-ray form: one code-gradient row per 16 points
    d_code = _empty(P // seg, 64, dev=dev) if fi else None
    if ps.per_ray:
        a.emb_dir_ray, a.n_rays, a.S = emb_dir_ray.data_ptr(), n, Sx
        if fi:
            a.codes, a.code_stride = codes_c.data_ptr(), codes_c.stride(0)
    scratch = _empty(l.objnerf_train_scratch_floats(P), dev=dev)
    if vox:      # the table scatter rides inside the call (objnerf_train_args.scatter_*)
        assert ps.xyz.is_contiguous() and d_table.is_contiguous()
        a.grid = meta["grid"]
        a.scatter_xyz, a.scatter_table_grad = ps.xyz.data_ptr(), d_table.data_ptr()
    _lib.check(l.objnerf_mlp_train_backward(C.byref(a), _lib.ptr(d_sigma), _lib.ptr(d_rgb), _lib.ptr(d_isig),
                                            _lib.ptr(d_irgb), gtable, _lib.ptr(d_emb), _lib.ptr(d_ov), _lib.ptr(d_code),
                                            _lib.ptr(scratch), st), "mlp_train_backward")
    if fi:
        _lib.check(l.objnerf_sum_over_samples(_lib.ptr(d_code), n, Sx // seg, 64, _lib.ptr(d_codes), st), "sum_over_samples")


def _flat_grad_views(param_lists, dev):
    """every parameter gradient of the given models in ONE zero-initialised flat buffer (one memset instead of one per tensor;
    the kernels accumulate into it): views at 256-byte boundaries, one list per model"""
    offs, tot = [], 0
    for pl in param_lists:
        for p in pl:
            offs.append(tot)
            tot += (p.numel() + 63) // 64 * 64
    flat_g = torch.zeros(tot, dtype=torch.float32, device=dev)
    out, i = [], 0
    for pl in param_lists:
        out.append([flat_g[offs[i + j]:offs[i + j] + p.numel()].view(p.shape) for j, p in enumerate(pl)])
        i += len(pl)
    return out


_PASS_KEYS = ("weights", "opacity", "z_vals", "rgb", "depth", "rgb_instance", "depth_instance", "opacity_instance")


class RenderPassFn(torch.autograd.Function):
    """ONE pass (coarse or fine) at given depths: the node of the two-node form.  Inputs: the depths z (no gradient: the sampler is
    detached in the reference too), rays, per-ray codes, the voxel table, this pass's model parameters.  Outputs in the order of
    `RenderPassFn.keys(meta)`."""

    @staticmethod
    def keys(meta):
        return sorted(k for k in _PASS_KEYS if meta["forward_instance"] or not k.endswith("_instance"))

    @staticmethod
    def forward(ctx, meta, typ, z, rays, codes, table, *params):
        rays_c = _lib.as_f32(rays.detach())
        codes_c = _lib.as_f32(codes.detach())
        pp = [_lib.as_f32(p.detach()) for p in params]
        shared = meta["shared"]
        if "emb_dir_ray" not in shared:
            shared["emb_dir_ray"] = _dir_embedding(rays_c)
        nz = meta["randoms"].get("noise", [None] * 4)
        i = 0 if typ == "coarse" else 1
        packed = (meta.get("packed") or (None, None))[i]
        ps, outs = _run_pass(meta, rays_c, codes_c, shared["emb_dir_ray"], z, pp, nz[2 * i], nz[2 * i + 1], packed)
        outs["z_vals"] = z
        keys = RenderPassFn.keys(meta)
        ctx.meta, ctx.typ, ctx.ps, ctx.pp, ctx.packed, ctx.keys = meta, typ, ps, pp, packed, keys
        ctx.rays_c, ctx.codes_c, ctx.emb_dir_ray = rays_c, codes_c, shared["emb_dir_ray"]
        ctx.table_shape = table.shape if table is not None else None
        ctx.mark_non_differentiable(outs["weights"], outs["z_vals"])
        ctx.set_materialize_grads(False)
        return tuple(outs[k] for k in keys)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        fi, vox = meta["forward_instance"], meta["use_voxel"]
        dev = ctx.rays_c.device
        n = ctx.rays_c.shape[0]
        g = dict(zip(ctx.keys, grads))
        gm_t = {k: (_lib.as_f32(g[k]) if g.get(k) is not None else None) for k in _MAPS}
        if all(v is None for v in gm_t.values()):
            return (None,) * (6 + len(ctx.pp))
        # The voxel-table gradient (76.8 MB at the reference's 800,000 rows) and the code gradient receive a contribution from BOTH
        # nodes.  Returned separately, autograd would add two full-size tensors (one more memset + 230 MB of traffic per step:
        # measured +0.3 ms of a 19.4 ms step).  Instead the node that runs first (the fine node) allocates the buffers, returns
        # them and leaves them in the call's shared state; the node that runs later accumulates INTO them (same stream: ordered
        # behind the first node's kernels) and returns None.  Autograd keeps the first node's tensors by reference until every
        # producer of that input has run, then hands them to AccumulateGrad / the DDP reducer -- which therefore sees the sum.
        # If only one of the nodes runs (a loss on one pass only), it is "first" and its buffers are complete.
        # "First" is per backward pass: the buffers carry the id of the autograd graph task that made them (a second backward over
        # a retained graph, or one that runs only one of the nodes, starts over).
        shared = meta["shared"]
        task = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else None
        first = task is None or task < 0 or shared.get("task") != task
        if first:
            shared["task"] = task
            shared["d_table"] = torch.zeros(ctx.table_shape, dtype=torch.float32, device=dev) if vox else None
            shared["d_codes"] = torch.zeros(n, 64, dtype=torch.float32, device=dev) if fi else None
        d_table, d_codes = shared["d_table"], shared["d_codes"]
        gp = _flat_grad_views([ctx.pp], dev)[0]
        _pass_backward(meta, ctx.ps, ctx.pp, ctx.packed, gm_t, gp, ctx.rays_c, ctx.emb_dir_ray, ctx.codes_c, d_table, d_codes)
        if first:
            return (None, None, None, None, d_codes, d_table, *gp)
        return (None, None, None, None, None, None, *gp)


def render_rays_nodes(meta, rays, codes, table, params_coarse, params_fine):
    """The differentiable render_rays as TWO autograd nodes (coarse pass, fine pass) around the detached sampler -> the results
    dict of the reference (16 tensors; 8 without the fine pass)."""
    l = _lib.lib()
    st = _lib.stream_ptr()
    dev = rays.device
    n, S, I = rays.shape[0], meta["S"], meta["I"]
    meta = dict(meta, shared={})
    rnd = meta["randoms"]
    rays_c = _lib.as_f32(rays.detach())
    z_c = _empty(n, S, dev=dev)
    pr = rnd.get("perturb_rand") if meta["perturb"] > 0 else None
    _lib.check(l.objnerf_sample_coarse(_lib.ptr(rays_c), _lib.ptr(meta["z_steps"]), _lib.ptr(pr) if pr is not None else None,
                                       float(meta["perturb"]), int(meta["use_disp"]), n, S, _lib.ptr(z_c), st), "sample_coarse")
    keys = RenderPassFn.keys(meta)
    results = {}
    oc = dict(zip(keys, RenderPassFn.apply(meta, "coarse", z_c, rays, codes, table, *params_coarse)))
    for k, v in oc.items():
        results["%s_coarse" % k] = v
    if I > 0:
        z_f = _empty(n, S + I, dev=dev)
        det = meta["perturb"] == 0
        u = meta["u_det"] if det else rnd["u_rand"]
        _lib.check(l.objnerf_sample_pdf_merge(_lib.ptr(z_c), _lib.ptr(oc["weights"].detach()), _lib.ptr(u), 0 if det else I, n, S, I,
                                              1e-5, None, _lib.ptr(z_f), st), "sample_pdf_merge")
        of = dict(zip(keys, RenderPassFn.apply(meta, "fine", z_f, rays, codes, table, *params_fine)))
        for k, v in of.items():
            results["%s_fine" % k] = v
    return results


class RenderRaysFn(torch.autograd.Function):
    """The single-node form (rounds 1-5; OBJNERF_TRAIN_NODES=1): both passes and the sampler inside one node."""

    @staticmethod
    def forward(ctx, meta, rays, codes, table, *params):
        l = _lib.lib()
        st = _lib.stream_ptr()
        dev = rays.device
        n, S, I = rays.shape[0], meta["S"], meta["I"]
        n_par = l.objnerf_num_param_ptrs()
        p_coarse = [_lib.as_f32(p.detach()) for p in params[:n_par]]
        p_fine = [_lib.as_f32(p.detach()) for p in params[n_par:2 * n_par]] if I > 0 else None
        rays_c = _lib.as_f32(rays.detach())
        codes_c = _lib.as_f32(codes.detach())
        rnd = meta["randoms"]
        pk = meta.get("packed") or (None, None)
        emb_dir_ray = _dir_embedding(rays_c)            # per-ray rows repeated per sample are copies, not arithmetic

        z_c = _empty(n, S, dev=dev)
        pr = rnd.get("perturb_rand") if meta["perturb"] > 0 else None
        _lib.check(l.objnerf_sample_coarse(_lib.ptr(rays_c), _lib.ptr(meta["z_steps"]), _lib.ptr(pr) if pr is not None else None,
                                           float(meta["perturb"]), int(meta["use_disp"]), n, S, _lib.ptr(z_c), st), "sample_coarse")
        nz = rnd.get("noise", [None] * 4)
        passes, results = [], {}
        ps, outs = _run_pass(meta, rays_c, codes_c, emb_dir_ray, z_c, p_coarse, nz[0], nz[1], pk[0])
        passes.append(ps)
        for k, v in outs.items():
            results["%s_coarse" % k] = v
        results["z_vals_coarse"] = z_c
        if I > 0:
            z_f = _empty(n, S + I, dev=dev)
            det = meta["perturb"] == 0
            u = meta["u_det"] if det else rnd["u_rand"]
            _lib.check(l.objnerf_sample_pdf_merge(_lib.ptr(z_c), _lib.ptr(outs["weights"]), _lib.ptr(u), 0 if det else I, n, S, I,
                                                  1e-5, None, _lib.ptr(z_f), st), "sample_pdf_merge")
            ps, outs = _run_pass(meta, rays_c, codes_c, emb_dir_ray, z_f, p_fine, nz[2], nz[3], pk[1])
            passes.append(ps)
            for k, v in outs.items():
                results["%s_fine" % k] = v
            results["z_vals_fine"] = z_f

        keys = sorted(results)
        ctx.meta, ctx.passes, ctx.keys = meta, passes, keys
        ctx.p_coarse, ctx.p_fine = p_coarse, p_fine
        ctx.rays_c = rays_c
        ctx.emb_dir_ray, ctx.codes_c = emb_dir_ray, codes_c
        ctx.table_shape = table.shape if table is not None else None
        ctx.n_params = len(params)
        out = tuple(results[k] for k in keys)
        ctx.mark_non_differentiable(*[results[k] for k in keys if k.startswith(("weights_", "z_vals_"))])
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        fi, vox = meta["forward_instance"], meta["use_voxel"]
        dev = ctx.rays_c.device
        n = ctx.rays_c.shape[0]
        g = dict(zip(ctx.keys, grads))
        d_table = torch.zeros(ctx.table_shape, dtype=torch.float32, device=dev) if vox else None
        d_codes = torch.zeros(n, 64, dtype=torch.float32, device=dev) if fi else None
        pk = meta.get("packed") or (None, None)
        param_grads = _flat_grad_views([ctx.p_coarse] + ([ctx.p_fine] if len(ctx.passes) > 1 else []), dev)
        for i, (typ, ps, pp, packed) in enumerate(zip(("coarse", "fine"), ctx.passes, (ctx.p_coarse, ctx.p_fine), pk)):
            gm_t = {}
            for k in _MAPS:
                t = g.get("%s_%s" % (k, typ))
                gm_t[k] = _lib.as_f32(t) if t is not None else None
            if all(v is None for v in gm_t.values()):
                continue
            _pass_backward(meta, ps, pp, packed, gm_t, param_grads[i], ctx.rays_c, ctx.emb_dir_ray, ctx.codes_c, d_table, d_codes)
        flat = list(param_grads[0]) + (list(param_grads[1]) if len(param_grads) > 1 else [])
        flat += [None] * (ctx.n_params - len(flat))
        return (None, None, d_codes, d_table, *flat)
