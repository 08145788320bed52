"""Drop-in `render_rays` / `sample_pdf` (reference: models/rendering.py:233-337, 11-61).

One call = one enqueue of objnerf_render_rays (csrc/api.hip): coarse depths -> fused
embed+MLP kernel -> scene/instance compositing -> inverse-CDF sampling + merge -> fine pass.
The Python below only validates arguments, allocates the result tensors and draws the random
tensors the training-mode paths need (perturb > 0, noise_std > 0); it never computes.

Signature, keyword plumbing (`is_eval` / `use_zero_as_last_delta` arrive via **dummy_kwargs,
rendering.py:75-76), result keys and quirks are the reference's:
  * `embedding_instance` is mandatory even with forward_instance=False (rendering.py:94);
  * `chunk` is accepted and ignored (the kernel is persistent; nothing is chunked for memory);
  * with rays_in_bbox the returned `weights_*` are the instance weights (rendering.py:228-229).
"""
import ctypes as C
import os
from typing import Any, Dict, Optional

import torch

from . import _lib
from .embedding_helper import Embedding, EmbeddingVoxel
from .nerf_model import pack_models

__all__ = ["render_rays", "sample_pdf"]

_tables = {}


def _linspace(n, device):
    """torch.linspace(0,1,n) on the device -- the reference's own table (it is not i/(n-1):
    30 of 64 entries differ by an ulp), cached per (n, device).
    Built on the CPU so that the values are the ones the CPU reference uses too."""
    key = (n, str(device))
    if key not in _tables:
        _tables[key] = torch.linspace(0, 1, n).to(device)
    return _tables[key]


@_lib.on_device_of(lambda bins, *a, **k: bins)
def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, u=None):
    """models/rendering.py:11-61.  `u` (N_rays, N_importance) optionally injects the uniform draws
    used when det=False (tests); otherwise they are drawn with torch.rand like the reference."""
    _lib.require_cuda(bins, "bins")
    n, nb = bins.shape
    assert weights.shape == (n, nb - 1)
    b, w = _lib.as_f32(bins), _lib.as_f32(weights.detach())
    if det:
        uu, stride = _linspace(N_importance, bins.device), 0
    else:
        uu = _lib.as_f32(u) if u is not None else torch.rand(n, N_importance, device=bins.device)
        stride = N_importance
    out = torch.empty(n, N_importance, dtype=torch.float32, device=bins.device)
    _lib.check(_lib.lib().objnerf_sample_pdf(_lib.ptr(b), _lib.ptr(w), _lib.ptr(uu), stride, n, nb, N_importance,
                                             eps, _lib.ptr(out), _lib.stream_ptr()), "sample_pdf")
    return out


def _alloc_out(n, s, dev, inst):
    o = {
        "weights": torch.empty(n, s, dtype=torch.float32, device=dev),
        "z_vals": torch.empty(n, s, dtype=torch.float32, device=dev),
        "opacity": torch.empty(n, dtype=torch.float32, device=dev),
        "depth": torch.empty(n, dtype=torch.float32, device=dev),
        "rgb": torch.empty(n, 3, dtype=torch.float32, device=dev),
    }
    if inst:
        o["rgb_instance"] = torch.empty(n, 3, dtype=torch.float32, device=dev)
        o["depth_instance"] = torch.empty(n, dtype=torch.float32, device=dev)
        o["opacity_instance"] = torch.empty(n, dtype=torch.float32, device=dev)
    return o


def _out_struct(o):
    s = _lib.RenderOut()
    for k, t in o.items():
        setattr(s, k, t.data_ptr())
    return s


def composite_mode():
    """Where a pass without occlusion mask / noise composites: "fused" (default; in the MLP kernel's epilogue, sigma / rgb
    never written, include/objnerf_hip.h objnerf_mlp_args.comp_*) or "separate" (MLP kernel, then objnerf_composite).
    Bit-equal results; the switch (environment variable OBJNERF_COMPOSITE) exists for that test and for A/B timing."""
    m = os.environ.get("OBJNERF_COMPOSITE", "fused")
    if m not in ("fused", "separate"):
        raise RuntimeError("OBJNERF_COMPOSITE must be 'fused' or 'separate', got %r" % m)
    return m


def hoist_enabled():
    """Inference passes take the terms that are constant along a ray (the object code's and the direction embedding's
    share of four layers) from per-ray vectors instead of contracting them per sample point (include/objnerf_hip.h,
    objnerf_mlp_args.ray_bias).  OBJNERF_HOIST=0 switches that off (A/B timing, tolerance tests)."""
    m = os.environ.get("OBJNERF_HOIST", "1")
    if m not in ("0", "1"):
        raise RuntimeError("OBJNERF_HOIST must be '0' or '1', got %r" % m)
    return m == "1"


def train_nodes():
    """How many autograd nodes the differentiable render_rays records (object_nerf_amd/autograd.py): 2 = coarse pass and fine pass
    (the fine model's gradients are final when the fine node's backward returns, so a data-parallel wrapper exchanges them while
    the coarse node's backward runs), 1 = one node for the whole call (rounds 1-5).  Same launches, same gradients (bit-equal,
    tests/test_gpu_train.py).  OBJNERF_TRAIN_NODES = 1 | 2 forces a form; unset: 2 when a torch.distributed process group with
    more than one rank is up (the only place the split buys anything), else 1."""
    m = os.environ.get("OBJNERF_TRAIN_NODES", "auto")
    if m in ("1", "2"):
        return int(m)
    if m != "auto":
        raise RuntimeError("OBJNERF_TRAIN_NODES must be '1', '2' or 'auto', got %r" % m)
    import torch.distributed as dist
    return 2 if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else 1


def fused_path_ok(models, embeddings, fine):
    """True when the persistent fused kernel can render this operator set: the default architecture (both models), 2^k
    frequency bands with the default counts, and -- in voxel mode -- the 16 + 8 channel / 6 frequency table layout.
    Everything else takes the layer-wise path (object_nerf_amd/generic.py)."""
    if os.environ.get("OBJNERF_PATH", "fused") == "layerwise":      # developer switch: the default architecture through the
        return False                                                   # layer-wise path too (cross-check, tools/arch_bench.py)
    ms = [models["coarse"]] + ([models["fine"]] if fine else [])
    if not all(getattr(m, "fused_architecture", False) for m in ms):
        return False
    ex, ed = embeddings["xyz"], embeddings.get("dir")
    def plain(e, nf):      # 2^k bands, the default count (modules unpickled from before `logscale` existed are logscale)
        return isinstance(e, Embedding) and getattr(e, "logscale", True) and e.N_freqs == nf and e.in_channels == 3
    if isinstance(ex, EmbeddingVoxel):
        if not getattr(ex, "fused_layout", ex.channels == 24 and ex.embedding_final.N_freqs == 6):
            return False
    elif not plain(ex, 10):
        return False
    return plain(ed, 4)


def _train_packs(coarse, fine):
    mode = os.environ.get("OBJNERF_TRAIN_LAYERWISE", "")
    if mode == "1":
        return None

    def one(m):
        if m is None:
            return None
        blob, aux = m.packed()          # gathered from the parameters as they are now (nothing is cached)
        fwd_blob = None if mode == "fwd" else blob
        # voxel mode with fused forward AND fused backward (the default): the backward's stream carries the embedding-column blocks and
        # the chain kernel forms the embedding gradients itself (round 6; OBJNERF_BWD_DX=0: the two GEMMs after the chain, rounds 2-5).
        # It rides on the mask-fed chain, i.e. on the masks the fused forward leaves.
        dx = (bool(m.use_voxel_embedding) and fwd_blob is not None and mode != "bwd" and os.environ.get("OBJNERF_BWD_DX", "1") != "0"
              and os.environ.get("OBJNERF_BWD_MASKS", "1") != "0")
        return (fwd_blob, aux, None if mode == "bwd" else m.packed_bwd(dx=dx), dx)
    return (one(coarse), one(fine))


@_lib.on_device_of(lambda *a, **k: k["rays"] if "rays" in k else a[2])
def render_rays(
    models: Dict[str, Any],
    embeddings: Dict[str, Any],
    rays: torch.Tensor,
    N_samples: int = 64,
    use_disp: bool = False,
    perturb: float = 0,
    noise_std: float = 1,
    N_importance: int = 0,
    chunk: int = 1024 * 32,
    white_back: bool = False,
    forward_instance: bool = True,
    embedding_instance: Optional[torch.Tensor] = None,
    frustum_bound_th: float = 0,
    pass_through_mask: Optional[torch.Tensor] = None,
    rays_in_bbox: bool = False,
    **dummy_kwargs,
):
    is_eval = bool(dummy_kwargs.get("is_eval", False))
    use_zero_as_last_delta = bool(dummy_kwargs.get("use_zero_as_last_delta", False))
    # test hook: pre-drawn random tensors {"perturb_rand","u_rand","noise":[4]} (never set by the reference callers)
    randoms = dummy_kwargs.get("_randoms", None)

    if embedding_instance is None:
        raise TypeError("render_rays: embedding_instance is required (models/rendering.py:94 repeats it unconditionally)")
    coarse = models["coarse"]
    _lib.require_cuda(rays, "rays")
    dev = rays.device
    n = rays.shape[0]
    S, I = int(N_samples), int(N_importance)
    emb_xyz = embeddings["xyz"]
    use_voxel = isinstance(emb_xyz, EmbeddingVoxel)
    if use_voxel != bool(coarse.use_voxel_embedding):
        raise RuntimeError("render_rays: embeddings['xyz'] and the model disagree about use_voxel_embedding")
    if not use_voxel and not isinstance(emb_xyz, Embedding):
        raise TypeError("render_rays: embeddings['xyz'] must be object_nerf_amd Embedding or EmbeddingVoxel")

    rays_c = _lib.as_f32(rays)
    if rays_c.shape[1] != 8:
        rays_c = rays_c[:, :8].contiguous()
    code_c = int(getattr(coarse, "N_obj_code_length", 64))
    if tuple(embedding_instance.shape) != (n, code_c):
        raise RuntimeError("embedding_instance must be (N_rays, %d), got %s" % (code_c, tuple(embedding_instance.shape)))

    # ---- any architecture other than the shipped default: the same pipeline stage by stage (generic.py) ----
    if not fused_path_ok(models, embeddings, I > 0):
        from . import generic
        gtable = emb_xyz.embedding_space_ftr.weight if use_voxel else None
        gplist = generic.train_param_list(coarse) + (generic.train_param_list(models["fine"]) if I > 0 else [])
        wants_grad = torch.is_grad_enabled() and (embedding_instance.requires_grad or any(p.requires_grad for p in gplist)
                                                  or (gtable is not None and gtable.requires_grad))
        from . import generic
        rnd = dict(randoms) if randoms else {}
        if perturb > 0:
            rnd.setdefault("perturb_rand", torch.rand(n, S, device=dev))
            if I > 0:
                rnd.setdefault("u_rand", torch.rand(n, I, device=dev))
        if noise_std != 0 and "noise" not in rnd:
            rnd["noise"] = [torch.randn(n, S, device=dev), torch.randn(n, S, device=dev),
                            torch.randn(n, S + I, device=dev), torch.randn(n, S + I, device=dev)]
        rnd = {k: ([_lib.as_f32(t) for t in v] if isinstance(v, (list, tuple)) else _lib.as_f32(v)) for k, v in rnd.items()}
        if wants_grad:
            # training a non-default architecture: the same stages with every layer's activations kept + their backward
            meta = dict(S=S, I=I, use_voxel=use_voxel, forward_instance=bool(forward_instance), use_disp=bool(use_disp),
                        perturb=float(perturb), noise_std=float(noise_std), white_back=bool(white_back), is_eval=is_eval,
                        use_zero_as_last_delta=use_zero_as_last_delta, frustum_bound_th=float(frustum_bound_th),
                        rays_in_bbox=bool(rays_in_bbox), randoms=rnd, z_steps=_linspace(S, dev),
                        u_det=_linspace(I, dev) if I > 0 else None, grid=emb_xyz.grid_struct() if use_voxel else None,
                        ptm=pass_through_mask.reshape(n).to(torch.uint8).contiguous() if pass_through_mask is not None else None,
                        models=(coarse, models["fine"] if I > 0 else None), emb_xyz=emb_xyz, emb_dir=embeddings["dir"])
            outs = generic.RenderRaysGenericFn.apply(meta, rays_c, embedding_instance, gtable, *gplist)
            keys = sorted(["%s_%s" % (k, t) for t in (("coarse", "fine") if I > 0 else ("coarse",))
                           for k in (["weights", "opacity", "z_vals", "rgb", "depth"]
                                     + (["rgb_instance", "depth_instance", "opacity_instance"] if forward_instance else []))])
            return dict(zip(keys, outs))
        flags = dict(use_disp=bool(use_disp), perturb=float(perturb), noise_std=float(noise_std), white_back=bool(white_back),
                     forward_instance=bool(forward_instance), is_eval=is_eval, use_zero_as_last_delta=use_zero_as_last_delta,
                     frustum_bound_th=float(frustum_bound_th), rays_in_bbox=bool(rays_in_bbox),
                     pass_through_mask=pass_through_mask.reshape(n).to(torch.uint8).contiguous() if pass_through_mask is not None else None)
        oc, of = generic.render_rays(models, embeddings, rays_c, _lib.as_f32(embedding_instance.detach()), S, I, flags, rnd,
                                     _linspace(S, dev), _linspace(I, dev) if I > 0 else None,
                                     lambda n_, s_: _alloc_out(n_, s_, dev, forward_instance))
        results = {}
        for typ, o in (("coarse", oc), ("fine", of)):
            if o is None:
                continue
            for k in ("weights", "opacity", "z_vals", "rgb", "depth") + (("rgb_instance", "depth_instance", "opacity_instance") if forward_instance else ()):
                results["%s_%s" % (k, typ)] = o[k]
        return results

    # ---- training: autograd is recording and something on the path wants a gradient -> differentiable path ----
    table = emb_xyz.embedding_space_ftr.weight if use_voxel else None
    plist = list(coarse._param_list()) + (list(models["fine"]._param_list()) if I > 0 else [])
    if torch.is_grad_enabled() and (embedding_instance.requires_grad or any(p.requires_grad for p in plist)
                                    or (table is not None and table.requires_grad)):
        from .autograd import RenderRaysFn, render_rays_nodes
        rnd = dict(randoms) if randoms else {}
        if perturb > 0:
            rnd.setdefault("perturb_rand", torch.rand(n, S, device=dev))
            if I > 0:
                rnd.setdefault("u_rand", torch.rand(n, I, device=dev))
        if noise_std != 0 and "noise" not in rnd:
            rnd["noise"] = [torch.randn(n, S, device=dev), torch.randn(n, S, device=dev),
                            torch.randn(n, S + I, device=dev), torch.randn(n, S + I, device=dev)]
        for k in ("perturb_rand", "u_rand"):
            if k in rnd:
                rnd[k] = _lib.as_f32(rnd[k])
        if "noise" in rnd:
            rnd["noise"] = [_lib.as_f32(t) for t in rnd["noise"]]
        meta = dict(S=S, I=I, use_voxel=use_voxel, forward_instance=bool(forward_instance), use_disp=bool(use_disp),
                    perturb=float(perturb), noise_std=float(noise_std), white_back=bool(white_back), is_eval=is_eval,
                    use_zero_as_last_delta=use_zero_as_last_delta, frustum_bound_th=float(frustum_bound_th),
                    rays_in_bbox=bool(rays_in_bbox), randoms=rnd, z_steps=_linspace(S, dev),
                    u_det=_linspace(I, dev) if I > 0 else None, grid=emb_xyz.grid_struct() if use_voxel else None,
                    ptm=pass_through_mask.reshape(n).to(torch.uint8).contiguous() if pass_through_mask is not None else None,
                    # packed weight streams (forward, aux, transposed hidden blocks): forward and the hidden dgrad chain run on
                    # the persistent MFMA kernels.  OBJNERF_TRAIN_LAYERWISE = 1 | fwd | bwd keeps the layer-by-layer GEMM
                    # version of both / the forward / the dgrad chain instead, "mem" makes the forward kernel read the
                    # materialised embeddings back (same workspace layout in every mode; developer A/B switch)
                    packed=_train_packs(coarse, models["fine"] if I > 0 else None))
        if train_nodes() == 1:       # the single autograd node of rounds 1-5
            outs = RenderRaysFn.apply(meta, rays_c, embedding_instance, table, *plist)
            keys = sorted(["%s_%s" % (k, t) for t in (("coarse", "fine") if I > 0 else ("coarse",))
                           for k in (["weights", "opacity", "z_vals", "rgb", "depth"]
                                     + (["rgb_instance", "depth_instance", "opacity_instance"] if forward_instance else []))])
            return dict(zip(keys, outs))
        # two nodes (coarse pass, fine pass): the fine model's gradients are final when the fine node's backward returns, a
        # data-parallel wrapper exchanges them while the coarse node's backward runs (object_nerf_amd/autograd.py)
        npar = len(coarse._param_list())
        return render_rays_nodes(meta, rays_c, embedding_instance, table, plist[:npar], plist[npar:])

    codes = _lib.as_f32(embedding_instance.detach())

    cfg = _lib.RenderCfg(
        use_voxel=int(use_voxel), N_samples=S, N_importance=I, use_disp=int(bool(use_disp)),
        perturb=float(perturb), noise_std=float(noise_std), white_back=int(bool(white_back)),
        forward_instance=int(bool(forward_instance)), is_eval=int(is_eval),
        use_zero_as_last_delta=int(use_zero_as_last_delta), frustum_bound_th=float(frustum_bound_th),
        rays_in_bbox=int(bool(rays_in_bbox)),
        separate_composite=int(composite_mode() == "separate"), no_hoist=int(not hoist_enabled()))
    l = _lib.lib()
    ws = torch.empty(l.objnerf_render_workspace_bytes(C.byref(cfg), n), dtype=torch.uint8, device=dev)

    rin = _lib.RenderIn()
    rin.rays, rin.n_rays = rays_c.data_ptr(), n
    rin.codes, rin.code_stride = codes.data_ptr(), 64
    keep = [rays_c, codes, ws]
    if pass_through_mask is not None:
        ptm = pass_through_mask.reshape(n).to(torch.uint8).contiguous()
        rin.pass_through_mask = ptm.data_ptr()
        keep.append(ptm)
    # both models' weight streams, gathered from the parameters as they are now, in one launch (nothing is cached)
    packs = pack_models([coarse] + ([models["fine"]] if I > 0 else []))
    keep.append(packs)
    rin.blob_coarse, rin.aux_coarse = packs[0][0].data_ptr(), packs[0][1].data_ptr()
    if I > 0:
        rin.blob_fine, rin.aux_fine = packs[1][0].data_ptr(), packs[1][1].data_ptr()
    if use_voxel:
        rin.grid = emb_xyz.grid_struct()
    rin.z_steps = _linspace(S, dev).data_ptr()
    if I > 0:
        rin.u_det = _linspace(I, dev).data_ptr()
    # random draws (training mode only), same distributions as rendering.py:276, 40, 156, 187
    if perturb > 0:
        # the CONVERTED tensors are the ones kept alive until the launch is enqueued: as_f32 may copy
        pr = _lib.as_f32(randoms["perturb_rand"] if randoms else torch.rand(n, S, device=dev))
        keep.append(pr)
        rin.perturb_rand = pr.data_ptr()
        if I > 0:
            ur = _lib.as_f32(randoms["u_rand"] if randoms else torch.rand(n, I, device=dev))
            keep.append(ur)
            rin.u_rand = ur.data_ptr()
    if noise_std != 0:
        shapes = [(n, S), (n, S), (n, S + I), (n, S + I)]
        for k in range(4 if I > 0 else 2):
            if not forward_instance and (k & 1):
                continue
            nz = _lib.as_f32(randoms["noise"][k] if randoms else torch.randn(*shapes[k], device=dev))
            keep.append(nz)
            rin.noise[k] = nz.data_ptr()
    rin.workspace = ws.data_ptr()

    oc = _alloc_out(n, S, dev, forward_instance)
    of = _alloc_out(n, S + I, dev, forward_instance) if I > 0 else None
    so_c = _out_struct(oc)
    so_f = _out_struct(of) if of is not None else None
    _lib.check(l.objnerf_render_rays(C.byref(cfg), C.byref(rin), C.byref(so_c),
                                     C.byref(so_f) if so_f is not None else None, _lib.stream_ptr()), "render_rays")

    results = {}
    for typ, o in (("coarse", oc), ("fine", of)):
        if o is None:
            continue
        results["weights_%s" % typ] = o["weights"]
        results["opacity_%s" % typ] = o["opacity"]
        results["z_vals_%s" % typ] = o["z_vals"]
        results["rgb_%s" % typ] = o["rgb"]
        results["depth_%s" % typ] = o["depth"]
        if forward_instance:
            results["rgb_instance_%s" % typ] = o["rgb_instance"]
            results["depth_instance_%s" % typ] = o["depth_instance"]
            results["opacity_instance_%s" % typ] = o["opacity_instance"]
    return results
