"""Drop-in `render_rays_multi` (reference: render_tools/multi_rendering.py:160-325), the
multi-object compositor behind the editing demo (render_tools/editable_renderer.py:125-140,
272-287).

K ray sets (one per active object, id 0 = background) -> per set: coarse depths, ONE branch of the
fused MLP kernel (id 0: scene branch; id > 0: object branch with that id's code, stride-0 code
operand), sigma masks (rays with near = far = 0, background samples inside removed boxes) ->
joint z-sorted compositing -> per-set importance sampling from the set's own weights -> fine pass.
Every step is a HIP kernel (include/objnerf_hip.h); the host round trip the reference makes for the
box test (utils/bbox_utils.py:119-130) is gone.
"""
import ctypes as C
from typing import Any, Dict

import torch

from . import _lib
from .bbox import pack_boxes
from .embedding_helper import EmbeddingVoxel
from .nerf_model import pack_models
from .rendering import _linspace, hoist_enabled

__all__ = ["render_rays_multi"]


@_lib.on_device_of(lambda *a, **k: (k["rays_list"] if "rays_list" in k else a[3])[0])
def render_rays_multi(
    models: Dict[str, Any],
    embeddings: Dict[str, torch.nn.Module],
    code_library: torch.nn.Module,
    rays_list: list,
    obj_instance_ids: list,
    N_samples: int = 64,
    use_disp: bool = False,
    perturb: float = 0,
    noise_std: float = 0,
    N_importance: int = 0,
    chunk: int = 1024 * 32,
    white_back: bool = False,
    background_skip_bbox: Dict[str, Any] = None,
    _randoms: Dict[str, Any] = None,
):
    # _randoms: test hook (never set by the reference's callers) -- pre-drawn tensors of the training-mode paths,
    # {"u_rand": K x (N, I) (or one (K, N, I) tensor), "noise": [(N, K*S), (N, K*(S+I))]}, instead of drawing them here
    assert len(rays_list) == len(obj_instance_ids)          # multi_rendering.py:179
    K = len(rays_list)
    l = _lib.lib()
    emb_xyz = embeddings["xyz"]
    use_voxel = isinstance(emb_xyz, EmbeddingVoxel)
    S, I = int(N_samples), int(N_importance)
    coarse = models["coarse"]
    coarse._check_no_grad(*rays_list)

    rays_c, clips = [], []
    for r in rays_list:
        _lib.require_cuda(r, "rays_list entry")
        if r.dim() != 2 or r.shape[1] not in (8, 10):
            raise RuntimeError("render_rays_multi: ray sets must be (N, 8) [o, d, near, far] or (N, 10) "
                               "[..., bbox_mask_near, bbox_mask_far]; got %s" % (tuple(r.shape),))
        r32 = _lib.as_f32(r)
        # 10 columns: the last two clip the fine depths (multi_rendering.py:277-285); the kernels read (N, 8) rows
        rays_c.append(r32 if r.shape[1] == 8 else r32[:, :8].contiguous())
        clips.append(None if r.shape[1] == 8 else r32[:, 8:10].contiguous())
    n = rays_c[0].shape[0]
    dev = rays_c[0].device
    if any(r.shape[0] != n for r in rays_c):
        raise RuntimeError("render_rays_multi: every ray set must have the same number of rays")
    boxes = pack_boxes(background_skip_bbox, dev) if background_skip_bbox else None
    ids = [int(i) for i in obj_instance_ids]
    table = None
    if any(i != 0 for i in ids):          # the code table is only read for object sets (multi_rendering.py:45-51)
        w = code_library.embedding_instance.weight.detach()
        _lib.require_cuda(w, "code_library.embedding_instance.weight")
        if w.device != dev:
            raise RuntimeError("render_rays_multi: the code library is on %s, the rays on %s" % (w.device, dev))
        table = _lib.as_f32(w)
        code_c = int(getattr(coarse, "N_obj_code_length", 64))
        if any(i < 0 or i >= table.shape[0] for i in ids) or table.shape[1] != code_c:
            raise RuntimeError("render_rays_multi: object ids must index the (N_max_objs, %d) code table" % code_c)
    elif any(i < 0 for i in ids):
        raise RuntimeError("render_rays_multi: negative object id")
    # limits, raised before anything is enqueued (objnerf_render_rays_multi checks them too).  The joint compositing itself
    # has no sample limit since round 4 (LDS staging up to K * (S + I) = 5,558, workspace staging beyond)
    if K > 64:
        raise RuntimeError("render_rays_multi: %d ray sets (at most 64 per call)" % K)
    if I > 0 and (S < 3 or S > 1025 or S + I > 2048):
        raise RuntimeError("render_rays_multi: the importance sampler takes 3 <= N_samples <= 1025 and N_samples + N_importance "
                           "<= 2048 per ray set; got %d + %d" % (S, I))

    from .rendering import fused_path_ok
    if not fused_path_ok(models, embeddings, I > 0):
        # any architecture other than the shipped default: the same pipeline stage by stage (object_nerf_amd/generic.py)
        from . import generic
        u_rand = noise = None
        if I > 0 and perturb != 0:
            u = _randoms["u_rand"] if (_randoms and "u_rand" in _randoms) else torch.rand(K, n, I, device=dev)
            u_rand = [_lib.as_f32(t.to(dev)) for t in u]
        if noise_std != 0:
            pre = _randoms.get("noise") if _randoms else None
            noise = [_lib.as_f32(pre[0].to(dev)) if pre else torch.randn(n, K * S, device=dev),
                     (_lib.as_f32(pre[1].to(dev)) if pre else torch.randn(n, K * (S + I), device=dev)) if I > 0 else None]
        oc, of = generic.render_rays_multi(models, embeddings, table, rays_c, clips, ids, S, I, use_disp, perturb, noise_std,
                                           white_back, boxes, _linspace(S, dev), _linspace(I, dev) if I > 0 else None, u_rand, noise)
        results = {"obj_ids_coarse": oc["obj_ids"], "weights_coarse": oc["weights"], "opacity_coarse": oc["opacity"],
                   "z_vals_coarse": oc["z_vals"], "rgb_coarse": oc["rgb"], "depth_coarse": oc["depth"]}
        if of is not None:
            results.update({"weights_fine": of["weights"], "opacity_fine": of["opacity"], "z_vals_fine": of["z_vals"],
                            "rgb_fine": of["rgb"], "depth_fine": of["depth"]})
        return results

    cfg = _lib.RenderMultiCfg(use_voxel=int(use_voxel), N_samples=S, N_importance=I, use_disp=int(bool(use_disp)),
                              perturb=float(perturb), noise_std=float(noise_std), white_back=int(bool(white_back)),
                              no_hoist=int(not hoist_enabled()))
    ws = torch.empty(l.objnerf_render_multi_workspace_bytes(C.byref(cfg), K, n), dtype=torch.uint8, device=dev)
    rin = _lib.RenderMultiIn()
    rin.n_rays, rin.K = n, K
    h_rays = (C.c_void_p * K)(*[r.data_ptr() for r in rays_c])
    h_ids = (C.c_int32 * K)(*ids)
    rin.h_rays, rin.h_obj_ids = h_rays, h_ids
    if table is not None:
        rin.code_table = table.data_ptr()
    packs = pack_models([coarse] + ([models["fine"]] if I > 0 else []))        # one launch, nothing cached
    rin.blob_coarse, rin.aux_coarse = packs[0][0].data_ptr(), packs[0][1].data_ptr()
    keep = [rays_c, table, ws, packs, boxes]
    if I > 0:
        rin.blob_fine, rin.aux_fine = packs[1][0].data_ptr(), packs[1][1].data_ptr()
        rin.u_det = _linspace(I, dev).data_ptr()
        if perturb != 0:                      # sample_pdf(det=False) draws torch.rand per set (rendering.py:40)
            if _randoms and "u_rand" in _randoms:
                u = _randoms["u_rand"]
                ur = _lib.as_f32(torch.stack([t.to(dev) for t in u]) if isinstance(u, (list, tuple)) else u.to(dev))
                if tuple(ur.shape) != (K, n, I):
                    raise RuntimeError("render_rays_multi: _randoms['u_rand'] must be (K, N, N_importance)")
            else:
                ur = torch.rand(K, n, I, device=dev)
            rin.u_rand = ur.data_ptr()
            keep.append(ur)
    if use_voxel:
        rin.grid = emb_xyz.grid_struct()
    rin.z_steps = _linspace(S, dev).data_ptr()
    if noise_std != 0:                        # multi_rendering.py:126: one randn per compositing
        pre = _randoms.get("noise") if _randoms else None
        nzc = _lib.as_f32(pre[0].to(dev)) if pre else torch.randn(n, K * S, device=dev)
        rin.noise_coarse = nzc.data_ptr()
        keep.append(nzc)
        if I > 0:
            nzf = _lib.as_f32(pre[1].to(dev)) if pre else torch.randn(n, K * (S + I), device=dev)
            rin.noise_fine = nzf.data_ptr()
            keep.append(nzf)
    if boxes is not None and boxes.shape[0] > 0:
        rin.boxes, rin.n_boxes = boxes.data_ptr(), boxes.shape[0]
    if any(c is not None for c in clips):
        h_clip = (C.c_void_p * K)(*[c.data_ptr() if c is not None else None for c in clips])
        rin.h_clip = h_clip
        keep += [clips, h_clip]
    rin.workspace = ws.data_ptr()

    def alloc(M, want_ids):
        o = {"z_vals": torch.empty(n, M, dtype=torch.float32, device=dev),
             "weights": torch.empty(n, M, dtype=torch.float32, device=dev),
             "opacity": torch.empty(n, dtype=torch.float32, device=dev),
             "depth": torch.empty(n, dtype=torch.float32, device=dev),
             "rgb": torch.empty(n, 3, dtype=torch.float32, device=dev)}
        if want_ids:
            o["obj_ids"] = torch.empty(n, M, dtype=torch.float32, device=dev)
        st = _lib.RenderMultiOut()
        for k, t in o.items():
            setattr(st, k, t.data_ptr())
        return o, st

    oc, so_c = alloc(K * S, True)
    of, so_f = alloc(K * (S + I), False) if I > 0 else (None, None)
    # ONE enqueue for the whole call: depths, ray culling, 2K launches of the fused MLP kernel, masks, compositing and
    # importance sampling are issued back to back on the stream; nothing is read back by the host in between
    _lib.check(l.objnerf_render_rays_multi(C.byref(cfg), C.byref(rin), C.byref(so_c),
                                           C.byref(so_f) if so_f is not None else None, _lib.stream_ptr()), "render_rays_multi")
    results = {"obj_ids_coarse": oc["obj_ids"], "weights_coarse": oc["weights"], "opacity_coarse": oc["opacity"],
               "z_vals_coarse": oc["z_vals"], "rgb_coarse": oc["rgb"], "depth_coarse": oc["depth"]}
    if of is not None:
        results.update({"weights_fine": of["weights"], "opacity_fine": of["opacity"], "z_vals_fine": of["z_vals"],
                        "rgb_fine": of["rgb"], "depth_fine": of["depth"]})
    return results
