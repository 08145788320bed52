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
from .rendering import _linspace, mfma_mode

__all__ = ["render_rays_multi"]


def _mlp_one_branch(model, use_voxel, grid, rays, z, oid, code_library, l):
    """sigma (N,S), rgb (N,S,3) of one ray set: scene branch for id 0, object branch otherwise
    (multi_rendering.py:45-51, 63-72).

    Rays whose last depth is 0 (they missed the object's box: near = far = 0, editable_renderer.py:175-176) get
    sigma = -1e5 afterwards (multi_rendering.py:40,83,92), i.e. exactly zero weight, so their MLP evaluation cannot
    influence any output.  The reference evaluates them anyway; here they are compacted away before the kernel
    (index gather / scatter only) -- for the editing demo's object ray sets that is most of the image."""
    n_all, S = z.shape
    dev = rays.device
    active = (z[:, -1] != 0).nonzero().squeeze(1) if oid > 0 else None
    if active is not None and active.numel() < n_all:
        sigma_all = torch.full((n_all, S), -1e5, dtype=torch.float32, device=dev)
        rgb_all = torch.zeros(n_all, S, 3, dtype=torch.float32, device=dev)
        if active.numel() > 0:
            sg, c = _mlp_one_branch(model, use_voxel, grid, rays.index_select(0, active).contiguous(),
                                    z.index_select(0, active).contiguous(), oid, code_library, l)
            sigma_all.index_copy_(0, active, sg)
            rgb_all.index_copy_(0, active, c)
        return sigma_all, rgb_all
    n = n_all
    b3 = mfma_mode() == "bf16x3"
    blob, aux = model.packed(split_bf16=b3)
    a = _lib.MlpArgs()
    a.use_voxel, a.mfma_bf16x3 = int(use_voxel), int(b3)
    a.blob, a.aux = blob.data_ptr(), aux.data_ptr()
    a.rays, a.z_vals, a.n_rays, a.S = rays.data_ptr(), z.data_ptr(), n, S
    if use_voxel:
        a.grid = grid
    sigma = torch.empty(n, S, dtype=torch.float32, device=dev)
    rgb = torch.empty(n, S, 3, dtype=torch.float32, device=dev)
    code = None
    if oid > 0:
        code = _lib.as_f32(code_library.embedding_instance.weight.detach()[oid])
        a.do_scene, a.do_object = 0, 1
        a.codes, a.code_stride = code.data_ptr(), 0
        a.inst_sigma, a.inst_rgb = sigma.data_ptr(), rgb.data_ptr()
    else:
        a.do_scene, a.do_object = 1, 0
        a.sigma, a.rgb = sigma.data_ptr(), rgb.data_ptr()
    _lib.check(l.objnerf_mlp_eval(C.byref(a), _lib.stream_ptr()), "mlp_eval")
    return sigma, rgb


def _composite(l, zs, sigmas, rgbs, noise_std, white_back, want_ids, want_own, noise=None):
    K = len(zs)
    n, S = zs[0].shape
    dev = zs[0].device
    M = K * S
    out = {
        "z": torch.empty(n, M, dtype=torch.float32, device=dev),
        "w": torch.empty(n, M, dtype=torch.float32, device=dev),
        "ids": torch.empty(n, M, dtype=torch.float32, device=dev) if want_ids else None,
        "opacity": torch.empty(n, dtype=torch.float32, device=dev),
        "rgb": torch.empty(n, 3, dtype=torch.float32, device=dev),
        "depth": torch.empty(n, dtype=torch.float32, device=dev),
    }
    own = [torch.empty(n, S, dtype=torch.float32, device=dev) for _ in range(K)] if want_own else None
    a = _lib.CompositeMultiArgs()
    a.n_rays, a.K, a.S = n, K, S
    arr = C.c_void_p * K
    hz, hs, hr = arr(*[t.data_ptr() for t in zs]), arr(*[t.data_ptr() for t in sigmas]), arr(*[t.data_ptr() for t in rgbs])
    a.h_z, a.h_sigma, a.h_rgb = hz, hs, hr
    a.noise = noise.data_ptr() if noise is not None else None
    a.noise_std, a.white_back = float(noise_std), int(bool(white_back))
    a.z_sorted, a.weights = out["z"].data_ptr(), out["w"].data_ptr()
    a.obj_ids = out["ids"].data_ptr() if want_ids else None
    a.opacity, a.rgb_map, a.depth = out["opacity"].data_ptr(), out["rgb"].data_ptr(), out["depth"].data_ptr()
    if want_own:
        ho = arr(*[t.data_ptr() for t in own])
        a.h_own_weights = ho
    _lib.check(l.objnerf_composite_multi(C.byref(a), _lib.stream_ptr()), "composite_multi")
    return out, own


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
):
    assert len(rays_list) == len(obj_instance_ids)          # multi_rendering.py:179
    K = len(rays_list)
    l = _lib.lib()
    emb_xyz = embeddings["xyz"]
    use_voxel = isinstance(emb_xyz, EmbeddingVoxel)
    grid = emb_xyz.grid_struct() if use_voxel else None
    S, I = int(N_samples), int(N_importance)
    coarse = models["coarse"]
    coarse._check_no_grad(*rays_list)

    rays_c = []
    for r in rays_list:
        _lib.require_cuda(r, "rays_list entry")
        if r.dim() != 2 or r.shape[1] != 8:
            # The reference also accepts 10 columns (bbox_mask_near / far clamp the fine depths,
            # multi_rendering.py:277-285); its callers only ever build 8 (editable_renderer.py:160,177-179) and that
            # variant is not built here -- silently dropping the two columns would change the result
            raise NotImplementedError("render_rays_multi: ray sets must be (N, 8) [o, d, near, far]; got %s. The "
                                      "10-column bbox-clamp variant of the reference is not implemented."
                                      % (tuple(r.shape),))
        rays_c.append(_lib.as_f32(r))
    n = rays_c[0].shape[0]
    dev = rays_c[0].device
    if any(r.shape[0] != n for r in rays_c):
        raise RuntimeError("render_rays_multi: every ray set must have the same number of rays")
    boxes = pack_boxes(background_skip_bbox, dev) if background_skip_bbox else None
    z_steps = _linspace(S, dev)

    def masked_branch(model, rays, z, oid):
        sigma, rgb = _mlp_one_branch(model, use_voxel, grid, rays, z, int(oid), code_library, l)
        use_boxes = boxes is not None and int(oid) == 0 and boxes.shape[0] > 0        # multi_rendering.py:239-241
        _lib.check(l.objnerf_mask_sigma(_lib.ptr(sigma), _lib.ptr(rays), _lib.ptr(z), n, z.shape[1],
                                        _lib.ptr(boxes) if use_boxes else None, boxes.shape[0] if use_boxes else 0,
                                        _lib.stream_ptr()), "mask_sigma")
        return sigma, rgb

    # coarse: depths are never perturbed here (multi_rendering.py:203-210)
    zs, sgs, cs = [], [], []
    for i in range(K):
        z = torch.empty(n, S, dtype=torch.float32, device=dev)
        _lib.check(l.objnerf_sample_coarse(_lib.ptr(rays_c[i]), _lib.ptr(z_steps), None, 0.0, int(bool(use_disp)), n, S,
                                           _lib.ptr(z), _lib.stream_ptr()), "sample_coarse")
        sg, c = masked_branch(coarse, rays_c[i], z, obj_instance_ids[i])
        zs.append(z); sgs.append(sg); cs.append(c)
    nz = torch.randn(n, K * S, device=dev) if noise_std != 0 else None
    out, own = _composite(l, zs, sgs, cs, noise_std, white_back, want_ids=True, want_own=I > 0, noise=nz)
    results = {"obj_ids_coarse": out["ids"], "weights_coarse": out["w"], "opacity_coarse": out["opacity"],
               "z_vals_coarse": out["z"], "rgb_coarse": out["rgb"], "depth_coarse": out["depth"]}

    if I > 0:
        fine = models["fine"]
        det = perturb == 0
        u = _linspace(I, dev) if det else None
        zf, sf, cf = [], [], []
        for i in range(K):
            z = torch.empty(n, S + I, dtype=torch.float32, device=dev)
            ui = u if det else torch.rand(n, I, device=dev)
            _lib.check(l.objnerf_sample_pdf_merge(_lib.ptr(zs[i]), _lib.ptr(own[i]), _lib.ptr(ui), 0 if det else I, n, S, I,
                                                  1e-5, None, _lib.ptr(z), _lib.stream_ptr()), "sample_pdf_merge")
            sg, c = masked_branch(fine, rays_c[i], z, obj_instance_ids[i])
            zf.append(z); sf.append(sg); cf.append(c)
        nz = torch.randn(n, K * (S + I), device=dev) if noise_std != 0 else None
        out, _ = _composite(l, zf, sf, cf, noise_std, white_back, want_ids=False, want_own=False, noise=nz)
        results.update({"weights_fine": out["w"], "opacity_fine": out["opacity"], "z_vals_fine": out["z"],
                        "rgb_fine": out["rgb"], "depth_fine": out["depth"]})
    return results
