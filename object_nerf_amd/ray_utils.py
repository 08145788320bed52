"""On-device ray generation for the editor.

Reference: datasets/ray_utils.py:5-51 (`get_ray_directions`, `get_rays`) and
render_tools/editable_renderer.py:153-181 (`EditableRenderer.generate_rays`, whose object branch calls
the CPU numba slab test utils/bbox_utils.py:132-156 -> datasets/geo_utils.py:111-162 and copies the
result host->device per object per frame).  Here one HIP kernel writes the (H*W, 8) ray set directly:
pixel directions, rotation by the camera-to-object matrix, normalisation, and near/far either constant
(background) or from the ray/oriented-box slab test in float64 with the reference's miss rules.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .bbox import _box_row


def row_share(H, rank=0, world=1, row_block=None):
    """(row0, n_rows, row_block, block_stride) of one rank's share of an H-row image for `generate_rays(rows=...)`:
    row_block None = the contiguous band of distributed.shard_bounds; row_block = b: blocks of b rows dealt round-robin
    (block-cyclic, distributed.cyclic_rows -- the cost-balanced split of render_rays_multi_sharded)."""
    if row_block is None:
        per = (H + world - 1) // world
        lo = min(rank * per, H)
        n = min(lo + per, H) - lo
        return lo, n, max(n, 1), 1
    nblk = (H + row_block - 1) // row_block
    mine = range(rank, nblk, world)
    n = sum(min(row_block, H - b * row_block) for b in mine)
    return rank * row_block, n, row_block, world


def generate_rays(H, W, focal, c2w, near=0.0, far=0.0, box=None, bbox_enlarge=0.0, scale_factor=None, device="cuda",
                  rows=None):
    """(H*W, 8) fp32 rays `[o, d, near, far]` in row-major pixel order (rows = (row0, n_rows, row_block, block_stride)
    from `row_share`: only that subset of the image rows, (n_rows*W, 8) -- each rank of a sharded frame writes its own).

    c2w: (3,4) or (4,4) camera-to-(object-)world matrix `Toc` with the translation already divided by the
    scene scale (editable_renderer.py:252-255).  box: None for the background ray set (constant near/far,
    editable_renderer.py:156-160) or a BBoxRayHelper-like object / synth.oriented_box dict, in which case
    near/far come from the box and rays that miss it get near = far = 0 (163-179)."""
    m = np.asarray(c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float32)[:3, :4]
    c2w_h = (C.c_float * 12)(*m.reshape(-1).tolist())
    row0, n_rows, row_block, block_stride = (0, H, H, 1) if rows is None else rows
    rays = torch.empty(n_rows * W, 8, dtype=torch.float32, device=device)
    box_h = None
    if box is not None:
        row = _box_row(box, scale_factor, 0.0)          # enlargement is applied symmetrically in the kernel
        box_h = (C.c_double * _lib.BOX_DOUBLES)(*row.tolist())
    with torch.cuda.device(rays.device):
        _lib.check(_lib.lib().objnerf_generate_rays_rows(int(H), int(W), float(focal), c2w_h, float(near), float(far), box_h,
                                                         float(bbox_enlarge), int(row0), int(n_rows), int(row_block),
                                                         int(block_stride), _lib.ptr(rays), _lib.stream_ptr()), "generate_rays")
    return rays


def get_ray_directions(H, W, focal, device="cuda"):
    """datasets/ray_utils.py:5-25 on the device: (H, W, 3) camera-frame pixel directions `[(i - W/2)/f, -(j - H/2)/f, -1]`
    (no +0.5).  The reference builds them on the CPU through kornia and the editor copies them over per frame
    (editable_renderer.py:191, 215)."""
    out = torch.empty(H, W, 3, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().objnerf_ray_directions(int(H), int(W), float(focal), _lib.ptr(out), _lib.stream_ptr()), "ray_directions")
    return out


@_lib.on_device_of(lambda directions, c2w: directions)
def get_rays(directions, c2w):
    """datasets/ray_utils.py:28-51 on the device: directions (H, W, 3) (or (n, 3)), c2w (3, 4) -> rays_o (n, 3), rays_d (n, 3),
    the rotated and normalised directions and the broadcast origin.  Same signature and result shapes as the reference's
    `get_rays`; no host round trip (c2w is read from device memory; a host c2w is copied over without blocking)."""
    _lib.require_cuda(directions, "directions")
    d = _lib.as_f32(directions).reshape(-1, 3)
    if not isinstance(c2w, torch.Tensor):
        c2w = torch.as_tensor(np.asarray(c2w, dtype=np.float32))
    if c2w.dim() != 2 or c2w.shape[0] < 3 or c2w.shape[1] != 4:
        raise RuntimeError("get_rays: c2w must be (3, 4) (or the top rows of a (4, 4)); got %s" % (tuple(c2w.shape),))
    m = c2w.to(device=d.device, dtype=torch.float32, non_blocking=True)
    if m.stride(1) != 1 or m.stride(0) < 4:
        m = m.contiguous()
    n = d.shape[0]
    rays_o, rays_d = torch.empty(n, 3, dtype=torch.float32, device=d.device), torch.empty(n, 3, dtype=torch.float32, device=d.device)
    _lib.check(_lib.lib().objnerf_get_rays(_lib.ptr(d), n, C.c_void_p(m.data_ptr()), int(m.stride(0)), _lib.ptr(rays_o),
                                           _lib.ptr(rays_d), _lib.stream_ptr()), "get_rays")
    return rays_o, rays_d
