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


def generate_rays(H, W, focal, c2w, near=0.0, far=0.0, box=None, bbox_enlarge=0.0, scale_factor=None, device="cuda"):
    """(H*W, 8) fp32 rays `[o, d, near, far]` in row-major pixel order.

    c2w: (3,4) or (4,4) camera-to-(object-)world matrix `Toc` with the translation already divided by the
    scene scale (editable_renderer.py:252-255).  box: None for the background ray set (constant near/far,
    editable_renderer.py:156-160) or a BBoxRayHelper-like object / synth.oriented_box dict, in which case
    near/far come from the box and rays that miss it get near = far = 0 (163-179)."""
    m = np.asarray(c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float32)[:3, :4]
    c2w_h = (C.c_float * 12)(*m.reshape(-1).tolist())
    rays = torch.empty(H * W, 8, dtype=torch.float32, device=device)
    box_h = None
    if box is not None:
        row = _box_row(box, scale_factor, 0.0)          # enlargement is applied symmetrically in the kernel
        box_h = (C.c_double * _lib.BOX_DOUBLES)(*row.tolist())
    _lib.check(_lib.lib().objnerf_generate_rays(int(H), int(W), float(focal), c2w_h, float(near), float(far), box_h,
                                                float(bbox_enlarge), _lib.ptr(rays), _lib.stream_ptr()), "generate_rays")
    return rays
