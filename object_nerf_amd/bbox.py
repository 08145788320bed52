"""Oriented-box masks on the device (reference: utils/bbox_utils.py:119-130, 158-207).

The reference tests sample points against object boxes on the HOST (GPU -> numpy float64 ->
torch -> GPU round trip inside the render loop, multi_rendering.py:239-241).  Here the boxes are
packed once into a small float64 device array and the test runs inside the HIP kernels
(csrc/ray_kernels.hip::in_any_box), with the same float64 transform / fp32 comparison split.

Accepted box objects: anything shaped like the reference's BBoxRayHelper (attributes
scale_factor, pose_avg, axis_align_mat, bbox_bounds) or the dicts of synth.oriented_box.
"""
import copy

import numpy as np
import torch

from . import _lib


_packed = {}


def _box_row(box, scale_factor=None, bbox_enlarge=0.0):
    if isinstance(box, dict):
        sf = box["scale_factor"] if scale_factor is None else scale_factor
        R_avg, t_avg = np.asarray(box["R_avg"], dtype=np.float64), np.asarray(box["t_avg"], dtype=np.float64)
        R_box, t_box = np.asarray(box["R_box"], dtype=np.float64), np.asarray(box["t_box"], dtype=np.float64)
        bounds = np.array([np.asarray(box["bmin"], dtype=np.float64), np.asarray(box["bmax"], dtype=np.float64)])
    else:
        sf = box.scale_factor if scale_factor is None else scale_factor
        pa = np.asarray(box.pose_avg, dtype=np.float64).squeeze()
        aa = np.asarray(box.axis_align_mat, dtype=np.float64)
        R_avg, t_avg, R_box, t_box = pa[:3, :3], pa[:3, 3], aa[:3, :3], aa[:3, 3]
        bounds = copy.deepcopy(np.asarray(box.bbox_bounds, dtype=np.float64))
    # bbox_enlarge rules, bbox_utils.py:171-181
    if bbox_enlarge > 0:
        z_min = bounds[0][2]
        bounds[0] -= bbox_enlarge
        bounds[1] += bbox_enlarge
        bounds[0][2] = z_min
    elif bbox_enlarge < 0:
        bounds[0][2] -= bbox_enlarge
    return np.concatenate([[float(sf)], R_avg.reshape(-1), t_avg.reshape(-1), R_box.reshape(-1), t_box.reshape(-1),
                           bounds[0], bounds[1]])


def pack_boxes(boxes, device, scale_factor=None, bbox_enlarge=0.0):
    """dict / list of boxes -> (n_boxes, 31) float64 device tensor (include/objnerf_hip.h layout)"""
    seq = list(boxes.values()) if isinstance(boxes, dict) else list(boxes)
    if not seq:
        return torch.zeros(0, _lib.BOX_DOUBLES, dtype=torch.float64, device=device)
    rows = np.stack([_box_row(b, scale_factor, bbox_enlarge) for b in seq])
    assert rows.shape[1] == _lib.BOX_DOUBLES
    # the editor passes the same boxes with every ray chunk of a frame (editable_renderer.py:270-287): keep the device
    # copy of the last few distinct box sets instead of a blocking host-to-device copy per call
    key = (rows.tobytes(), str(device))
    hit = _packed.get(key)
    if hit is None:
        if len(_packed) >= 16:
            _packed.pop(next(iter(_packed)))
        hit = _packed[key] = torch.from_numpy(rows).to(device)
    return hit


def check_in_any_boxes(boxes, xyz, scale_factor=None, bbox_enlarge=0.0):
    """Drop-in for utils/bbox_utils.py::check_in_any_boxes (189-207): bool mask shaped like xyz[..., 0]."""
    _lib.require_cuda(xyz, "xyz")
    shp = xyz.shape[:-1]
    pts = _lib.as_f32(xyz).reshape(-1, 3)
    packed = pack_boxes(boxes, xyz.device, scale_factor, bbox_enlarge)
    out = torch.empty(pts.shape[0], dtype=torch.uint8, device=xyz.device)
    _lib.check(_lib.lib().objnerf_points_in_boxes(_lib.ptr(pts), pts.shape[0], _lib.ptr(packed), packed.shape[0],
                                                  _lib.ptr(out), _lib.stream_ptr()), "points_in_boxes")
    return out.bool().view(*shp)


@_lib.on_device_of(lambda box, rays_o, *a, **k: rays_o)
def ray_bbox_intersections(box, rays_o, rays_d, scale_factor=None, bbox_enlarge=0):
    """Drop-in for `BBoxRayHelper.get_ray_bbox_intersections` (utils/bbox_utils.py:132-156): (hit (n) bool, near (n,1),
    far (n,1)) of rays against one oriented box, near/far already divided by scale_factor, 0/0 on a miss.  The reference
    copies the rays to the host, runs a numba loop (datasets/geo_utils.py:111-162) and copies three arrays back, per object
    per frame; here it is one kernel on the rays where they are (float64 slab test, the reference's miss rules)."""
    _lib.require_cuda(rays_o, "rays_o")
    _lib.require_cuda(rays_d, "rays_d")
    o, d = _lib.as_f32(rays_o).reshape(-1, 3), _lib.as_f32(rays_d).reshape(-1, 3)
    if o.shape != d.shape:
        raise RuntimeError("ray_bbox_intersections: rays_o %s and rays_d %s differ" % (tuple(o.shape), tuple(d.shape)))
    import ctypes as C
    row = _box_row(box, scale_factor, 0.0)              # both bounds grow by bbox_enlarge in the kernel (bbox_utils.py:141-145)
    box_h = (C.c_double * _lib.BOX_DOUBLES)(*row.tolist())
    n = o.shape[0]
    hit = torch.empty(n, dtype=torch.uint8, device=o.device)
    near, far = torch.empty(n, 1, dtype=torch.float32, device=o.device), torch.empty(n, 1, dtype=torch.float32, device=o.device)
    _lib.check(_lib.lib().objnerf_ray_box_near_far(_lib.ptr(o), _lib.ptr(d), n, box_h, float(bbox_enlarge), _lib.ptr(hit),
                                                   _lib.ptr(near), _lib.ptr(far), _lib.stream_ptr()), "ray_box_near_far")
    return hit.bool(), near, far
