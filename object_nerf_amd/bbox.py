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
