"""Reference import path `utils.bbox_utils` (editable_renderer.py:18, multi_rendering.py:12) -> device box tests.

`BBoxRayHelper` is the reference's class (its constructor and file readers, `get_world_to_object_transform`, the numpy
transforms: all inherited from the reference's utils/bbox_utils.py) with the two methods the render loop calls per object
per frame replaced:

    get_ray_bbox_intersections   utils/bbox_utils.py:132-156: rays -> host numpy -> numba slab test
                                 (datasets/geo_utils.py:111-162) -> three host-to-device copies.  Here: one HIP kernel on
                                 the rays where they are (object_nerf_amd.bbox.ray_bbox_intersections).
    check_xyz_in_bounds          utils/bbox_utils.py:158-186: points -> host numpy float64 -> device.  Here: the device
                                 point-in-box kernel (object_nerf_amd.bbox.check_in_any_boxes).

CPU inputs (nothing on the render path passes them) fall through to the reference's implementation.
"""
import os

from _objnerf_dropin import load_reference_module
from object_nerf_amd import bbox as _hip

_ref = load_reference_module("utils", "bbox_utils", os.path.dirname(os.path.abspath(__file__)))
if _ref is None:
    raise ImportError("utils.bbox_utils: the reference checkout is not on sys.path (BBoxRayHelper derives from its class)")


def _on_device(t):
    return hasattr(t, "is_cuda") and t.is_cuda


class BBoxRayHelper(_ref.BBoxRayHelper):
    def get_ray_bbox_intersections(self, rays_o, rays_d, scale_factor=None, bbox_enlarge=0):
        if not _on_device(rays_o):
            return super().get_ray_bbox_intersections(rays_o, rays_d, scale_factor, bbox_enlarge)
        return _hip.ray_bbox_intersections(self, rays_o, rays_d, scale_factor, bbox_enlarge)

    def check_xyz_in_bounds(self, xyz, scale_factor=None, bbox_enlarge=0):
        if not _on_device(xyz):
            return super().check_xyz_in_bounds(xyz, scale_factor, bbox_enlarge)
        return _hip.check_in_any_boxes({"0": self}, xyz, scale_factor, bbox_enlarge)


def check_in_any_boxes(boxes, xyz, scale_factor=None, bbox_enlarge=0.0):
    if not _on_device(xyz):
        return _ref.check_in_any_boxes(boxes, xyz, scale_factor, bbox_enlarge)
    return _hip.check_in_any_boxes(boxes, xyz, scale_factor, bbox_enlarge)
