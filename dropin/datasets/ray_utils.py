"""Reference import path `datasets.ray_utils` (editable_renderer.py:21, generic_dataset.py:14) -> device ray generation.

    get_rays(directions, c2w)       directions on the GPU (the editor: editable_renderer.py:191-198, 215, 257) ->
                                    object_nerf_amd.ray_utils.get_rays, one HIP kernel, results stay on the device;
                                    CPU directions (the dataset: generic_dataset.py:144, 397, inside DataLoader workers,
                                    which must never touch the GPU library) -> the reference's own function.
    get_ray_directions(H, W, focal) the reference's own function (CPU): the same call site serves the dataset, which keeps
                                    the grid on the host for its workers; the editor copies it over once per frame
                                    (`.cuda()`, editable_renderer.py:191) -- 3.7 MB at 640x480, not per object.  Callers
                                    that want it made on the device call object_nerf_amd.ray_utils.get_ray_directions.
    get_ndc_rays                    the reference's own function (not on the hot path).
"""
import os

from _objnerf_dropin import load_reference_module
from object_nerf_amd import ray_utils as _hip


def _ref():
    m = load_reference_module("datasets", "ray_utils", os.path.dirname(os.path.abspath(__file__)))
    if m is None:
        raise ImportError("datasets.ray_utils: the reference checkout is not on sys.path (needed for CPU-side ray code)")
    return m


def _on_device(t):
    return t.is_cuda


def get_ray_directions(H, W, focal):
    return _ref().get_ray_directions(H, W, focal)


def get_rays(directions, c2w):
    if _on_device(directions):
        return _hip.get_rays(directions, c2w)
    return _ref().get_rays(directions, c2w)


def get_ndc_rays(*args, **kwargs):
    return _ref().get_ndc_rays(*args, **kwargs)
