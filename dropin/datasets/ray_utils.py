"""Reference import path `datasets.ray_utils` (editable_renderer.py:21, generic_dataset.py:14) -> device ray generation.

    get_rays(directions, c2w)       directions on the GPU (the editor: editable_renderer.py:191-198, 215, 257) ->
                                    object_nerf_amd.ray_utils.get_rays, one HIP kernel, results stay on the device;
                                    CPU directions (the dataset: generic_dataset.py:144, 397, inside DataLoader workers,
                                    which must never touch the GPU library) -> the reference's own function.
    get_ray_directions(H, W, focal) the same call site serves the dataset -- which keeps the grid on the host for its
                                    DataLoader workers (generic_dataset.py:144, 397) -- and the editor, which moves it to
                                    the GPU once per frame (`get_ray_directions(h, w, focal).cuda()`,
                                    editable_renderer.py:191, 215).  The result is therefore the reference's own CPU grid,
                                    wrapped in a tensor type whose `.cuda()` / `.to("cuda")` does not copy: it has the
                                    grid WRITTEN on the device by objnerf_ray_directions (bit-equal to the host grid:
                                    the same two IEEE operations per component).  Every other use of the tensor is a
                                    plain CPU tensor's.
    get_ndc_rays                    the reference's own function (not on the hot path).
"""
import os

import torch

from _objnerf_dropin import load_reference_module
from object_nerf_amd import ray_utils as _hip


def _ref():
    m = load_reference_module("datasets", "ray_utils", os.path.dirname(os.path.abspath(__file__)))
    if m is None:
        raise ImportError("datasets.ray_utils: the reference checkout is not on sys.path (needed for CPU-side ray code)")
    return m


def _on_device(t):
    return t.is_cuda


class HostDirections(torch.Tensor):
    """the reference's (H, W, 3) CPU direction grid; moving it to a GPU generates it there instead of copying it"""
    __torch_function__ = torch._C._disabled_torch_function_impl      # results of any op on it are plain tensors

    @staticmethod
    def wrap(t, H, W, focal):
        d = torch.Tensor._make_subclass(HostDirections, t)
        d._hwf = (int(H), int(W), float(focal))
        d._version_at_wrap = t._version          # an in-place edit of the host grid afterwards must travel with it (real copy)
        return d

    def _pristine(self):
        return self._version == getattr(self, "_version_at_wrap", -1)

    def _device_grid(self, device):
        H, W, focal = self._hwf
        return _hip.get_ray_directions(H, W, focal, device=device)

    def cuda(self, device=None, non_blocking=False, **kw):
        if kw or not self._pristine():           # memory_format etc., or an edited grid: torch's own copy
            return torch.Tensor.cuda(self.as_subclass(torch.Tensor), device, non_blocking, **kw)
        return self._device_grid("cuda" if device is None else (torch.device("cuda", device) if isinstance(device, int) else device))

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args and isinstance(args[0], (str, torch.device, int)) else None)
        dtype = kwargs.get("dtype", next((a for a in args if isinstance(a, torch.dtype)), None))
        plain = not (set(kwargs) - {"device", "dtype", "non_blocking"}) and not any(isinstance(a, (bool, torch.memory_format)) for a in args)
        if (dev is not None and torch.device(dev).type == "cuda" and dtype in (None, torch.float32) and plain and self._pristine()
                and not kwargs.get("copy", False)):
            return self._device_grid(torch.device(dev))
        return torch.Tensor.to(self.as_subclass(torch.Tensor), *args, **kwargs)


def get_ray_directions(H, W, focal):
    return HostDirections.wrap(_ref().get_ray_directions(H, W, focal), H, W, focal)


def get_rays(directions, c2w):
    if _on_device(directions):
        return _hip.get_rays(directions, c2w)
    return _ref().get_rays(directions, c2w)


def get_ndc_rays(*args, **kwargs):
    return _ref().get_ndc_rays(*args, **kwargs)
