"""Import harness for the REAL reference (zju3dv/object_nerf at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  This module is used in the build container (where
/root/reference is mounted) for two things:

  * tools that GENERATE the committed golden vectors (oracle/make_golden.py), and
  * `-m "not gpu"` tests that pin the oracle restatement (oracle/objnerf_oracle.py)
    against the reference itself, when the mount is present.

It never travels to the GPU box (there is no /root/reference there) and nothing in
the product package (object_nerf_amd/) may import it.

What it does (SURVEY.md §8c recipe, nothing in the reference tree is modified):
  * inserts empty stub modules for `torch_optimizer`, `open3d` (with
    io.read_point_cloud), `ipdb` so `utils/__init__.py:5`, `utils/util.py:12`,
    `render_tools/multi_rendering.py:1` import;
  * pre-registers a stub `utils.bbox_utils` exposing `check_in_any_boxes`
    (the real file drags in cv2/numba/kornia via datasets/__init__.py);
  * makes `Tensor.cuda()` / `Module.cuda()` identity when no GPU is visible, because
    `models/embedding_helper.py:103,125,163,166,193,200,367` hard-code `.cuda()`.
"""
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("OBJNERF_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "rendering.py"))


class _PointCloud:
    def __init__(self, pts):
        self.points = np.asarray(pts, dtype=np.float64)


# path -> (M,3) float64 array; lets tests hand a synthetic cloud to the reference's
# own EmbeddingVoxel.set_pointclouds (embedding_helper.py:86-94) without a .ply file
POINT_CLOUDS = {}


def _read_point_cloud(path):
    if path in POINT_CLOUDS:
        return _PointCloud(POINT_CLOUDS[path])
    raise FileNotFoundError(path)


_loaded = None


def load_reference():
    """Returns a namespace with the reference's hot-path symbols."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)

    # the reference's top-level packages are called `models`, `utils`, `render_tools`:
    # make sure no foreign module of that name is already imported
    for name in ("models", "utils", "render_tools"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(REF_ROOT):
            raise RuntimeError("module %r already imported from elsewhere" % name)

    for name in ("torch_optimizer", "ipdb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "open3d" not in sys.modules:
        o3d = types.ModuleType("open3d")
        o3d.io = types.ModuleType("open3d.io")
        o3d.io.read_point_cloud = _read_point_cloud
        sys.modules["open3d"] = o3d
        sys.modules["open3d.io"] = o3d.io

    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self

    cwd = os.getcwd()
    sys.path.insert(0, REF_ROOT)
    try:
        import utils  # noqa: F401  (real package; needs the stubs above)

        bb = types.ModuleType("utils.bbox_utils")

        def check_in_any_boxes(boxes, xyz, scale_factor=None, bbox_enlarge=0.0):
            # restatement of utils/bbox_utils.py:189-207 for duck-typed box objects that
            # offer check_xyz_in_bounds(xyz, scale_factor, bbox_enlarge) -> bool (n,)
            need_reshape = False
            if len(xyz.shape) == 3:
                n1, n2, _ = xyz.shape
                xyz = xyz.reshape(-1, 3)
                need_reshape = True
            in_bounds = torch.zeros_like(xyz[:, 0]).bool()
            for _, box in boxes.items():
                in_bounds = torch.logical_or(box.check_xyz_in_bounds(xyz, scale_factor, bbox_enlarge), in_bounds)
            if need_reshape:
                in_bounds = in_bounds.view(n1, n2)
            return in_bounds

        bb.check_in_any_boxes = check_in_any_boxes
        sys.modules["utils.bbox_utils"] = bb
        utils.bbox_utils = bb

        import models.rendering as rendering
        import models.nerf_model as nerf_model
        import models.embedding_helper as embedding_helper
        import models.code_library as code_library
        import render_tools.multi_rendering as multi_rendering
    finally:
        os.chdir(cwd)

    ns = types.SimpleNamespace(
        render_rays=rendering.render_rays,
        sample_pdf=rendering.sample_pdf,
        inference_model=rendering.inference_model,
        ObjectNeRF=nerf_model.ObjectNeRF,
        Embedding=embedding_helper.Embedding,
        EmbeddingVoxel=embedding_helper.EmbeddingVoxel,
        CodeLibrary=code_library.CodeLibrary,
        render_rays_multi=multi_rendering.render_rays_multi,
        volume_rendering_multi=multi_rendering.volume_rendering_multi,
        inference_from_model=multi_rendering.inference_from_model,
        modules=dict(rendering=rendering, nerf_model=nerf_model, embedding_helper=embedding_helper,
                     code_library=code_library, multi_rendering=multi_rendering),
    )
    _loaded = ns
    return ns
