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
  * registers a `datasets` package object pointing at the reference's datasets/ directory (so that the
    REAL datasets/geo_utils.py, datasets/ray_utils.py and utils/bbox_utils.py import without running
    datasets/__init__.py, which drags in cv2/torchvision), an identity `numba.jit`, and a restatement of
    `kornia.create_meshgrid` (kornia==0.4.1 is a pinned, un-vendored dependency); `make_box()` builds a
    BBoxRayHelper without its file-reading constructor;
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


def install_stubs():
    """Stub modules for the reference's un-installed, off-path dependencies (idempotent)."""
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
    if "datasets" not in sys.modules:
        # a package object whose __path__ is the reference's datasets/ directory: submodules
        # (geo_utils.py, ray_utils.py) import from the REAL files, datasets/__init__.py is never run
        ds = types.ModuleType("datasets")
        ds.__path__ = [os.path.join(REF_ROOT, "datasets")]
        sys.modules["datasets"] = ds
    if "numba" not in sys.modules:      # datasets/geo_utils.py:2,111,126: @nb.jit(nopython=True) -> plain Python
        nb = types.ModuleType("numba")
        nb.jit = lambda *a, **k: (lambda f: f)
        sys.modules["numba"] = nb
    if "kornia" not in sys.modules:
        # datasets/ray_utils.py:2,17 uses kornia.create_meshgrid (requirements.txt pins kornia==0.4.1, not
        # vendored, not installed here).  Restatement of its published behaviour for normalized_coordinates=False:
        # grid[0, y, x] = (x, y) as float32.
        kn = types.ModuleType("kornia")

        def create_meshgrid(height, width, normalized_coordinates=True, device=None):
            assert not normalized_coordinates
            xs = torch.linspace(0, width - 1, width, dtype=torch.float)
            ys = torch.linspace(0, height - 1, height, dtype=torch.float)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            return torch.stack([gx, gy], -1)[None]

        kn.create_meshgrid = create_meshgrid
        sys.modules["kornia"] = kn


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

    install_stubs()

    cwd = os.getcwd()
    sys.path.insert(0, REF_ROOT)
    try:
        import utils  # noqa: F401  (real package; needs the stubs above)

        # utils/bbox_utils.py:6 imports datasets.geo_utils (-> datasets/__init__.py -> cv2, numba,
        # kornia, torchvision).  Only `bbox_intersection_batch` is taken from it and the hot path
        # (check_in_any_boxes / check_xyz_in_bounds, bbox_utils.py:158-207) never calls it, so a stub
        # `datasets` package lets the REAL utils/bbox_utils.py import unmodified.
        import datasets.geo_utils as geo_utils
        import datasets.ray_utils as ray_utils
        import utils.bbox_utils as bbox_utils

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
        get_ray_directions=ray_utils.get_ray_directions,
        get_rays=ray_utils.get_rays,
        bbox_intersection_batch=geo_utils.bbox_intersection_batch,
        BBoxRayHelper=bbox_utils.BBoxRayHelper,
        check_in_any_boxes=bbox_utils.check_in_any_boxes,
        modules=dict(rendering=rendering, bbox_utils=bbox_utils, nerf_model=nerf_model, embedding_helper=embedding_helper,
                     code_library=code_library, multi_rendering=multi_rendering),
    )
    _loaded = ns
    return ns


def make_box(box):
    """A reference BBoxRayHelper (utils/bbox_utils.py:9-117) for a box dict as produced by
    object_nerf_amd.synth.oriented_box, bypassing the constructor that reads dataset files."""
    ns = load_reference()
    b = object.__new__(ns.BBoxRayHelper)
    b.scale_factor = box["scale_factor"]
    b.pose_avg = np.eye(4)
    b.pose_avg[:3, :3] = box["R_avg"]
    b.pose_avg[:3, 3] = box["t_avg"]
    b.axis_align_mat = np.eye(4)
    b.axis_align_mat[:3, :3] = box["R_box"]
    b.axis_align_mat[:3, 3] = box["t_box"]
    b.bbox_bounds = np.array([np.asarray(box["bmin"], dtype=np.float64), np.asarray(box["bmax"], dtype=np.float64)])
    return b


class inject_randoms:
    """Context manager: while active, torch.rand_like / torch.randn_like / torch.rand return the
    queued tensors (in call order) instead of drawing.  Lets the reference's training-mode paths
    (rendering.py:276, 40, 156, 187) run on the same random tensors as the oracle / HIP path."""

    def __init__(self, rand_like=(), randn_like=(), rand=()):
        self.q = {"rand_like": list(rand_like), "randn_like": list(randn_like), "rand": list(rand)}

    def __enter__(self):
        self.saved = {k: getattr(torch, k) for k in self.q}
        for k in self.q:
            def fake(*a, _k=k, **kw):
                return self.q[_k].pop(0).clone()
            setattr(torch, k, fake)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(torch, k, v)
        left = {k: len(v) for k, v in self.q.items() if v}
        if left and exc[0] is None:
            raise RuntimeError("inject_randoms: unused tensors %r" % left)
        return False
