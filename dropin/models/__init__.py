"""`models` package shim: put this directory's parent (`dropin/`) BEFORE the reference checkout on
PYTHONPATH and the reference's own `train.py` / `render_tools/editable_renderer.py` /
`tools/extract_mesh.py` pick up the MI355X hot path through their unchanged import lines
(train.py:15-18, editable_renderer.py:22):

    from models.nerf_model import ObjectNeRF            -> object_nerf_amd.nerf_model
    from models.embedding_helper import EmbeddingVoxel, Embedding
    from models.rendering import render_rays
    from models.code_library import CodeLibrary
    from render_tools.multi_rendering import render_rays_multi

Everything else of the reference's `models` package (models/losses.py) still resolves to the
reference: the package path is extended over every `models/` directory on sys.path.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
