"""Reference import path `render_tools.multi_rendering` -> object_nerf_amd.multi_rendering.
(`render_tools` has no __init__.py in the reference either: it is a namespace package, so
`render_tools.editable_renderer` keeps resolving to the reference's file.)"""
from object_nerf_amd.multi_rendering import render_rays_multi  # noqa: F401
from object_nerf_amd.rendering import sample_pdf  # noqa: F401
from object_nerf_amd.bbox import check_in_any_boxes  # noqa: F401
