"""Reference import path `models.rendering` -> object_nerf_amd.rendering (train.py:17)."""
from object_nerf_amd.rendering import render_rays, sample_pdf  # noqa: F401

__all__ = ["render_rays", "sample_pdf"]
