"""Reference import path `models.nerf_model` -> object_nerf_amd.nerf_model (train.py:15)."""
from object_nerf_amd.nerf_model import ObjectNeRF  # noqa: F401
