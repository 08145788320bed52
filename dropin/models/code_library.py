"""Reference import path `models.code_library` -> object_nerf_amd.code_library (train.py:18)."""
from object_nerf_amd.code_library import CodeLibrary  # noqa: F401
