"""Reference import path `models.embedding_helper` -> object_nerf_amd.embedding_helper (train.py:16)."""
from object_nerf_amd.embedding_helper import Embedding, EmbeddingVoxel  # noqa: F401
