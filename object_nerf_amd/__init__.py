"""object_nerf_amd -- MI355X (gfx950) native drop-in for the volume-rendering hot path of
zju3dv/object_nerf: render_rays / render_rays_multi and the operator types they take.

Python here is plumbing (argument checks, tensor allocation, ctypes calls into
libobjnerf_hip.so).  All arithmetic of the path runs in the HIP kernels under csrc/.
"""
from .config import AttrDict, default_model_config  # noqa: F401
from .nerf_model import ObjectNeRF  # noqa: F401
from .embedding_helper import Embedding, EmbeddingVoxel  # noqa: F401
from .code_library import CodeLibrary  # noqa: F401
from .rendering import render_rays, sample_pdf  # noqa: F401

__all__ = ["ObjectNeRF", "Embedding", "EmbeddingVoxel", "CodeLibrary", "render_rays", "sample_pdf",
           "AttrDict", "default_model_config"]
