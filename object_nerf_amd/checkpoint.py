"""Checkpoint interop: the reference's Lightning checkpoints hold one flat
`state_dict` whose keys are prefixed by the LightningModule attribute names
(`nerf_coarse.*`, `nerf_fine.*`, `code_library.embedding_instance.weight`,
`embedding_xyz.embedding_space_ftr.weight`, `embedding_xyz.{voxel_size,bounds,voxel_offset,
voxel_shape,voxel_count,voxel_occupancy,voxel_idx_map}`; train.py:45-65).

`build_from_state_dict` rebuilds the drop-in operator set straight from such a dict -- without
Lightning and without the scene's point cloud (the voxel grid state is taken from the checkpoint's
buffers instead of being recomputed from the .ply) -- so released checkpoints can be rendered on the
HIP path; `export_state_dict` writes the same key layout back.
"""
from collections import OrderedDict

from torch import nn

from .code_library import CodeLibrary
from .config import AttrDict, default_model_config
from .embedding_helper import Embedding, EmbeddingVoxel
from .nerf_model import ObjectNeRF

_VOXEL_BUFFERS = ("voxel_size", "bounds", "voxel_offset", "voxel_shape", "voxel_count", "voxel_occupancy", "voxel_idx_map")


def _sub(sd, prefix):
    return OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix))


def embedding_voxel_from_state(sub, n_freq_voxel=6):
    """EmbeddingVoxel whose grid state comes from checkpoint buffers (no point cloud needed).  The channel count is the
    table's; the number of voxel frequencies is not in a checkpoint -- it comes from config.model.N_freq_voxel."""
    table = sub["embedding_space_ftr.weight"]
    ev = EmbeddingVoxel.__new__(EmbeddingVoxel)
    nn.Module.__init__(ev)
    ev.N_freqs = int(n_freq_voxel)
    ev.fused_layout = (table.shape[1] == 24 and ev.N_freqs == 6)       # what the fused kernels embed in registers
    ev.embedding_final = Embedding(table.shape[1], ev.N_freqs)
    ev.embedding_space_ftr = nn.Embedding(table.shape[0], table.shape[1])
    for b in _VOXEL_BUFFERS:
        ev.register_buffer(b, sub[b].clone())
    ev.instance_ftr_C = 8
    ev.channels = table.shape[1]
    ev.embedding_xyz_classical = Embedding(3, 10)
    ev.conf = None
    ev._idx32 = None
    ev._idx32_key = None
    ev.load_state_dict(sub, strict=True)
    return ev


def build_from_state_dict(state_dict, model_config=None, device=None):
    """-> AttrDict(models={'coarse','fine'?}, embeddings={'xyz','dir'}, code_library, cfg)"""
    sd = state_dict.get("state_dict", state_dict)
    use_voxel = any(k.startswith("embedding_xyz.embedding_space_ftr") for k in sd)
    cfg = model_config if model_config is not None else default_model_config(use_voxel_embedding=use_voxel)
    if bool(cfg.use_voxel_embedding) != use_voxel:
        raise RuntimeError("checkpoint and model_config disagree about use_voxel_embedding")
    models = {}
    for typ in ("coarse", "fine"):
        sub = _sub(sd, "nerf_%s." % typ)
        if sub:
            m = ObjectNeRF(cfg)
            m.load_state_dict(sub, strict=True)
            models[typ] = m
    if "coarse" not in models:
        raise RuntimeError("checkpoint has no nerf_coarse.* parameters")
    emb_xyz = embedding_voxel_from_state(_sub(sd, "embedding_xyz."), cfg.get("N_freq_voxel", 6)) if use_voxel else Embedding(3, cfg["N_freq_xyz"])
    codes = CodeLibrary(cfg)
    sub = _sub(sd, "code_library.")
    if sub:
        codes = CodeLibrary(AttrDict(N_max_objs=sub["embedding_instance.weight"].shape[0],
                                     N_obj_code_length=sub["embedding_instance.weight"].shape[1]))
        codes.load_state_dict(sub, strict=True)
    mods = list(models.values()) + [codes] + ([emb_xyz] if use_voxel else [])
    for m in mods:
        if device is not None:
            m.to(device)
        m.eval()
    return AttrDict(models=models, embeddings={"xyz": emb_xyz, "dir": Embedding(3, cfg["N_freq_dir"])},
                    code_library=codes, cfg=cfg)


def export_state_dict(scene):
    """Flat Lightning-style state_dict of a scene built by build_from_state_dict / synth.build_scene, keys in the order
    the reference's LightningModule registers its children (train.py:45-65: embedding_xyz, nerf_coarse, nerf_fine,
    code_library)."""
    out = OrderedDict()
    if isinstance(scene.embeddings["xyz"], EmbeddingVoxel):
        for k, v in scene.embeddings["xyz"].state_dict().items():
            out["embedding_xyz." + k] = v.detach().cpu()
    for typ in ("coarse", "fine"):
        if typ in scene.models:
            for k, v in scene.models[typ].state_dict().items():
                out["nerf_%s.%s" % (typ, k)] = v.detach().cpu()
    for k, v in scene.code_library.state_dict().items():
        out["code_library." + k] = v.detach().cpu()
    return out
