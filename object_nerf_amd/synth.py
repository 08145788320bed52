"""Deterministic synthetic workloads (SURVEY.md §8d "Synthetic inputs").

There are no datasets or checkpoints in the build container or on the GPU box, so every test,
the smoke test and bench.py render these: a pinhole camera inside a synthetic room, a sparse
voxel grid built from a seeded point cloud, and seeded MLP weights with enough contrast
("W1": He-normal weights, sigma heads with gain 10 and bias -5) that compositing, importance
sampling and the occlusion logic are all exercised.

Everything here is CPU-side construction of INPUTS (plain tensors); no reference or oracle
code is involved.
"""
import math

import numpy as np
import torch

from .config import AttrDict, default_model_config

SCANNET_LIKE = dict(near=0.15, far=3.0, scale_factor=2.0, voxel_size=0.1, scene_center=[2.0, 2.0, 0.0],
                    frustum_bound_th=0.025)
TOYDESK_LIKE = dict(near=0.05, far=1.5, scale_factor=2.0, voxel_size=0.1, scene_center=[2.0, 2.0, 0.0],
                    frustum_bound_th=-1.0 / 16)


def room_point_cloud(n_points=200_000, seed=0):
    """200k points uniform in a 6 x 6 x 2.5 m room, first half snapped to the floor."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([0, 0, 0], [6, 6, 2.5], size=(n_points, 3))
    pts[: n_points // 2, 2] = 0.0
    return pts


def dataset_extra(preset=SCANNET_LIKE, n_points=200_000, seed=0):
    return AttrDict(pcd_xyz=room_point_cloud(n_points, seed), scene_center=preset["scene_center"],
                    scale_factor=preset["scale_factor"], voxel_size=preset["voxel_size"], neighbor_marks=3)


def camera_rays(W=640, H=480, fov_x_deg=60.0, near=0.15, far=3.0, origin=(0.5, 0.5, 0.6), yaw_deg=35.0,
                pitch_deg=-15.0):
    """Pinhole rays in the reference's layout [o(3), d(3), near, far]:
    dirs = [(i - W/2)/f, -(j - H/2)/f, -1] (datasets/ray_utils.py:17-23, no +0.5), rotated by a
    camera-to-world rotation, normalised (ray_utils.py:43-49); row-major pixel order."""
    f = 0.5 * W / math.tan(0.5 * math.radians(fov_x_deg))
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    dirs = torch.stack([(i - W / 2) / f, -(j - H / 2) / f, -torch.ones_like(i)], -1).reshape(-1, 3)
    # camera looks along -z; tilt it to look across the room
    cy, sy = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    cp, sp = math.cos(math.radians(90 + pitch_deg)), math.sin(math.radians(90 + pitch_deg))
    Rp = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=torch.float32)
    Ry = torch.tensor([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=torch.float32)
    R = Ry @ Rp
    d = dirs @ R.T
    d = d / d.norm(dim=-1, keepdim=True)
    o = torch.tensor(origin, dtype=torch.float32).expand_as(d)
    n = d.shape[0]
    return torch.cat([o, d, torch.full((n, 1), near), torch.full((n, 1), far)], -1).contiguous()


def fill_w1(model, seed):
    """"W1" weights: He-normal N(0, sqrt(2/fan_in)) on every weight, zero biases, sigma heads
    N(0, 10/sqrt(fan_in)) with bias -5.  Filled in sorted-name order from one seeded generator,
    so the reference module and the drop-in module (same parameter names) get identical values."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    new = {}
    for name in sorted(sd.keys()):
        t = sd[name]
        if not name.endswith((".weight", ".bias")) or t.dtype != torch.float32:
            new[name] = t
            continue
        is_sigma = name.startswith("sigma.") or name.startswith("instance_sigma.")
        if name.endswith(".weight"):
            fan_in = t.shape[1]
            std = 10.0 / math.sqrt(fan_in) if is_sigma else math.sqrt(2.0 / fan_in)
            new[name] = torch.randn(t.shape, generator=g) * std
        else:
            new[name] = torch.full(t.shape, -5.0) if is_sigma else torch.zeros(t.shape)
    model.load_state_dict(new)
    return model


def fill_table(embedding_voxel, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = embedding_voxel.embedding_space_ftr.weight
    with torch.no_grad():
        w.copy_(torch.randn(w.shape, generator=g))
    return embedding_voxel


def fill_codes(code_library, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = code_library.embedding_instance.weight
    with torch.no_grad():
        w.copy_(torch.randn(w.shape, generator=g))
    return code_library


def per_ray_ids(n_rays, ids=(5, 4, 2, 1, 3), seed=0):
    """config 3: per-ray object ids drawn from the 5 ToyDesk ids (config/toy_desk_2.yml:46)"""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.choice(np.asarray(ids), size=n_rays)).long()


def build_scene(types, use_voxel=True, preset=SCANNET_LIKE, max_voxels=800_000, n_points=200_000, device="cpu",
                n_importance=64):
    """Builds {models, embeddings, code_library} with the given module types (a namespace offering
    ObjectNeRF, Embedding, EmbeddingVoxel, CodeLibrary -- the drop-in package or the reference)."""
    cfg = default_model_config(use_voxel_embedding=use_voxel, N_max_voxels=max_voxels, N_importance=n_importance)
    if use_voxel:
        emb_xyz = types.EmbeddingVoxel(24, 6, max_voxels, dataset_extra(preset, n_points))
        fill_table(emb_xyz, 0)
    else:
        emb_xyz = types.Embedding(3, 10)
    emb_dir = types.Embedding(3, 4)
    coarse = fill_w1(types.ObjectNeRF(cfg), 1)
    fine = fill_w1(types.ObjectNeRF(cfg), 2)
    codes = fill_codes(types.CodeLibrary(cfg), 0)
    mods = [coarse, fine, codes] + ([emb_xyz] if use_voxel else [])
    for m in mods:
        m.to(device)
        m.eval()
    return AttrDict(models={"coarse": coarse, "fine": fine}, embeddings={"xyz": emb_xyz, "dir": emb_dir},
                    code_library=codes, cfg=cfg, preset=preset)


def oriented_box(center, size, yaw_deg, scene_center, scale_factor):
    """A box in the reference's BBoxRayHelper terms (utils/bbox_utils.py:9-117): pose_avg =
    [I | scene_center] (3x4), axis_align_mat = world->box rigid transform, bbox_bounds in box
    coordinates.  Returned as the dict oracle.points_in_boxes / object_nerf_amd.bbox consume."""
    c, s = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    R = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])     # world -> box rotation
    t = -R @ np.asarray(center, dtype=np.float64)
    half = 0.5 * np.asarray(size, dtype=np.float64)
    return dict(scale_factor=float(scale_factor), R_avg=np.eye(3), t_avg=np.asarray(scene_center, dtype=np.float64),
                R_box=R, t_box=t, bmin=-half, bmax=half)
