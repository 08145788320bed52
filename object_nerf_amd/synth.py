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
                    frustum_bound_th=-1.0 / 16)     # round-1 bench preset: the room with ToyDesk near/far only
# config/toy_desk_2.yml:8-11,15,61-62,64 as they reach the renderer: near/far 0.8/24 and voxel_size 0.3 in reconstruction
# units divided by scale_factor 16 (generic_dataset.py:444-447, embedding_helper.py:102-103); frustum_bound -1 -> disabled
TOYDESK2 = dict(near=0.8 / 16.0, far=24.0 / 16.0, scale_factor=16.0, voxel_size=0.3, scene_center=[0.2, 1.4, 7.1],
                frustum_bound_th=-1.0 / 16.0, cloud="desk", cam_origin=(0.34, -0.28, 0.34), cam_target=(0.0, 0.0, 0.03))
SCANNET_LIKE.update(cloud="room", cam_origin=None, cam_target=None)
TOYDESK_LIKE.update(cloud="room", cam_origin=None, cam_target=None)


def room_point_cloud(n_points=200_000, seed=0):
    """200k points uniform in a 6 x 6 x 2.5 m room, first half snapped to the floor."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([0, 0, 0], [6, 6, 2.5], size=(n_points, 3))
    pts[: n_points // 2, 2] = 0.0
    return pts


def desk_point_cloud(n_points=200_000, seed=0, center=(0.2, 1.4, 7.1)):
    """ToyDesk-2-like cloud in reconstruction units ("desk real width 1.06m, recon width 8.3", toy_desk_2.yml:5):
    half of the points on the 8.3 x 8.3 desk top, half in the 3-unit-high volume of the objects standing on it, placed at
    the config's scene_center."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-4.15, -4.15, 0.0], [4.15, 4.15, 3.0], size=(n_points, 3))
    pts[: n_points // 2, 2] = 0.0
    return pts + np.asarray(center, dtype=np.float64)


def dataset_extra(preset=SCANNET_LIKE, n_points=200_000, seed=0):
    if preset.get("cloud", "room") == "desk":
        cloud = desk_point_cloud(n_points, seed, preset["scene_center"])
    else:
        cloud = room_point_cloud(n_points, seed)
    return AttrDict(pcd_xyz=cloud, scene_center=preset["scene_center"],
                    scale_factor=preset["scale_factor"], voxel_size=preset["voxel_size"], neighbor_marks=3)


def rotate_rows(v, R, normalise=False):
    """rows of v (n, 3) times R^T, REPRODUCIBLY on any host: float64 element-wise products and sums in a fixed order (every one
    of them a correctly rounded IEEE operation), optional normalisation by sqrt(x^2 + y^2 + z^2) the same way, rounded to fp32
    once at the end.  A `v @ R.T` in fp32 goes through the host's BLAS and `v.norm(dim=-1)` through a vectorised reduction:
    both round differently on different CPUs (measured, round 6: the same call gave directions one ulp apart on the Intel build
    container and on the AMD host of the GPU box, i.e. the goldens and the GPU tests rendered different rays -- the whole
    "coarse keys 1.6e-4 from the reference" of rounds 3-5 was that, tools/coarse_worst_ray.py)."""
    v = v.double()
    R = [[float(R[r][c]) for c in range(3)] for r in range(3)]
    cols = []
    for r in range(3):
        a = v[:, 0] * R[r][0]
        b = v[:, 1] * R[r][1]
        c = v[:, 2] * R[r][2]
        cols.append((a + b) + c)
    if normalise:
        n = torch.sqrt((cols[0] * cols[0] + cols[1] * cols[1]) + cols[2] * cols[2])
        cols = [c / n for c in cols]
    return torch.stack(cols, -1).float()


def _matmul3(A, B):
    """3x3 product in Python floats (float64, fixed order)"""
    return [[(A[r][0] * B[0][c] + A[r][1] * B[1][c]) + A[r][2] * B[2][c] for c in range(3)] for r in range(3)]


def _pinhole_rays(W, H, f, R, origin, near, far):
    """[o(3), d(3), near, far] rows of a pinhole camera: dirs = [(i - W/2)/f, -(j - H/2)/f, -1] (datasets/ray_utils.py:17-23, no
    +0.5) in fp32 like the reference's grid, rotated by the camera-to-world rotation R and normalised (ray_utils.py:43-49)
    through `rotate_rows` (host-independent); row-major pixel order."""
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    dirs = torch.stack([(i - W / 2) / f, -(j - H / 2) / f, -torch.ones_like(i)], -1).reshape(-1, 3)
    d = rotate_rows(dirs, R, normalise=True)
    n = d.shape[0]
    o = torch.tensor([float(x) for x in origin], dtype=torch.float64).float().expand_as(d)
    return torch.cat([o, d, torch.full((n, 1), float(near)), torch.full((n, 1), float(far))], -1).contiguous()


def camera_rays(W=640, H=480, fov_x_deg=60.0, near=0.15, far=3.0, origin=(0.5, 0.5, 0.6), yaw_deg=35.0,
                pitch_deg=-15.0):
    """Pinhole rays in the reference's layout [o(3), d(3), near, far] (see _pinhole_rays): the camera looks along -z, tilted
    to look across the room."""
    f = 0.5 * W / math.tan(0.5 * math.radians(fov_x_deg))
    cy, sy = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    cp, sp = math.cos(math.radians(90 + pitch_deg)), math.sin(math.radians(90 + pitch_deg))
    Rp = [[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]]
    Ry = [[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]]
    return _pinhole_rays(W, H, f, _matmul3(Ry, Rp), origin, near, far)


def look_at_rays(W, H, origin, target, near, far, fov_x_deg=60.0, up=(0.0, 0.0, 1.0)):
    """The same pinhole model as `camera_rays` with the pose given as eye point / look-at point (normalised units)."""
    f = 0.5 * W / math.tan(0.5 * math.radians(fov_x_deg))

    def unit(v):
        n = math.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
        return [v[0] / n, v[1] / n, v[2] / n]

    def cross(a, b):
        return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
    o = [float(x) for x in origin]
    fwd = unit([float(t) - x for t, x in zip(target, o)])
    right = unit(cross(fwd, [float(x) for x in up]))
    upv = cross(right, fwd)
    R = [[right[r], upv[r], -fwd[r]] for r in range(3)]     # camera x, y, z axes as columns (camera looks along -z)
    return _pinhole_rays(W, H, f, R, o, near, far)


def preset_rays(preset, W=640, H=480, yaw_offset_deg=0.0):
    """The bench / golden camera of a preset: 640x480, 60 degrees, inside the room (ScanNet-like) or looking at the desk
    from 0.54 normalised units = 8.6 reconstruction units (ToyDesk-2: 88 % of the pixels see the desk volume).  yaw_offset_deg rotates the view (one camera per rank in weak scaling)."""
    if preset.get("cam_origin") is not None:
        o, t = preset["cam_origin"], preset["cam_target"]
        a = math.radians(yaw_offset_deg)
        ox, oy = o[0] - t[0], o[1] - t[1]
        o2 = (t[0] + ox * math.cos(a) - oy * math.sin(a), t[1] + ox * math.sin(a) + oy * math.cos(a), o[2])
        return look_at_rays(W, H, o2, t, preset["near"], preset["far"])
    return camera_rays(W, H, near=preset["near"], far=preset["far"], yaw_deg=35.0 + yaw_offset_deg)


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
                n_importance=64, model_overrides=None, logscale=True):
    """Builds {models, embeddings, code_library} with the given module types (a namespace offering
    ObjectNeRF, Embedding, EmbeddingVoxel, CodeLibrary -- the drop-in package or the reference), the way train.py:40-65
    derives them from config.model.  model_overrides: non-default config.model entries (D, W, skips, inst_*, N_freq_*,
    N_scn_voxel_size, N_obj_code_length, ...); logscale=False: linearly spaced frequency bands (embedding_helper.py:54-55)."""
    cfg = default_model_config(use_voxel_embedding=use_voxel, N_max_voxels=max_voxels, N_importance=n_importance)
    cfg.update(model_overrides or {})
    lk = {} if logscale else {"logscale": False}
    if use_voxel:
        emb_xyz = types.EmbeddingVoxel(cfg.N_scn_voxel_size + cfg.N_obj_voxel_size, cfg.N_freq_voxel, max_voxels,
                                       dataset_extra(preset, n_points))
        fill_table(emb_xyz, 0)
    else:
        emb_xyz = types.Embedding(3, cfg.N_freq_xyz, **lk)
    emb_dir = types.Embedding(3, cfg.N_freq_dir, **lk)
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


def edit_demo_geometry(preset, W):
    """Camera, object box and the two object poses of the duplicating + moving demo (test/demo_editable_render.py:33-42
    shape: each copy gets a small x/y offset and a z rotation; render_tools/editable_renderer.py:231-263 turns a pose T
    into the camera-to-object matrix inv(T) @ Twc).  A sofa-sized box (1.0 x 0.8 x 1.0 m) 2.6 m in front of the camera:
    the moved object covers 21 % of the 640x480 frame, its duplicate 19 %, the background set all of it."""
    focal = (W / 2) / np.tan(math.radians(60.0) / 2)                       # editable_renderer.py:214
    cy, sy = math.cos(math.radians(35.0)), math.sin(math.radians(35.0))
    cp, sp = math.cos(math.radians(75.0)), math.sin(math.radians(75.0))
    R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]) @ np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]])
    eye = np.array([0.5, 0.5, 0.6])
    Twc = np.concatenate([R, eye[:, None]], 1)
    cn = eye + 1.3 * (R @ np.array([0.0, 0.0, -1.0]))                      # object centre, normalised units, on the view axis
    cn[2] = 0.25
    center_w = cn * preset["scale_factor"] + np.asarray(preset["scene_center"], dtype=np.float64)
    box = oriented_box(center_w, [1.0, 0.8, 1.0], 20.0, preset["scene_center"], preset["scale_factor"])

    def moved(dx, dy, dz, yaw):
        c, s = math.cos(math.radians(yaw)), math.sin(math.radians(yaw))
        Rz, A, B, D, M = np.eye(4), np.eye(4), np.eye(4), np.eye(4), np.eye(4)
        Rz[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
        A[:3, 3], B[:3, 3], D[:3, 3] = cn, -cn, [dx, dy, dz]              # rotate about the object's centre, then shift
        M[:3] = Twc
        return (np.linalg.inv(D @ A @ Rz @ B) @ M)[:3]
    # the small lifts keep the two copies' top / bottom faces on different planes: with equal planes both ray sets leave
    # their boxes at the SAME depth on many pixels, and the order of exactly tied depths in the joint sort is unspecified
    # in the reference (unstable torch.sort, multi_rendering.py:112) -- parity could not be pinned on such pixels
    return focal, [Twc, moved(0.05, 0.2, 0.013, 10.0), moved(-0.25, 0.15, 0.031, -10.0)], box
