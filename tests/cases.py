"""Deterministic parity cases shared by oracle/make_golden.py (which runs the REFERENCE on them and
commits the outputs under tests/golden/) and by the tests (which run the oracle and the HIP path
on the same inputs).  Inputs are regenerated from seeds; only reference OUTPUTS are stored."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from object_nerf_amd import synth  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
N_RAYS = 48
MAX_VOXELS = 120_000

# scene variants: name -> (use_voxel, n_points of the synthetic cloud[, preset, rows of the voxel table])
SCENES = {"voxel": (True, 200_000), "plain": (False, 0), "sparse": (True, 1500),
          # the scenes bench.py measures on (its presets, its 800000-row table)
          "toydesk2_800k": (True, 200_000, synth.TOYDESK2, 800_000),
          "scannet_800k": (True, 200_000, synth.SCANNET_LIKE, 800_000)}

# architectures OTHER than the shipped default (models/nerf_model.py:18-95 builds any of them; the product renders them on
# its layer-wise path, object_nerf_amd/generic.py): name -> (use_voxel, n_points, config.model overrides, logscale)
ARCH_SCENES = {
    # narrower / shallower, skips elsewhere, fewer frequencies with LINEAR bands, a 32-d code
    "arch_plain_small": (False, 0, dict(D=6, W=128, skips=[3], inst_D=3, inst_W=64, inst_skips=[1], N_freq_xyz=8, N_freq_dir=3,
                                        N_obj_code_length=32), False),
    # voxel mode with 12 + 8 feature channels and 4 voxel frequencies, no skip in the object branch, two skips in the scene one
    "arch_voxel_odd": (True, 200_000, dict(D=5, W=192, skips=[2, 4], inst_D=2, inst_W=96, inst_skips=[], N_scn_voxel_size=12,
                                           N_freq_voxel=4, N_freq_dir=5), True),
}
ARCH_RENDER = dict(N_samples=24, N_importance=40, n_rays=20)

# render_rays cases: scene, kwargs for render_rays, extras
RENDER_CASES = {
    # BASELINE configs[1]-like: scene + object, 64+64, eval
    "voxel_eval": dict(scene="voxel", kw=dict(N_samples=64, N_importance=64, is_eval=True)),
    "plain_eval": dict(scene="plain", kw=dict(N_samples=64, N_importance=64, is_eval=True)),
    # BASELINE configs[0]: scene branch only, 64 coarse
    "voxel_scene_only": dict(scene="voxel", kw=dict(N_samples=64, N_importance=0, forward_instance=False, is_eval=True)),
    # training-time flags: occlusion mask + pass-through + rays_in_bbox weights overwrite + white background
    "voxel_train_flags": dict(scene="voxel", ptm=True,
                              kw=dict(N_samples=64, N_importance=64, is_eval=False, frustum_bound_th=0.025,
                                      rays_in_bbox=True, white_back=True)),
    "voxel_disp_zero": dict(scene="voxel", kw=dict(N_samples=64, N_importance=64, use_disp=True,
                                                   use_zero_as_last_delta=True, is_eval=True)),
    # BASELINE configs[2]: 5 object codes, 64+128, frustum bound, val-time use_bbox
    "voxel_imp128": dict(scene="voxel", kw=dict(N_samples=64, N_importance=128, is_eval=True, frustum_bound_th=0.025,
                                                rays_in_bbox=True)),
    # empty voxels and rays leaving the grid
    "sparse_eval": dict(scene="sparse", far=6.0, kw=dict(N_samples=64, N_importance=64, is_eval=True)),
    # RNG paths with injected draws: perturb > 0 (rendering.py:268-277, 40) and noise_std > 0 (156, 187)
    "voxel_random": dict(scene="voxel", ptm=True, randoms=True,
                         kw=dict(N_samples=64, N_importance=64, perturb=1.0, noise_std=1.0, is_eval=False,
                                 frustum_bound_th=0.025)),
    "plain_odd_sizes": dict(scene="plain", kw=dict(N_samples=40, N_importance=24, is_eval=True)),
    # ---- the workloads bench.py measures, on 48 rays spread over ITS 640x480 camera (bench.py --config 1 / 2) ----
    # BASELINE configs[1]: true ToyDesk-2 geometry (config/toy_desk_2.yml:8-11,61-64: scale 16, voxel 0.3, near/far 0.8/24,
    # frustum bound disabled), one object code (val_instance_id = 1), 64 + 64, 800000-row table
    "bench_toydesk2": dict(scene="toydesk2_800k", frame=(640, 480), ids=1,
                           kw=dict(N_samples=64, N_importance=64, is_eval=True, white_back=False, forward_instance=True,
                                   frustum_bound_th=synth.TOYDESK2["frustum_bound_th"])),
    # BASELINE configs[2]/[3]: ScanNet-0113-multi-like, 5 object codes per ray, 64 + 128, frustum bound, rays_in_bbox
    "bench_scannet_multi": dict(scene="scannet_800k", frame=(640, 480), ids="five",
                                kw=dict(N_samples=64, N_importance=128, is_eval=True, white_back=False, forward_instance=True,
                                        frustum_bound_th=0.025, rays_in_bbox=True)),
}


def scene_for(types, name, device="cpu"):
    if name in ARCH_SCENES:
        use_voxel, n_points, over, logscale = ARCH_SCENES[name]
        return synth.build_scene(types, use_voxel, preset=synth.SCANNET_LIKE, max_voxels=MAX_VOXELS, n_points=max(n_points, 1),
                                 device=device, model_overrides=over, logscale=logscale)
    use_voxel, n_points = SCENES[name][:2]
    preset, max_voxels = (SCENES[name][2], SCENES[name][3]) if len(SCENES[name]) > 2 else (synth.SCANNET_LIKE, MAX_VOXELS)
    return synth.build_scene(types, use_voxel, preset=preset, max_voxels=max_voxels, n_points=max(n_points, 1), device=device)


def oracle_arch(name):
    """the oracle's `arch` dict (oracle.objnerf_oracle.DEFAULT_ARCH keys) of an ARCH_SCENES entry"""
    _, _, over, logscale = ARCH_SCENES[name]
    m = {"D": "D", "skips": "skips", "inst_D": "inst_D", "inst_skips": "inst_skips", "N_freq_xyz": "n_freq_xyz",
         "N_freq_dir": "n_freq_dir", "N_freq_voxel": "n_freq_voxel"}
    a = {m[k]: (tuple(v) if isinstance(v, list) else v) for k, v in over.items() if k in m}
    a["logscale"] = logscale
    return a


def arch_inputs(name):
    """rays, per-ray ids, pass-through mask and the multi ray sets / box of the non-default-architecture goldens"""
    n = ARCH_RENDER["n_rays"]
    rays_all = synth.camera_rays(64, 48, far=3.0)
    rays = rays_all[torch.arange(0, rays_all.shape[0], 149)[:n]].contiguous()
    ids = synth.per_ray_ids(n, seed=5)
    ptm = (torch.arange(n) % 4 == 0).view(n, 1)
    sets, boxes = multi_inputs()
    return rays, ids, ptm, [s_[:n].contiguous() for s_ in sets], boxes


def render_inputs(case):
    """rays (N,8), per-ray ids (N), pass_through_mask (N,1) or None, randoms dict or None"""
    c = RENDER_CASES[case]
    if "frame" in c:       # bench workload: rays spread over the bench camera's full frame, bench.py's code assignment
        w, h = c["frame"]
        rays_all = synth.preset_rays(SCENES[c["scene"]][2], w, h)
        idx = torch.arange(0, w * h, 6421)[:N_RAYS]
        rays = rays_all[idx].contiguous()
        n = rays.shape[0]
        ids = synth.per_ray_ids(w * h)[idx] if c["ids"] == "five" else torch.full((n,), int(c["ids"]), dtype=torch.long)
        return rays, ids, None, None
    far = c.get("far", 3.0)
    rays_all = synth.camera_rays(64, 48, far=far)
    idx = torch.arange(0, rays_all.shape[0], 61)[:N_RAYS]
    rays = rays_all[idx].contiguous()
    n = rays.shape[0]
    ids = synth.per_ray_ids(n, seed=3)
    ptm = (torch.arange(n) % 3 == 0).view(n, 1) if c.get("ptm") else None
    randoms = None
    if c.get("randoms"):
        g = torch.Generator().manual_seed(11)
        S, I = c["kw"]["N_samples"], c["kw"]["N_importance"]
        randoms = dict(perturb_rand=torch.rand(n, S, generator=g), u_rand=torch.rand(n, I, generator=g),
                       noise=[torch.randn(n, S, generator=g), torch.randn(n, S, generator=g),
                              torch.randn(n, S + I, generator=g), torch.randn(n, S + I, generator=g)])
    return rays, ids, ptm, randoms


# ---- stage-level inputs -----------------------------------------------------------------------------
def pe_inputs():
    g = torch.Generator().manual_seed(21)
    return {"xyz10": (torch.randn(200, 3, generator=g) * 1.5, 10), "dir4": (torch.randn(120, 3, generator=g), 4),
            "vox6": (torch.randn(90, 16, generator=g) * 2.0, 6)}


def voxel_points(n=400):
    """points inside, near the faces of, and outside the normalised room"""
    g = torch.Generator().manual_seed(22)
    lo, hi = torch.tensor([-1.3, -1.3, -0.3]), torch.tensor([2.3, 2.3, 1.6])
    p = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    p[:8] = torch.tensor([[-1.0, -1.0, 0.0], [2.0, 2.0, 1.25], [-1.0, 2.0, 0.0], [0.0, 0.0, 0.0],
                          [1e3, 0.0, 0.0], [-1e3, 5.0, 0.5], [0.5, 0.5, -5.0], [0.05, 0.05, 0.05]])
    return p


def mlp_inputs(use_voxel, n=200):
    g = torch.Generator().manual_seed(23)
    return dict(emb_xyz=torch.randn(n, 271 if use_voxel else 63, generator=g), emb_dir=torch.randn(n, 27, generator=g),
                obj_voxel=torch.randn(n, 104, generator=g) if use_voxel else None, obj_code=torch.randn(n, 64, generator=g))


# ---- density-grid query (tools/extract_mesh.py:53-66) ----------------------------------
SIGMA_GRID = dict(N=32, x_range=(-1.3, 2.2), y_range=(-1.2, 2.1), z_range=(-0.3, 1.5), obj_id=4)


def sigma_grid_axes(n=None):
    """np.linspace axes as the script builds them (float64; the consumer rounds to fp32): a box around the synthetic room
    (normalised x, y in [-1, 2], z in [0, 1.25]) that also reaches outside the voxel grid on every side"""
    g = SIGMA_GRID
    n = n or (g["N"],) * 3
    return tuple(np.linspace(r[0], r[1], k) for r, k in zip((g["x_range"], g["y_range"], g["z_range"]), n))


def pdf_inputs():
    g = torch.Generator().manual_seed(24)
    n, nb = 37, 63
    bins = torch.sort(torch.rand(n, nb, generator=g) * 3.0, -1)[0]
    w = torch.rand(n, nb - 1, generator=g) ** 4
    w[::3, 10:30] = 0.0        # empty bins -> denom < eps branch (rendering.py:53-54)
    w[5] = 0.0                 # all-zero row
    u = torch.rand(n, 50, generator=g)
    return bins, w, u


# ---- multi-object case (render_rays_multi, BASELINE configs[4]: ids [0,4,4]) ----------------------------
MULTI = dict(obj_ids=[0, 4, 4], N_samples=64, N_importance=64, n_rays=40)


def multi_inputs():
    n = MULTI["n_rays"]
    rays_all = synth.camera_rays(64, 48, far=3.0)
    idx = torch.arange(0, rays_all.shape[0], 73)[:n]
    bg = rays_all[idx].contiguous()
    sets = [bg]
    for k, (shift, near, far) in enumerate([((0.05, 0.10, 0.0), 0.613, 1.707), ((-0.05, -0.20, 0.0), 0.917, 2.231)]):
        r = bg.clone()
        r[:, 0:3] += torch.tensor(shift)
        ang = math.radians(10.0 * (k + 1))
        R = [[math.cos(ang), -math.sin(ang), 0.0], [math.sin(ang), math.cos(ang), 0.0], [0.0, 0.0, 1.0]]
        r[:, 3:6] = synth.rotate_rows(r[:, 3:6], R)       # host-independent (an fp32 `@` rounds differently on different CPUs)
        r[:, 6], r[:, 7] = near, far
        miss = (torch.arange(n) % (4 + k)) == 0        # rays that miss the object's box: near = far = 0
        r[miss, 6:8] = 0.0
        sets.append(r.contiguous())
    # Exact z ties between DIFFERENT sets are resolved by torch.sort's unspecified (unstable) order in
    # the reference (multi_rendering.py:112) and by set order here; the near/far values above avoid them
    # (asserted below).  Ties at z == 0 (rays that miss their box) remain: they carry zero weight.
    zc = [r[:, 6:7] + (r[:, 7:8] - r[:, 6:7]) * torch.linspace(0, 1, MULTI["N_samples"]) for r in sets]
    for a in range(3):
        for b in range(a + 1, 3):
            tie = (zc[a][:, :, None] == zc[b][:, None, :]) & (zc[a][:, :, None] != 0)
            assert not tie.any(), "multi_inputs: cross-set z tie"
    box = synth.oriented_box(center=[2.9, 3.1, 0.5], size=[1.0, 0.8, 1.0], yaw_deg=20.0,
                             scene_center=synth.SCANNET_LIKE["scene_center"], scale_factor=synth.SCANNET_LIKE["scale_factor"])
    return sets, [box]


def multi_randoms():
    """injected draws of the training-mode render_rays_multi golden: u of sample_pdf(det=False) per ray set
    (multi_rendering.py:272-274 -> rendering.py:40) and the randn_like of each joint compositing (:126)"""
    g = torch.Generator().manual_seed(31)
    n, S, I, K = MULTI["n_rays"], MULTI["N_samples"], MULTI["N_importance"], len(MULTI["obj_ids"])
    return dict(u_rand=[torch.rand(n, I, generator=g) for _ in range(K)],
                noise=[torch.randn(n, K * S, generator=g), torch.randn(n, K * (S + I), generator=g)])


def multi_inputs_clip():
    """the same ray sets with the two extra columns of the reference's 10-column variant (multi_rendering.py:277-285):
    object sets carry (bbox_mask_near, bbox_mask_far) strictly inside their (near, far); the background set stays (N, 8)"""
    sets, boxes = multi_inputs()
    out = [sets[0]]
    for k, r in enumerate(sets[1:]):
        near, far = r[:, 6:7], r[:, 7:8]
        lo = near + (0.30 + 0.05 * k) * (far - near)
        hi = near + (0.55 + 0.05 * k) * (far - near)
        out.append(torch.cat([r, lo, hi], 1).contiguous())
    return out, boxes


# ---- bench.py --config 4 (BASELINE configs[4]): the editing demo's three ray sets of the 640x480 frame, 40 strided pixels ----
BENCH_MULTI = dict(obj_ids=[0, 4, 4], N_samples=64, N_importance=64, frame=(640, 480), n_rays=40, stride=6911, bbox_enlarge=0.06)


def bench_multi_geometry():
    """(focal, [Toc of the background set, of object 4, of its moved copy], removed-object box) of bench.py's config 4"""
    return synth.edit_demo_geometry(synth.SCANNET_LIKE, BENCH_MULTI["frame"][0])


def bench_multi_pixels():
    w, h = BENCH_MULTI["frame"]
    return torch.arange(0, w * h, BENCH_MULTI["stride"])[:BENCH_MULTI["n_rays"]]


# ---- editor ray generation (row f2): one background set and one object set with an enlarged box ----
RAYGEN = dict(H=36, W=48, fov_x_deg=60.0, near=0.15, far=3.0, bbox_enlarge=0.06)


def raygen_inputs():
    h, w = RAYGEN["H"], RAYGEN["W"]
    focal = (w / 2) / np.tan((RAYGEN["fov_x_deg"] / 2) / (180 / np.pi))       # editable_renderer.py:214
    cy, sy = math.cos(math.radians(35.0)), math.sin(math.radians(35.0))
    cp, sp = math.cos(math.radians(75.0)), math.sin(math.radians(75.0))
    R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]) @ np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Toc = np.concatenate([R, np.array([[0.5], [0.5], [0.6]])], 1)               # (3,4), translation in scene units
    box = synth.oriented_box(center=[3.6, 3.9, 0.5], size=[1.0, 0.8, 1.0], yaw_deg=20.0,
                             scene_center=synth.SCANNET_LIKE["scene_center"], scale_factor=synth.SCANNET_LIKE["scale_factor"])
    return h, w, float(focal), torch.from_numpy(Toc).float(), box


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}


# ---- image-scale parity: a whole 160x120 frame of each bench workload rendered by the REAL reference ----------------
# (VERDICT r2 missing #3 / SURVEY §8d "PSNR definition": utils/metrics.py:5-15 is a mean over a frame; the 48-ray cases
# above cannot carry that).  Same scenes, presets, weights and code assignment as bench.py --config 1 / 2 / 4, the camera
# of the preset at 160x120 (same pose and field of view: every fourth pixel direction of the 640x480 frame, roughly).
# Stored: the pixel maps only (+ the fine depths of every FRAME["sub"]-th ray for the moved-ray count, the float64
# oracle's distances from the reference per key = the fp32 noise floor, and its moved-ray count), see make_golden.frames().
FRAME = dict(W=160, H=120, sub=8, sub_multi=24)
FRAME_CASES = {
    "frame_toydesk2": dict(kind="single", render_case="bench_toydesk2"),            # BASELINE configs[1]
    "frame_scannet_multi": dict(kind="single", render_case="bench_scannet_multi"),  # BASELINE configs[2] / [3]
    "frame_edit_demo": dict(kind="multi"),                                          # BASELINE configs[4]
}
FRAME_MAPS = ["rgb", "depth", "opacity", "rgb_instance", "depth_instance", "opacity_instance"]


def frame_inputs(case):
    """single: (rays (n,8), ids (n), render_rays kwargs, scene name); multi: (focal, poses, box, scene name)"""
    fc = FRAME_CASES[case]
    w, h = FRAME["W"], FRAME["H"]
    if fc["kind"] == "single":
        c = RENDER_CASES[fc["render_case"]]
        rays = synth.preset_rays(SCENES[c["scene"]][2], w, h)
        n = rays.shape[0]
        ids = synth.per_ray_ids(n) if c["ids"] == "five" else torch.full((n,), int(c["ids"]), dtype=torch.long)
        kw = dict(c["kw"], perturb=0, noise_std=0)
        return rays, ids, kw, c["scene"]
    focal, poses, box = synth.edit_demo_geometry(synth.SCANNET_LIKE, w)
    return focal, poses, box, "scannet_800k"


# ---- BASELINE configs[1] at its own size: the whole 640x480 ToyDesk-2 frame rendered by the reference (make_golden.full_frame).
# Stored: rgb_fine of all 307,200 pixels as uint16 (x / 65535: quantisation rmse 4.4e-6 = 107 dB, far below the 74 dB the
# renders differ by) and depth_fine of every FULL_FRAME["sub"]-th pixel in fp32.
FULL_FRAME = dict(W=640, H=480, sub=16, render_case="bench_toydesk2")


def full_frame_inputs():
    c = RENDER_CASES[FULL_FRAME["render_case"]]
    rays = synth.preset_rays(SCENES[c["scene"]][2], FULL_FRAME["W"], FULL_FRAME["H"])
    n = rays.shape[0]
    ids = torch.full((n,), int(c["ids"]), dtype=torch.long)
    return rays, ids, dict(c["kw"], perturb=0, noise_std=0), c["scene"]


def frame_multi_sets(gen, case="frame_edit_demo"):
    """The three ray sets of the 160x120 editing-demo frame through a `generate_rays(H, W, focal, c2w, near, far, box=,
    bbox_enlarge=)`-shaped function (the oracle's, bit-equal to the reference's ray / box code; or the device kernel)."""
    focal, poses, box, _ = frame_inputs(case)
    pre, bm = synth.SCANNET_LIKE, BENCH_MULTI
    return [gen(FRAME["H"], FRAME["W"], focal, torch.from_numpy(np.asarray(T)).float(), pre["near"], pre["far"],
                box=None if k == 0 else box, bbox_enlarge=bm["bbox_enlarge"]) for k, T in enumerate(poses)]


def golden_multi_sets(case="frame_edit_demo"):
    """The ray sets the reference rendered the multi frame FROM, as stored with its golden (oracle/make_golden.py::frames): the
    reference's get_rays is an fp32 matmul + a vector norm, which round differently on different host CPUs, so sets regenerated
    on the test host are not bit-for-bit the rays of the golden frame (round 6)."""
    g = load_golden(case)
    sets = []
    k = 0
    while "_set%d_d" % k in g:
        d, nf = g["_set%d_d" % k], g["_set%d_nf" % k]
        sets.append(torch.cat([g["_set%d_o" % k].reshape(1, 3).expand(d.shape[0], 3), d, nf], 1).contiguous())
        k += 1
    return sets


# ---- input digests -------------------------------------------------------------------------------------------------------------
def _digest(t):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float32:
        return int(t.view(torch.int32).to(torch.int64).sum().item())
    if t.dtype == torch.float64:
        return int((t.view(torch.int64) >> 8).sum().item())
    return int(t.to(torch.int64).sum().item())


def input_digests(A=None):
    """{name: bit digest} of every synthetic input a golden depends on, as generated on THIS host: ray batches of every case,
    model weights, voxel tables / index maps / geometry, code tables, random draws.  tests/golden/input_digests.json holds the
    values of the host that made the goldens; a difference means a test would compare renders of different inputs (round 6: ray
    directions one ulp apart between the build container and the GPU box made every frame-scale coarse key look 1.6e-4 off)."""
    if A is None:
        import object_nerf_amd as A
    out = {}
    seen = set()

    def scene_digests(sname):
        if sname in seen:
            return
        seen.add(sname)
        sc = scene_for(A, sname)
        for typ in ("coarse", "fine"):
            out["scene.%s.%s.weights" % (sname, typ)] = sum(_digest(p) for p in sc.models[typ].parameters())
        out["scene.%s.codes" % sname] = _digest(sc.code_library.embedding_instance.weight)
        ev = sc.embeddings["xyz"]
        if hasattr(ev, "voxel_idx_map"):
            out["scene.%s.table" % sname] = _digest(ev.embedding_space_ftr.weight)
            out["scene.%s.idx_map" % sname] = _digest(ev.voxel_idx_map)
            out["scene.%s.geometry" % sname] = _digest(ev.voxel_offset) + _digest(ev.voxel_size.reshape(-1)) + _digest(ev.voxel_shape)
    for case in sorted(FRAME_CASES):
        fi = frame_inputs(case)
        if FRAME_CASES[case]["kind"] == "single":
            out["frame.%s.rays" % case] = _digest(fi[0])
            out["frame.%s.ids" % case] = _digest(fi[1])
        scene_digests(fi[3])
    out["full_frame.rays"] = _digest(full_frame_inputs()[0])
    for case in sorted(RENDER_CASES):
        rays, ids, ptm, randoms = render_inputs(case)
        out["render.%s.rays" % case] = _digest(rays)
        out["render.%s.ids" % case] = _digest(ids)
        if randoms is not None:
            out["render.%s.randoms" % case] = sum(_digest(t) for v in randoms.values() for t in (v if isinstance(v, (list, tuple)) else [v]))
        scene_digests(RENDER_CASES[case]["scene"])
    sets, boxes = multi_inputs()
    out["multi.sets"] = sum(_digest(t) for t in sets)
    out["voxel_points"] = _digest(voxel_points())
    return out

