"""CPU restatement of the reference's render_rays / render_rays_multi hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; object_nerf_amd/ never does (and fails loudly when
its HIP library is missing instead of falling back here).

What it is: an independent plain-PyTorch (CPU, fp32) statement of the algorithm of
  models/rendering.py            (render_rays 233-337, inference_model 64-230, sample_pdf 11-61)
  models/nerf_model.py           (ObjectNeRF.forward 97-121, forward_instance 123-152)
  models/embedding_helper.py     (Embedding.forward 57-74, EmbeddingVoxel.forward 325-411)
  render_tools/multi_rendering.py (16-325)
  utils/bbox_utils.py            (119-130, 158-207)
of zju3dv/object_nerf, written against plain tensors / state_dicts rather than the reference's
module types, each function citing the lines it follows.  It uses the same ATen CPU kernels the
reference's CPU path uses (torch 2.10), so on identical inputs it agrees with the reference to
the last bit where the op sequence is the same, and it is the timed "port" CPU baseline.

Pinning: the reference has NO tests or golden vectors for this path.  The oracle
is pinned instead against outputs of the reference ITSELF run in the build container
(oracle/make_golden.py imports /root/reference via oracle/ref_import.py and commits the
vectors under tests/golden/; tests/test_oracle_vs_golden.py re-checks them on CPU, and
tests/test_oracle_vs_reference.py compares live whenever the mount is present).
"""
import itertools

import torch

LEAKY_SLOPE = 0.01      # nn.LeakyReLU() default, nerf_model.py:38


# ---------------------------------------------------------------------------------------------
# embeddings
# ---------------------------------------------------------------------------------------------
# config.model entries the restatement is parameterised by (models/nerf_model.py:18-95; defaults = config/default_conf.yml:7-36);
# `arch` arguments below are dicts holding any subset of these keys
DEFAULT_ARCH = dict(D=8, skips=(4,), inst_D=4, inst_skips=(2,), n_freq_xyz=10, n_freq_dir=4, n_freq_voxel=6, logscale=True)


def _arch(arch):
    a = dict(DEFAULT_ARCH)
    a.update(arch or {})
    return a


def pos_encode(x, n_freqs, logscale=True):
    """Embedding.forward, embedding_helper.py:57-74: [x, sin(f0 x), cos(f0 x), sin(f1 x), ...], bands 2^k or (logscale=False,
    :54-55) torch.linspace(1, 2^(F-1), F)"""
    freqs = 2 ** torch.linspace(0, n_freqs - 1, n_freqs) if logscale else torch.linspace(1, 2 ** (n_freqs - 1), n_freqs)
    out = [x]
    for f in freqs:
        out.append(torch.sin(f * x))
        out.append(torch.cos(f * x))
    return torch.cat(out, -1)


def voxel_features(xyz, grid):
    """Trilinear sparse-voxel lookup, embedding_helper.py:331-411 (before the positional encoding).

    grid: dict(voxel_idx_map (X,Y,Z) int64, table (n,24), voxel_offset (3), voxel_size (), voxel_shape (3) int64)
    returns (N,24)
    """
    idx_map, table = grid["voxel_idx_map"], grid["table"]
    shape = grid["voxel_shape"]
    n = xyz.shape[0]
    s = (xyz + grid["voxel_offset"]) / grid["voxel_size"]                 # :361
    q = s.floor().long()                                                  # :363
    corners = [q + torch.tensor(c) for c in itertools.product([0, 1], repeat=3)]   # :364-368
    qa = torch.cat(corners, 0)
    invalid = ((qa < 0).sum(1) > 0) | ((qa >= shape).sum(1) > 0)          # :336-338
    qa = torch.where(invalid[:, None], torch.zeros_like(qa), qa)          # :339
    rows = idx_map[qa[:, 0], qa[:, 1], qa[:, 2]]                          # :342-344
    invalid = invalid | (rows < 0)                                        # :346-347
    rows = torch.where(invalid, torch.full_like(rows, table.shape[0] - 1), rows)   # :349
    ftr = table[rows]
    ftr = torch.where(invalid[:, None], torch.zeros_like(ftr), ftr)       # :351
    p = s - q.float()                                                     # :371
    u, v, w = p[:, 0], p[:, 1], p[:, 2]
    lu, lv, lw = 1 - u, 1 - v, 1 - w
    wts = torch.cat([lu * lv * lw, lu * lv * w, lu * v * lw, lu * v * w,
                     u * lv * lw, u * lv * w, u * v * lw, u * v * w], 0)  # :373-385
    return (ftr * wts.view(-1, 1)).view(8, n, -1).sum(0)                  # :387-389


def voxel_embed(xyz, grid, n_freq_voxel=6, n_freq_xyz=10, inst_c=8):
    """EmbeddingVoxel.forward, embedding_helper.py:325-329 + 403-409 -> (N,271), (N,104)"""
    f = voxel_features(xyz, grid)
    c = f.shape[1]
    scene, inst = f[:, : c - inst_c], f[:, c - inst_c:]
    scene_ftr = torch.cat([pos_encode(scene, n_freq_voxel), pos_encode(xyz, n_freq_xyz)], -1)
    return scene_ftr, pos_encode(inst, n_freq_voxel)


# ---------------------------------------------------------------------------------------------
# the two MLP branches (params: state_dict-style mapping name -> tensor)
# ---------------------------------------------------------------------------------------------
def _lin(params, name, x):
    return torch.addmm(params[name + ".bias"], x, params[name + ".weight"].t())


def _leaky(x):
    return torch.nn.functional.leaky_relu(x, LEAKY_SLOPE)


def mlp_scene(params, emb_xyz, emb_dir, D=8, skips=(4,), sigma_only=False):
    """ObjectNeRF.forward, nerf_model.py:97-121"""
    h = emb_xyz
    for i in range(D):
        if i in skips:
            h = torch.cat([emb_xyz, h], -1)
        h = _leaky(_lin(params, "xyz_encoding_%d.0" % (i + 1), h))
    sigma = _lin(params, "sigma", h)
    if sigma_only:
        return sigma, None
    final = _lin(params, "xyz_encoding_final", h)
    d = _leaky(_lin(params, "dir_encoding.0", torch.cat([final, emb_dir], -1)))
    return sigma, torch.sigmoid(_lin(params, "rgb.0", d))


def mlp_object(params, emb_xyz, emb_dir, obj_voxel, obj_code, inst_D=4, inst_skips=(2,), sigma_only=False):
    """ObjectNeRF.forward_instance, nerf_model.py:123-152"""
    parts = [emb_xyz] + ([obj_voxel] if obj_voxel is not None else []) + [obj_code]
    x_in = torch.cat(parts, -1)
    h = x_in
    for i in range(inst_D):
        if i in inst_skips:
            h = torch.cat([x_in, h], -1)
        h = _leaky(_lin(params, "instance_encoding_%d.0" % (i + 1), h))
    sigma = _lin(params, "instance_sigma", h)
    if sigma_only:
        return sigma, None
    final = _lin(params, "instance_encoding_final.0", h)
    d = _leaky(_lin(params, "inst_dir_encoding.0", torch.cat([final, emb_dir], -1)))
    return sigma, torch.sigmoid(_lin(params, "inst_rgb.0", d))


# ---------------------------------------------------------------------------------------------
# sampling / compositing
# ---------------------------------------------------------------------------------------------
def coarse_depths(rays, n_samples, use_disp=False, perturb=0.0, perturb_rand=None):
    """rendering.py:256-277"""
    near, far = rays[:, 6:7], rays[:, 7:8]
    t = torch.linspace(0, 1, n_samples)
    if not use_disp:
        z = near * (1 - t) + far * t
    else:
        z = 1 / (1 / near * (1 - t) + 1 / far * t)
    z = z.expand(rays.shape[0], n_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        r = perturb_rand if perturb_rand is not None else torch.rand_like(z)
        z = lower + (upper - lower) * (perturb * r)
    return z


def sample_pdf(bins, weights, n_importance, det=False, eps=1e-5, u=None):
    """rendering.py:11-61"""
    n, nw = weights.shape
    weights = weights + eps
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros(n, 1), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0, 1, n_importance).expand(n, n_importance)
    elif u is None:
        u = torch.rand(n, n_importance)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp_min(0)
    above = inds.clamp_max(nw)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


def alpha_weights(z, sigma, last_delta, noise=None, noise_std=0.0, alpha_mask=None):
    """alphas / transmittance / weights, rendering.py:140-162 (187-209 for the instance set)"""
    deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], last_delta)], -1)
    s = sigma if noise is None or noise_std == 0 else sigma + noise * noise_std
    alphas = 1 - torch.exp(-deltas * torch.relu(s))
    if alpha_mask is not None:
        alphas = torch.where(alpha_mask, torch.zeros_like(alphas), alphas)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    return alphas * torch.cumprod(shifted[:, :-1], -1)


def composite(z, sigma, rgb, inst_sigma=None, inst_rgb=None, noise=None, noise_inst=None, noise_std=0.0,
              white_back=False, use_zero_as_last_delta=False, occlusion=False, frustum_bound_th=0.0,
              pass_through_mask=None, rays_in_bbox=False):
    """Scene (rendering.py:139-182) and instance (185-229) compositing of one pass.
    Returns a dict with the un-suffixed result keys."""
    out = {}
    w = alpha_weights(z, sigma, 0.0 if use_zero_as_last_delta else 1e10, noise, noise_std)
    out["weights"] = w
    out["opacity"] = w.sum(1)
    out["z_vals"] = z
    rgb_map = (w[..., None] * rgb).sum(1)
    out["depth"] = (w * z).sum(1)
    if white_back:
        rgb_map = rgb_map + 1 - out["opacity"][:, None]
    out["rgb"] = rgb_map
    if inst_sigma is not None:
        mask = None
        if occlusion:                                                    # :192-202
            mask = (out["depth"][:, None] + frustum_bound_th) < z
            if pass_through_mask is not None:
                mask = mask & ~pass_through_mask.reshape(-1, 1).bool()
        wi = alpha_weights(z, inst_sigma, 0.0, noise_inst, noise_std, mask)
        out["opacity_instance"] = wi.sum(1)
        out["rgb_instance"] = (wi[..., None] * inst_rgb).sum(1) + 1 - out["opacity_instance"][:, None]   # :223
        out["depth_instance"] = (wi * z).sum(1)
        if rays_in_bbox:                                                 # :228-229
            out["weights"] = wi
    return out


# ---------------------------------------------------------------------------------------------
# density-grid query of the mesh tool
# ---------------------------------------------------------------------------------------------
def sigma_grid(params, grid, x, y, z, obj_code=None, chunk=32768):
    """tools/extract_mesh.py:62-113: sigma of `nerf_fine` on the lattice np.meshgrid(x, y, z) (float64 axes rounded to fp32 by
    torch.FloatTensor, :66), chunk by chunk: embedding_xyz -> forward(..., sigma_only=True) (:85-108), or forward_instance
    with ONE repeated code when obj_id > 0 (:97-104).  grid None = plain positional encoding.  Returns (nx*ny*nz, 1)."""
    import numpy as np
    xyz_ = torch.FloatTensor(np.stack(np.meshgrid(np.asarray(x), np.asarray(y), np.asarray(z)), -1).reshape(-1, 3))   # :66
    out = []
    for i in range(0, xyz_.shape[0], chunk):                                                          # :80
        p = xyz_[i:i + chunk]
        if grid is not None:
            e_xyz, e_obj = voxel_embed(p, grid)                                                       # :84-87
        else:
            e_xyz, e_obj = pos_encode(p, 10), None                                                    # :88-91
        if obj_code is not None:
            code = obj_code.reshape(1, -1).expand(p.shape[0], -1)                                     # :99-101
            out.append(mlp_object(params, e_xyz, None, e_obj, code, sigma_only=True)[0])              # :102-104
        else:
            out.append(mlp_scene(params, e_xyz, None, sigma_only=True)[0])                            # :106-108
    return torch.cat(out, 0)                                                                          # :111


# ---------------------------------------------------------------------------------------------
# render_rays
# ---------------------------------------------------------------------------------------------
def eval_points(params, grid, xyz, rays_d, codes, forward_instance=True, scene=True, chunk=32768, arch=None):
    """The MLP chunk loop of inference_model, rendering.py:86-137.  xyz (N,S,3); rays_d (N,3);
    codes (N,64).  grid None = plain positional encoding.  arch: non-default config.model entries (DEFAULT_ARCH)."""
    A = _arch(arch)
    n, s, _ = xyz.shape
    pts = xyz.reshape(-1, 3)
    emb_dir = pos_encode(rays_d, A["n_freq_dir"], A["logscale"]).repeat_interleave(s, 0)
    code_rep = codes.repeat_interleave(s, 0) if codes is not None else None
    outs = [[], [], [], []]
    for i in range(0, pts.shape[0], chunk):
        p = pts[i:i + chunk]
        if grid is not None:
            e_xyz, e_obj = voxel_embed(p, grid, n_freq_voxel=A["n_freq_voxel"])        # xyz: Embedding(3, 10) whatever the config (:84)
        else:
            e_xyz, e_obj = pos_encode(p, A["n_freq_xyz"], A["logscale"]), None
        if scene:
            sg, c = mlp_scene(params, e_xyz, emb_dir[i:i + chunk], D=A["D"], skips=A["skips"])
            outs[0].append(sg); outs[1].append(c)
        if forward_instance:
            sg, c = mlp_object(params, e_xyz, emb_dir[i:i + chunk], e_obj, code_rep[i:i + chunk], inst_D=A["inst_D"],
                               inst_skips=A["inst_skips"])
            outs[2].append(sg); outs[3].append(c)
    sigma = torch.cat(outs[0], 0).view(n, s) if scene else None
    rgb = torch.cat(outs[1], 0).view(n, s, 3) if scene else None
    isig = torch.cat(outs[2], 0).view(n, s) if forward_instance else None
    irgb = torch.cat(outs[3], 0).view(n, s, 3) if forward_instance else None
    return sigma, rgb, isig, irgb


def render_rays(params_coarse, params_fine, grid, rays, N_samples=64, use_disp=False, perturb=0.0, noise_std=0.0,
                N_importance=0, white_back=False, forward_instance=True, embedding_instance=None,
                frustum_bound_th=0.0, pass_through_mask=None, rays_in_bbox=False, is_eval=False,
                use_zero_as_last_delta=False, randoms=None, chunk=32768, z_fine_override=None, arch=None):
    """models/rendering.py:233-337.  randoms: optional {"perturb_rand","u_rand","noise":[4]}.  arch: see DEFAULT_ARCH.
    z_fine_override: test hook -- evaluate the fine pass at these (N, S+I) depths instead of the sampled ones
    (teacher forcing; removes the importance sampler's fp32 sensitivity from gradient comparisons)."""
    o, d = rays[:, 0:3], rays[:, 3:6]
    randoms = randoms or {}
    noise = randoms.get("noise", [None] * 4)
    z = coarse_depths(rays, N_samples, use_disp, perturb, randoms.get("perturb_rand"))
    results = {}

    def one_pass(typ, params, z, nz, nz_i):
        xyz = o[:, None, :] + d[:, None, :] * z[..., None]              # :279 / :316
        sg, c, isg, ic = eval_points(params, grid, xyz, d, embedding_instance, forward_instance, True, chunk, arch)
        r = composite(z, sg, c, isg, ic, nz, nz_i, noise_std, white_back, use_zero_as_last_delta,
                      (not is_eval) and frustum_bound_th > 0, frustum_bound_th, pass_through_mask, rays_in_bbox)
        for k, v in r.items():
            results["%s_%s" % (k, typ)] = v

    one_pass("coarse", params_coarse, z, noise[0], noise[1])
    if N_importance > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        # .detach(): no gradient from the fine depths back into the coarse weights (rendering.py:307)
        z_new = sample_pdf(mid, results["weights_coarse"][:, 1:-1].detach(), N_importance, det=(perturb == 0),
                           u=randoms.get("u_rand"))
        z = torch.sort(torch.cat([z, z_new], -1), -1)[0]
        if z_fine_override is not None:
            z = z_fine_override
        one_pass("fine", params_fine, z, noise[2], noise[3])
    return results


# ---------------------------------------------------------------------------------------------
# ray generation for the editor
# ---------------------------------------------------------------------------------------------
def get_ray_directions(H, W, focal):
    """datasets/ray_utils.py:5-25 (kornia.create_meshgrid(H, W, False): grid[y, x] = (x, y))"""
    j, i = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """datasets/ray_utils.py:28-51: rotate, normalise, broadcast the origin"""
    d = directions @ c2w[:, :3].T
    d = d / torch.norm(d, dim=-1, keepdim=True)
    o = c2w[:, 3].expand(d.shape)
    return o.reshape(-1, 3), d.reshape(-1, 3)


def ray_box_near_far(rays_o, rays_d, box, bbox_enlarge=0.0):
    """BBoxRayHelper.get_ray_bbox_intersections (utils/bbox_utils.py:100-117, 132-156) with the slab test of
    datasets/geo_utils.py:126-162, vectorised in float64.  Returns hit (N) bool, near (N,1), far (N,1) fp32
    already divided by scale_factor.  Mirrors the reference's quirks: the direction is rotated by the box
    rotation only (bbox_utils.py:115), zero direction components become 1e-14 (geo_utils.py:131), a ray whose
    origin is inside the box (tmin < 0) counts as a miss (158-160)."""
    import numpy as np
    sf = box["scale_factor"]
    R_avg, t_avg = np.asarray(box["R_avg"], dtype=np.float64), np.asarray(box["t_avg"], dtype=np.float64)
    R_box, t_box = np.asarray(box["R_box"], dtype=np.float64), np.asarray(box["t_box"], dtype=np.float64)
    o = rays_o.detach().cpu().numpy() * sf
    d = rays_d.detach().cpu().numpy()
    o = (R_avg @ o.T).T + t_avg
    o = (R_box @ o.T).T + t_box
    d = (R_box @ d.T).T
    lo = np.asarray(box["bmin"], dtype=np.float64) - (bbox_enlarge if bbox_enlarge > 0 else 0.0)
    hi = np.asarray(box["bmax"], dtype=np.float64) + (bbox_enlarge if bbox_enlarge > 0 else 0.0)
    d = np.where(d == 0, 1.0e-14, d)
    inv = 1 / d
    t0 = (np.where(inv < 0, hi, lo) - o) * inv
    t1 = (np.where(inv < 0, lo, hi) - o) * inv
    tmin, tmax = t0[:, 0].copy(), t1[:, 0].copy()
    hit = np.ones(o.shape[0], dtype=bool)
    for a in (1, 2):
        hit &= ~((tmin > t1[:, a]) | (t0[:, a] > tmax))
        tmin = np.where(t0[:, a] > tmin, t0[:, a], tmin)
        tmax = np.where(t1[:, a] < tmax, t1[:, a], tmax)
    hit &= ~((tmin < 0) | (tmax < 0))
    near = torch.Tensor(np.where(hit, tmin, 0.0)[:, None]) / sf
    far = torch.Tensor(np.where(hit, tmax, 0.0)[:, None]) / sf
    return torch.from_numpy(hit), near, far


def generate_rays(H, W, focal, c2w, near, far, box=None, bbox_enlarge=0.0):
    """render_tools/editable_renderer.py:153-181 + 213-215, 257: (H*W, 8) rays of one ray set.
    box None: constant near/far (background, id 0); else near/far from the box, 0/0 where missed."""
    o, d = get_rays(get_ray_directions(H, W, focal), c2w)
    if box is None:
        n_, f_ = near * torch.ones_like(o[:, :1]), far * torch.ones_like(o[:, :1])
    else:
        _, n_, f_ = ray_box_near_far(o, d, box, bbox_enlarge)
    return torch.cat([o, d, n_, f_], 1)


# ---------------------------------------------------------------------------------------------
# multi-object path
# ---------------------------------------------------------------------------------------------
def points_in_boxes(xyz, boxes):
    """check_in_any_boxes, utils/bbox_utils.py:189-207 with check_xyz_in_bounds 158-186 and
    transform_xyz_to_bbox_coordinates 119-130.  boxes: list of dicts scale_factor, R_avg (3,3),
    t_avg (3), R_box (3,3), t_box (3), bmin (3), bmax (3) (float64, bounds already enlarged)."""
    import numpy as np
    shp = xyz.shape[:-1]
    pts = xyz.reshape(-1, 3).detach().cpu().numpy()
    inside = torch.zeros(pts.shape[0], dtype=torch.bool)
    for b in boxes:
        p = pts * b["scale_factor"]
        p = (np.asarray(b["R_avg"], dtype=np.float64) @ p.T).T + np.asarray(b["t_avg"], dtype=np.float64)
        p = (np.asarray(b["R_box"], dtype=np.float64) @ p.T).T + np.asarray(b["t_box"], dtype=np.float64)
        pt = torch.from_numpy(p).float()
        lo = [float(np.float32(v)) for v in b["bmin"]]
        hi = [float(np.float32(v)) for v in b["bmax"]]
        ib = torch.ones(pts.shape[0], dtype=torch.bool)
        for a in range(3):
            ib &= (pt[:, a] >= lo[a]) & (pt[:, a] <= hi[a])
        inside |= ib
    return inside.view(*shp)


def composite_multi(z_list, rgb_list, sigma_list, noise_std=0.0, white_back=False, noise=None):
    """volume_rendering_multi, multi_rendering.py:96-157 (stable joint sort; last delta 0).  noise: the (N, K*S) draw of
    `torch.randn_like(sigmas)` (:126) -- it meets the sigmas AFTER the sort, i.e. it is indexed by sorted position."""
    z = torch.cat(z_list, 1)
    rgb = torch.cat(rgb_list, 1)
    sg = torch.cat(sigma_list, 1)
    ids = torch.cat([torch.full_like(s, float(i)) for i, s in enumerate(sigma_list)], 1)
    z, idx = torch.sort(z, dim=-1, stable=True)
    rgb = torch.gather(rgb, 1, idx[..., None].expand(-1, -1, 3))
    sg = torch.gather(sg, 1, idx)
    ids = torch.gather(ids, 1, idx)
    w = alpha_weights(z, sg, 0.0, noise, noise_std)
    out = {"weights": w, "opacity": w.sum(1), "z_vals": z, "obj_ids": ids}
    rgb_map = (w[..., None] * rgb).sum(1)
    if white_back:
        rgb_map = rgb_map + 1 - out["opacity"][:, None]
    out["rgb"] = rgb_map
    out["depth"] = (w * z).sum(1)
    return out


def render_rays_multi(params_coarse, params_fine, grid, code_table, rays_list, obj_instance_ids, N_samples=64,
                      use_disp=False, perturb=0.0, noise_std=0.0, N_importance=0, white_back=False,
                      skip_boxes=None, chunk=32768, randoms=None, arch=None):
    """render_rays_multi, multi_rendering.py:160-325.  skip_boxes: list of box dicts for points_in_boxes, applied to the
    id-0 (background) ray set.  Training-mode draws (perturb != 0: sample_pdf(det=False) draws torch.rand(N, I) once per
    ray set, :272-274 -> rendering.py:40; noise_std != 0: one torch.randn_like per joint compositing, :126) are taken
    from randoms = {"u_rand": [K x (N, I)], "noise": [(N, K*S), (N, K*(S+I))]} when given, else drawn here."""
    K = len(rays_list)

    def branch(params, rays, z, oid):
        o, d = rays[:, 0:3], rays[:, 3:6]
        xyz = o[:, None, :] + d[:, None, :] * z[..., None]
        n = rays.shape[0]
        if oid > 0:   # object branch with that id's code (multi_rendering.py:45-51, 63-69)
            codes = code_table[oid][None].expand(n, -1)
            _, _, sg, c = eval_points(params, grid, xyz, d, codes, True, False, chunk, arch)
        else:
            sg, c, _, _ = eval_points(params, grid, xyz, d, None, False, True, chunk, arch)
        sg = sg.clone()
        sg[z[:, -1] == 0] = -1e5                                          # :40,83,92
        if oid == 0 and skip_boxes:
            sg[points_in_boxes(xyz, skip_boxes)] = -1e5                   # :239-241
        return c, sg

    zs = [coarse_depths(r, N_samples, use_disp) for r in rays_list]
    cs, sgs = zip(*[branch(params_coarse, rays_list[i], zs[i], obj_instance_ids[i]) for i in range(K)])
    res = {}
    rnd = randoms or {}
    nz = rnd.get("noise", [None, None])
    if noise_std != 0 and nz[0] is None:
        nz = [torch.randn(rays_list[0].shape[0], K * N_samples), torch.randn(rays_list[0].shape[0], K * (N_samples + N_importance))]
    r = composite_multi(list(zs), list(cs), list(sgs), noise_std, white_back, noise=nz[0])
    for k, v in r.items():
        res["%s_coarse" % k] = v
    if N_importance > 0:
        zf, cf, sf = [], [], []
        for i in range(K):
            n = rays_list[i].shape[0]
            w_own = res["weights_coarse"][res["obj_ids_coarse"] == i].view(n, N_samples)   # :269-271
            mid = 0.5 * (zs[i][:, :-1] + zs[i][:, 1:])
            z_new = sample_pdf(mid, w_own[:, 1:-1].detach(), N_importance, det=(perturb == 0),
                               u=rnd["u_rand"][i] if (perturb != 0 and "u_rand" in rnd) else None)
            z = torch.sort(torch.cat([zs[i], z_new], -1), -1)[0]
            if rays_list[i].shape[1] == 10:                               # :277-285 ray mask (e.g. bbox): clip z values
                lo, hi = rays_list[i][:, 8:9], rays_list[i][:, 9:10]
                inside = torch.logical_and(z > lo, z < hi)
                z = torch.where(inside, hi.expand_as(z), z)
            c, sg = branch(params_fine, rays_list[i], z, obj_instance_ids[i])
            zf.append(z); cf.append(c); sf.append(sg)
        r = composite_multi(zf, cf, sf, noise_std, white_back, noise=nz[1])
        for k, v in r.items():
            if k != "obj_ids":
                res["%s_fine" % k] = v
    return res
