#!/usr/bin/env python
"""Runs the reference's CALLERS of the hot path, unmodified, in the build container (b, f3).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/test_reference_callers.py (CPU, skipped
where /root/reference is absent) and to generate tests/golden/callers_*.npz.  Nothing in the product imports it.

The callers are the REAL files, imported from where they lie:
  * train.py::ObjectNeRFSystem -- __init__ (36-71: builds ObjectNeRF / Embedding / EmbeddingVoxel / CodeLibrary through
    `from models.* import ...`), forward (73-105: the ray-chunk loop), training_step (147-180), validation_step (182-224);
  * render_tools/editable_renderer.py::EditableRenderer -- __init__/load_model (53-80: Lightning `load_from_checkpoint`),
    generate_rays (153-181), render_edit (203-294: per-object ray sets + the chunk loop over render_rays_multi),
    render_origin / scene_inference (112-151, 183-201).
Their un-installed dependencies (pytorch_lightning, omegaconf, cv2, torchvision, kornia, numba, open3d, torch_optimizer)
are stub modules; `datasets/__init__.py` (cv2, torchvision) is bypassed as in oracle/ref_import.py.

Two flavours, each in its OWN process because both own the module names `models`, `train`, ...:

  reference   sys.path = [reference]: the callers run over the reference's models/* on the CPU.  Writes
              <work>/callers.npz (every scenario's outputs) and <work>/reference.ckpt, a Lightning-keyed checkpoint
              ({"state_dict": system.state_dict()}) of the reference's own module types.
  dropin      sys.path = [repo/dropin, reference]: the SAME caller files now bind `models.*` and
              `render_tools.multi_rendering` to object_nerf_amd.  The system is constructed by the real
              train.py::__init__ from the drop-in types, <work>/reference.ckpt is loaded by the real
              EditableRenderer.load_model -> load_from_checkpoint (strict), and the callers run.  There is no GPU in the
              build container and the product has no CPU path, so the two render entry points are replaced by a STAND-IN
              that (1) binds the call against the drop-in function's real signature (a caller/argument mismatch fails
              here), (2) records the call (tensors + scalars) and (3) computes the result with the CPU oracle from the
              drop-in modules' own parameters.  Outputs must equal the reference flavour's bit for bit; the recorded
              calls are written to <work>/calls.npz.  Also writes <work>/dropin.ckpt (checkpoint.export_state_dict).
  reference-load   loads <work>/dropin.ckpt into the reference-typed system (strict) and re-renders one scenario.
  checkpoint  writes <work>/reference_small.ckpt (a small scene, reference module types, Lightning layout) and
              <work>/ckpt_render.npz (what the reference renders from it): the committed fixture of the `-m gpu`
              checkpoint test (tests/test_gpu_checkpoint.py; `python oracle/ref_callers.py checkpoint tests/golden`).

tests/golden/callers_{outputs,calls}.npz are these files, committed; the `-m gpu` test replays the recorded calls
through the HIP entry points on the GPU box (where the reference is absent) and compares with the real callers' outputs.
"""
import inspect
import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OBJNERF_REFERENCE_ROOT", "/root/reference")

# scenario constants (the GPU replay needs none of them: everything it uses is in the recorded calls)
N_RAYS, CHUNK = 40, 16            # 3 chunks, the last one ragged (8 rays)
EDIT_HW, EDIT_CHUNK = (10, 12), 50  # 120 pixels -> chunks of 50, 50, 20
MAX_VOXELS = 120_000


def _attr(d):
    from object_nerf_amd.config import AttrDict
    return AttrDict({k: (_attr(v) if isinstance(v, dict) else v) for k, v in d.items()})


def install_caller_stubs():
    from oracle import ref_import
    ref_import.install_stubs()
    sys.modules["datasets"].dataset_dict = {}                                     # train.py:9
    # --- omegaconf: the callers only touch OmegaConf in main()/read_testing_config(), never on the paths run here
    om = types.ModuleType("omegaconf")
    om.OmegaConf = type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", om)
    # --- pytorch_lightning: LightningModule = nn.Module + the three members the callers use
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        current_epoch = 0
        global_step = 0

        def log(self, *a, **k):
            self.__dict__.setdefault("_logged", {})[a[0]] = a[1]

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, **kwargs):
            """Lightning's contract: construct with the given kwargs, then strict load of checkpoint['state_dict']"""
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
            obj = cls(**kwargs)
            obj.load_state_dict(ckpt["state_dict"], strict=True)
            return obj

    pl.LightningModule = LightningModule
    pl.Trainer = type("Trainer", (), {})
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.ModelCheckpoint = type("ModelCheckpoint", (), {})
    lg = types.ModuleType("pytorch_lightning.loggers")
    lg.TensorBoardLogger = type("TensorBoardLogger", (), {})
    pl.callbacks, pl.loggers = cb, lg
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.callbacks": cb, "pytorch_lightning.loggers": lg})
    # --- cv2 / torchvision.transforms (utils/train_helper.py:1,5: visualisation only), kornia.losses.ssim (utils/metrics.py:2)
    cv2 = types.ModuleType("cv2")
    cv2.COLORMAP_JET = 2                                                # default argument at utils/train_helper.py:8
    sys.modules.setdefault("cv2", cv2)
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tv.transforms)
    kl = types.ModuleType("kornia.losses")
    kl.ssim = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("kornia.losses.ssim stub"))
    sys.modules["kornia"].losses = kl
    sys.modules.setdefault("kornia.losses", kl)


def make_config(work, n_points=200_000, max_voxels=MAX_VOXELS):
    """the training config the callers read (config/default_conf.yml + config/scannet_base_0113_multi.yml shapes)"""
    from object_nerf_amd import synth
    from object_nerf_amd.config import default_model_config
    from oracle import ref_import
    extra = dict(synth.dataset_extra(synth.SCANNET_LIKE, n_points))
    cloud = extra.pop("pcd_xyz")
    ref_import.POINT_CLOUDS["callers.ply"] = np.asarray(cloud)
    extra["pcd_path"] = "callers.ply"                                 # read through the stub open3d by BOTH flavours
    extra.update(near=synth.SCANNET_LIKE["near"], far=synth.SCANNET_LIKE["far"])
    model = dict(default_model_config(use_voxel_embedding=True, N_max_voxels=max_voxels, N_importance=64))
    model.update(frustum_bound=0.05)
    cfg = dict(model=model, dataset_extra=extra, dataset_name="scannet_base", img_wh=[12, 10],
               train=dict(chunk=CHUNK, optimizer="adam", lr=5e-4, weight_decay=0, lr_scheduler="steplr", decay_step=[100],
                          decay_gamma=0.5, num_epochs=1, progressive_train=False, warmup_epochs=0),
               loss=dict(color_loss_weight=1.0, depth_loss_weight=0.1, opacity_loss_weight=10.0,
                         instance_color_loss_weight=1.0, instance_depth_loss_weight=0.1))
    return _attr(cfg)


def fill_system(system):
    """the seeded "W1" fill of tests/cases.py scene 'voxel' (synth.build_scene): the GPU test rebuilds the same state"""
    from object_nerf_amd import synth
    synth.fill_table(system.embedding_xyz, 0)
    synth.fill_w1(system.nerf_coarse, 1)
    synth.fill_w1(system.nerf_fine, 2)
    synth.fill_codes(system.code_library, 0)


def make_batch(train):
    """Batches shaped like the DataLoader's (datasets/generic_dataset.py:470-500 + default collation):
    val  : one image per batch -> rays (1,N,8), rgbs (1,N,3), masks / ids / depths (1,N);
    train: B independent rays  -> rays (B,8), rgbs (B,3), depths / valid_mask (B,), the per-instance columns (B,1)."""
    from object_nerf_amd import synth
    g = torch.Generator().manual_seed(77)
    rays_all = synth.camera_rays(64, 48)
    rays = rays_all[torch.arange(0, rays_all.shape[0], 71)[:N_RAYS]].contiguous()
    n = rays.shape[0]
    rgbs, depths = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 2.0
    valid, inst = torch.arange(n) % 7 != 0, torch.arange(n) % 3 == 0
    imw, ids = 0.5 + torch.rand(n, generator=g), synth.per_ray_ids(n, seed=5)
    if not train:
        return dict(rays=rays[None], rgbs=rgbs[None], valid_mask=valid[None], instance_mask=inst[None],
                    instance_mask_weight=imw[None], instance_ids=ids[None], depths=depths[None])
    return dict(rays=rays, rgbs=rgbs, depths=depths, valid_mask=valid, instance_mask=inst[:, None],
                instance_mask_weight=imw[:, None], instance_ids=ids[:, None],
                pass_through_mask=(torch.arange(n) % 5 == 1)[:, None])


def render_randoms(seed):
    """random tensors of the training-mode render (perturb = 1, noise_std = 1), per chunk, in the reference's draw order"""
    g = torch.Generator().manual_seed(seed)
    S, I = 64, 64
    out = []
    for lo in range(0, N_RAYS, CHUNK):
        n = min(CHUNK, N_RAYS - lo)
        out.append(dict(perturb_rand=torch.rand(n, S, generator=g), u_rand=torch.rand(n, I, generator=g),
                        noise=[torch.randn(n, S, generator=g), torch.randn(n, S, generator=g),
                               torch.randn(n, S + I, generator=g), torch.randn(n, S + I, generator=g)]))
    return out


# ----------------------------------------------------------------------------------------------------------------
# the stand-in device of the dropin flavour
# ----------------------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.calls = []          # [(scenario, fn, scalars: dict, tensors: dict name -> tensor)]
        self.scenario = None
        self.randoms = None      # queue of per-call random dicts (training mode)

    def _params(self, m):
        return dict(m.named_parameters())

    def render_rays(self, **kw):
        import object_nerf_amd as A
        from oracle import objnerf_oracle as O
        import helpers as H
        inspect.signature(A.render_rays).bind(**kw)                   # the caller's keywords fit the drop-in signature
        models, emb = kw["models"], kw["embeddings"]
        assert isinstance(models["coarse"], A.ObjectNeRF) and isinstance(emb["xyz"], A.EmbeddingVoxel)
        scal = {k: v for k, v in kw.items() if not isinstance(v, (torch.Tensor, dict))}
        tens = {k: v.detach().clone() for k, v in kw.items() if isinstance(v, torch.Tensor)}
        rnd = self.randoms.pop(0) if self.randoms else None
        if rnd is not None:
            tens.update(perturb_rand=rnd["perturb_rand"], u_rand=rnd["u_rand"])
            tens.update({"noise%d" % i: t for i, t in enumerate(rnd["noise"])})
        self.calls.append((self.scenario, "render_rays", scal, tens))
        okw = {k: v for k, v in kw.items() if k not in ("models", "embeddings", "rays", "chunk")}
        return O.render_rays(self._params(models["coarse"]), self._params(models["fine"]), H.oracle_grid(emb["xyz"], keep_graph=True),
                             kw["rays"], randoms=rnd, chunk=kw["chunk"], **okw)     # same point chunking as the reference: same BLAS shapes

    def render_rays_multi(self, **kw):
        import object_nerf_amd as A
        from object_nerf_amd import bbox
        from oracle import objnerf_oracle as O
        import helpers as H
        from object_nerf_amd.multi_rendering import render_rays_multi as product_fn
        inspect.signature(product_fn).bind(**kw)
        models, emb = kw["models"], kw["embeddings"]
        assert isinstance(models["coarse"], A.ObjectNeRF) and isinstance(kw["code_library"], A.CodeLibrary)
        scal = {k: v for k, v in kw.items() if isinstance(v, (int, float, bool, list)) and k != "rays_list"}
        tens = {"rays_%d" % i: r.detach().clone() for i, r in enumerate(kw["rays_list"])}
        boxes = kw.get("background_skip_bbox") or {}
        # the packed (n_boxes, 31) float64 rows are what the product derives from the helpers (object_nerf_amd/bbox.py)
        tens["boxes"] = bbox.pack_boxes(boxes, "cpu")
        self.calls.append((self.scenario, "render_rays_multi", scal, tens))
        box_dicts = [H.box_dict_from_helper(b) for b in boxes.values()]
        with torch.no_grad():
            return O.render_rays_multi(self._params(models["coarse"]), self._params(models["fine"]), H.oracle_grid(emb["xyz"]),
                                       kw["code_library"].embedding_instance.weight, kw["rays_list"], kw["obj_instance_ids"],
                                       N_samples=kw["N_samples"], use_disp=kw["use_disp"], perturb=kw["perturb"],
                                       noise_std=kw["noise_std"], N_importance=kw["N_importance"], white_back=kw["white_back"],
                                       skip_boxes=box_dicts, chunk=kw["chunk"])


    # ---- row f2: the device ray generation the shims `dropin/datasets/ray_utils.py`, `dropin/utils/bbox_utils.py` route to ----
    def get_ray_directions(self, H, W, focal, device="cuda"):
        """what `get_ray_directions(h, w, focal).cuda()` (editable_renderer.py:191, 215) reaches on the drop-in: the grid
        WRITTEN on the device (dropin/datasets/ray_utils.py::HostDirections) -- here the oracle's grid, recorded"""
        from oracle import objnerf_oracle as O
        inspect.signature(self.product["get_ray_directions"]).bind(H, W, focal, device=device)
        d = O.get_ray_directions(H, W, focal)
        self.calls.append((self.scenario, "get_ray_directions", dict(H=int(H), W=int(W), focal=float(focal)), dict(out=d.clone())))
        return d

    def get_rays(self, directions, c2w):
        from oracle import objnerf_oracle as O
        inspect.signature(self.product["get_rays"]).bind(directions, c2w)
        rays_o, rays_d = O.get_rays(directions, c2w)
        self.calls.append((self.scenario, "get_rays", {}, dict(directions=directions.detach().clone(), c2w=c2w.detach().clone(),
                                                                 out_rays_o=rays_o.clone(), out_rays_d=rays_d.clone())))
        return rays_o, rays_d

    def ray_bbox_intersections(self, box, rays_o, rays_d, scale_factor=None, bbox_enlarge=0):
        from object_nerf_amd import bbox
        from oracle import objnerf_oracle as O
        import helpers as H
        inspect.signature(self.product["ray_bbox_intersections"]).bind(box, rays_o, rays_d, scale_factor, bbox_enlarge)
        bd = H.box_dict_from_helper(box)
        if scale_factor is not None:
            bd["scale_factor"] = float(scale_factor)
        hit, near, far = O.ray_box_near_far(rays_o, rays_d, bd, bbox_enlarge)
        row = torch.from_numpy(bbox._box_row(box, scale_factor, 0.0))     # what the product hands to the kernel
        self.calls.append((self.scenario, "ray_bbox_intersections", dict(bbox_enlarge=float(bbox_enlarge)),
                           dict(rays_o=rays_o.detach().clone(), rays_d=rays_d.detach().clone(), box=row,
                                out_hit=hit.clone(), out_near=near.clone(), out_far=far.clone())))
        return hit, near, far


def save_calls(path, calls):
    arrs, meta = {}, []
    for i, (scn, fn, scal, tens) in enumerate(calls):
        meta.append(dict(scenario=scn, fn=fn, scalars=scal, tensors=sorted(tens)))
        for k, t in tens.items():
            arrs["c%d_%s" % (i, k)] = t.numpy()
    arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)


# ----------------------------------------------------------------------------------------------------------------
# scenarios (identical code for both flavours: only what `train` / `editable_renderer` resolved to differs)
# ----------------------------------------------------------------------------------------------------------------
def run_scenarios(flavour, work, rec):
    from oracle import ref_import
    from object_nerf_amd import synth
    import train as T
    import render_tools.editable_renderer as ER
    assert T.__file__.startswith(REF) and ER.__file__.startswith(REF)
    want_mod = "object_nerf_amd." if flavour == "dropin" else "models."
    assert T.render_rays.__module__.startswith(want_mod), T.render_rays.__module__
    assert ER.render_rays_multi.__module__.startswith(want_mod.replace("models.", "render_tools.")), ER.render_rays_multi.__module__
    if rec is not None:
        T.render_rays = lambda **kw: rec.render_rays(**kw)
        ER.render_rays_multi = lambda **kw: rec.render_rays_multi(**kw)
        # row f2: the editor's ray generation binds to the device shims (editable_renderer.py:18,21 unchanged) ...
        import datasets.ray_utils as DRU
        import utils.bbox_utils as DBU
        import datasets.geo_utils as GEO
        from object_nerf_amd import bbox as hip_bbox, ray_utils as hip_rays
        assert DRU.__file__.startswith(os.path.join(ROOT, "dropin")) and DBU.__file__.startswith(os.path.join(ROOT, "dropin"))
        assert ER.get_rays is DRU.get_rays and ER.BBoxRayHelper is DBU.BBoxRayHelper
        assert issubclass(DBU.BBoxRayHelper, sys.modules["utils._reference_bbox_utils"].BBoxRayHelper)
        # ... "tensor is on the GPU" is always true for the stand-in device (as Tensor.cuda() is the identity here) ...
        DRU._on_device = DBU._on_device = lambda t: True
        # ... the product functions the shims call are the recording stand-in ...
        rec.product = dict(get_rays=hip_rays.get_rays, ray_bbox_intersections=hip_bbox.ray_bbox_intersections,
                           get_ray_directions=hip_rays.get_ray_directions)
        hip_rays.get_rays = rec.get_rays
        hip_rays.get_ray_directions = rec.get_ray_directions
        hip_bbox.ray_bbox_intersections = rec.ray_bbox_intersections

        # ... and the host slab test must never run: datasets/geo_utils.py:111-162 raises from here on
        def _host_slab_test(*a, **k):
            raise AssertionError("datasets/geo_utils.py::bbox_intersection[_batch] was called on the drop-in path")
        GEO.bbox_intersection_batch = GEO.bbox_intersection = _host_slab_test
        sys.modules["utils._reference_bbox_utils"].bbox_intersection_batch = _host_slab_test
    cfg = make_config(work)
    out = OrderedDict()

    def scenario(name):
        if rec is not None:
            rec.scenario = name

    def run_with_randoms(fn, rnd):
        """the same pre-drawn random tensors for both flavours: queued into torch.rand_like / rand / randn_like for the
        reference's render_rays (rendering.py:276, 40, 156, 187), handed to the stand-in for the drop-in's"""
        if rec is not None:
            rec.randoms = [dict(r) for r in rnd]
            return fn()
        q = dict(rand_like=[], rand=[], randn_like=[])
        for r in rnd:
            q["rand_like"].append(r["perturb_rand"]); q["rand"].append(r["u_rand"]); q["randn_like"] += list(r["noise"])
        with ref_import.inject_randoms(**q):
            return fn()

    # ---------------- ObjectNeRFSystem: constructed by the real __init__, filled, check-pointed ----------------
    system = T.ObjectNeRFSystem(cfg)
    assert type(system.nerf_coarse).__module__.startswith(want_mod)
    fill_system(system)
    ckpt = os.path.join(work, "%s.ckpt" % flavour)
    if flavour == "reference":
        torch.save({"state_dict": system.state_dict(), "epoch": 0}, ckpt)
    else:
        # f3, export direction: the product's exporter writes the Lightning layout from the drop-in operator set
        from object_nerf_amd import checkpoint
        from object_nerf_amd.config import AttrDict
        sc = AttrDict(models=system.models, embeddings=system.embeddings, code_library=system.code_library)
        exported = checkpoint.export_state_dict(sc)
        assert sorted(exported) == sorted(system.state_dict()), "export_state_dict keys differ from the LightningModule's"
        torch.save({"state_dict": exported, "epoch": 0}, ckpt)
    system.train_dataset = types.SimpleNamespace(white_back=False, is_rays_in_bbox=lambda: False)
    system.val_dataset = types.SimpleNamespace(white_back=False, is_rays_in_bbox=lambda: True)
    system.configure_optimizers()                                      # train.py:116-119 (training_step logs the lr)

    captured = {}
    system.loss.register_forward_pre_hook(lambda m, args: captured.update(results=args[0]))   # the dict forward() returned

    # ---- validation_step (182-224) -> forward chunk loop (73-105), eval mode, rays_in_bbox from the val dataset ----
    scenario("validation_step")
    system.eval()
    batch = make_batch(train=False)
    # train.py:88-89 passes config.model.perturb / noise_std (1 / 1 in config/default_conf.yml) in validation too
    with torch.no_grad():
        log = run_with_randoms(lambda: system.validation_step(batch, 1), render_randoms(79))
    for k, v in captured["results"].items():
        out["val_" + k] = v.detach().clone()
    out["val__loss"], out["val__psnr"] = log["val_loss"].detach(), log["val_psnr"].detach()

    # ---- training_step (147-180) + loss.backward(): perturb = 1, noise_std = 1, occlusion mask, pass-through mask ----
    scenario("training_step")
    system.train()
    batch = make_batch(train=True)
    loss = run_with_randoms(lambda: system.training_step(batch, 0), render_randoms(78))
    res = captured["results"]
    keys = sorted(res)
    for k in keys:
        if res[k].requires_grad:
            res[k].retain_grad()
    loss.backward()
    out["train__loss"] = loss.detach().clone()
    for k in keys:
        out["train_" + k] = res[k].detach().clone()
        if res[k].grad is not None:
            out["train_dL_" + k] = res[k].grad.detach().clone()
    named = dict(system.named_parameters())
    table_g = named["embedding_xyz.embedding_space_ftr.weight"].grad
    rows = table_g.abs().sum(1).nonzero().squeeze(1)
    out["train_grad__table_rows"], out["train_grad__table_vals"] = rows, table_g[rows]
    for k, p in named.items():
        if k != "embedding_xyz.embedding_space_ftr.weight" and p.grad is not None:
            # fixture size: the L2 norm of every gradient, and at most ~1000 strided entries of it
            flat = p.grad.detach().reshape(-1)
            out["train_gradnorm_" + k] = flat.double().norm().float()
            out["train_grad_" + k] = flat[::max(1, flat.numel() // 1000)].clone()
    system.zero_grad()

    # ---------------- EditableRenderer: real __init__ -> load_model -> load_from_checkpoint of the REFERENCE's checkpoint ----
    ecfg = _attr(dict(ckpt_path=os.path.join(work, "reference.ckpt"), ckpt_config=cfg, chunk=EDIT_CHUNK,
                      ckpt_config_path="unused"))
    renderer = ER.EditableRenderer(ecfg)
    assert type(renderer.system.nerf_fine).__module__.startswith(want_mod)
    for (k, a), (k2, b) in zip(renderer.system.state_dict().items(), system.state_dict().items()):
        assert k == k2 and torch.equal(a, b), "checkpoint round trip changed %s" % k
    focal, poses, box = synth.edit_demo_geometry(synth.SCANNET_LIKE, EDIT_HW[1])
    import utils.bbox_utils as BU
    helper = object.__new__(BU.BBoxRayHelper)                          # the constructor reads dataset files (bbox_utils.py:10-34)
    helper.dataset_name = "toydesk"                                     # read_bbox_info_desk's conventions (bbox_utils.py:67-98)
    helper.scale_factor = box["scale_factor"]
    helper.pose_avg = np.eye(4); helper.pose_avg[:3, :3] = box["R_avg"]; helper.pose_avg[:3, 3] = box["t_avg"]
    helper.axis_align_mat = np.eye(4); helper.axis_align_mat[:3, :3] = box["R_box"]; helper.axis_align_mat[:3, 3] = box["t_box"]
    helper.bbox_bounds = np.array([np.asarray(box["bmin"], dtype=np.float64), np.asarray(box["bmax"], dtype=np.float64)])
    renderer.object_bbox_ray_helpers["4"] = helper                     # what initialize_object_bbox(4) would store (306-309)
    renderer.object_to_remove = [4]                                    # remove_scene_object_by_ids([4]) minus the file read
    renderer.bbox_enlarge = 0.06
    c10, s10 = np.cos(np.radians(10.0)), np.sin(np.radians(10.0))
    for dup, (dx, dy, yaw_s) in enumerate([(0.4, 1.6, s10), (-2.0, 1.2, -s10)]):     # test/demo_editable_render.py:33-42 shape
        pose = np.eye(4)
        pose[:3, :3] = [[c10, -yaw_s, 0], [yaw_s, c10, 0], [0, 0, 1]]
        pose[:3, 3] = [dx, dy, 0.1 * (dup + 1)]
        renderer.set_object_pose_transform(4, pose, dup)
    # a camera pose in WORLD units looking at the box (render_edit re-centres and rescales it: 215, 254-255)
    Twc = np.eye(4)
    Twc[:3] = poses[0]
    Twc[:3, 3] = Twc[:3, 3] * synth.SCANNET_LIKE["scale_factor"] + np.asarray(synth.SCANNET_LIKE["scene_center"])
    scenario("render_edit")
    r = renderer.render_edit(EDIT_HW[0], EDIT_HW[1], Twc[:3].copy(), fovx_deg=60.0, show_progress=False)
    for k, v in r.items():
        out["edit_" + k] = v.detach().clone()
    scenario("render_origin")
    r = renderer.render_origin(EDIT_HW[0], EDIT_HW[1], Twc[:3].copy(), fovx_deg=60.0)
    for k, v in r.items():
        out["origin_" + k] = v.detach().clone()
    return out


# the committed checkpoint fixture (row f3): a 300-point cloud (6,025 occupied voxels of the 62 x 62 x 27 grid) and a
# 6,500-row table keep the file at 8 MB -- 7.1 MB of it are the two MLPs, whose architecture the kernels fix
CKPT_POINTS, CKPT_MAX_VOXELS = 300, 6500


def _box_helper(box):
    """a reference BBoxRayHelper for a synth.oriented_box dict (its constructor reads dataset files, bbox_utils.py:10-34)"""
    import utils.bbox_utils as BU
    helper = object.__new__(BU.BBoxRayHelper)
    helper.scale_factor = box["scale_factor"]
    helper.pose_avg = np.eye(4); helper.pose_avg[:3, :3] = box["R_avg"]; helper.pose_avg[:3, 3] = box["t_avg"]
    helper.axis_align_mat = np.eye(4); helper.axis_align_mat[:3, :3] = box["R_box"]; helper.axis_align_mat[:3, 3] = box["t_box"]
    helper.bbox_bounds = np.array([np.asarray(box["bmin"], dtype=np.float64), np.asarray(box["bmax"], dtype=np.float64)])
    return helper


def write_checkpoint_fixture(work):
    """tests/golden/reference_small.ckpt + ckpt_render.npz: the REAL train.py::ObjectNeRFSystem built from the REFERENCE's
    module types, filled with the seeded W1 weights, saved the way Lightning saves it ({"state_dict": ...}), and what the
    reference renders from it: render_rays on cases.render_inputs("voxel_eval") (64 + 64, eval) and render_rays_multi on
    cases.multi_inputs() (ids [0, 4, 4], removed-object box) -- editable_renderer.py:75-79 then 125-140 / 272-287."""
    from oracle import ref_import
    import cases
    import train as T
    import render_tools.multi_rendering as MR
    assert T.__file__.startswith(REF) and T.render_rays.__module__ == "models.rendering"
    cfg = make_config(work, CKPT_POINTS, CKPT_MAX_VOXELS)
    system = T.ObjectNeRFSystem(cfg)
    assert type(system.nerf_coarse).__module__ == "models.nerf_model"
    fill_system(system)
    system.eval()
    torch.save({"state_dict": system.state_dict(), "epoch": 0, "global_step": 0}, os.path.join(work, "reference_small.ckpt"))
    out = {}
    with torch.no_grad():
        rays, ids, _, _ = cases.render_inputs("voxel_eval")
        codes = system.code_library({"instance_ids": ids})["embedding_instance"]
        r = T.render_rays(system.models, system.embeddings, rays, N_samples=64, N_importance=64, perturb=0, noise_std=0,
                          chunk=32768, embedding_instance=codes, is_eval=True)
        out.update({"single_" + k: v for k, v in r.items()})
        sets, boxes = cases.multi_inputs()
        m = cases.MULTI
        r = MR.render_rays_multi(system.models, system.embeddings, system.code_library, [s.clone() for s in sets], m["obj_ids"],
                                 N_samples=m["N_samples"], N_importance=m["N_importance"], perturb=0, noise_std=0, chunk=32768,
                                 white_back=False, background_skip_bbox={4: _box_helper(boxes[0])})
        out.update({"multi_" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(work, "ckpt_render.npz"), **{k: v.numpy() for k, v in out.items()})
    print("checkpoint flavour: %d tensors in the state_dict, %d rendered arrays" % (len(system.state_dict()), len(out)))


def main():
    flavour, work = sys.argv[1], sys.argv[2]
    os.makedirs(work, exist_ok=True)
    torch.set_num_threads(8)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    if flavour in ("reference", "reference-load", "checkpoint"):
        sys.path.insert(0, REF)
    else:
        sys.path.insert(0, REF)
        sys.path.insert(0, os.path.join(ROOT, "dropin"))
    install_caller_stubs()
    if flavour == "dropin":
        # the REAL dropin/datasets/__init__.py runs (package-path extension + the reference's own datasets/__init__.py
        # with its cv2 / torchvision imports stubbed above) instead of the stub package of the reference flavours
        del sys.modules["datasets"]
        import datasets
        assert datasets.__file__.startswith(os.path.join(ROOT, "dropin")) and "scannet_base" in datasets.dataset_dict
    if flavour == "checkpoint":
        write_checkpoint_fixture(work)
    elif flavour == "reference":
        out = run_scenarios("reference", work, None)
        np.savez_compressed(os.path.join(work, "callers.npz"), **{k: v.numpy() for k, v in out.items()})
        print("reference flavour: %d arrays" % len(out))
    elif flavour == "dropin":
        rec = Recorder()
        out = run_scenarios("dropin", work, rec)
        z = np.load(os.path.join(work, "callers.npz"))
        assert sorted(z.files) == sorted(out), (sorted(set(z.files) ^ set(out)))
        for k in z.files:
            a, b = out[k], torch.from_numpy(z[k])
            assert a.dtype == b.dtype and a.shape == b.shape, k
            if k.endswith("obj_ids_coarse"):     # order of exactly tied depths (rays that missed their box: z = 0) is unspecified
                keep = torch.from_numpy(z[k.replace("obj_ids", "z_vals")]) != 0
                a, b = a[keep], b[keep]
            assert torch.equal(a, b), "dropin flavour differs from the reference's at %s" % k
        save_calls(os.path.join(work, "calls.npz"), rec.calls)
        print("dropin flavour: %d arrays equal to the reference flavour's, %d calls recorded" % (len(out), len(rec.calls)))
    elif flavour == "reference-load":
        # f3, the other direction: the reference's own module types load the product's exported checkpoint strictly
        import train as T
        system = T.ObjectNeRFSystem.load_from_checkpoint(os.path.join(work, "dropin.ckpt"), config=make_config(work))
        ref = torch.load(os.path.join(work, "reference.ckpt"), weights_only=False)["state_dict"]
        for k, v in system.state_dict().items():
            assert torch.equal(v, ref[k]), k
        print("reference-load: %d tensors equal" % len(ref))
    else:
        raise SystemExit("usage: ref_callers.py reference|dropin|reference-load|checkpoint <workdir>")


if __name__ == "__main__":
    main()
