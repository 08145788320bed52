#!/usr/bin/env python
"""Generates tests/golden/*.npz by running the REAL reference (/root/reference, imported through
oracle/ref_import.py) on the deterministic cases of tests/cases.py.

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py
The committed vectors pin both the oracle (tests/test_oracle_vs_golden.py, CPU) and the HIP path
(tests/test_gpu_*.py).  The reference itself ships no golden vectors or tests for this path
(SURVEY.md §4), so these are outputs of the reference code itself, fp32, torch 2.10 CPU kernels.
"""
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_import  # noqa: E402
import cases  # noqa: E402
from object_nerf_amd import synth  # noqa: E402
from object_nerf_amd.config import AttrDict  # noqa: E402


def ref_types(ref):
    def mk_ev(ch, nf, mv, conf):
        c = AttrDict(conf)
        key = "synthetic_%d.ply" % len(ref_import.POINT_CLOUDS)
        ref_import.POINT_CLOUDS[key] = np.asarray(conf["pcd_xyz"])
        c["pcd_path"] = key
        return ref.EmbeddingVoxel(ch, nf, mv, c)
    return types.SimpleNamespace(ObjectNeRF=ref.ObjectNeRF, Embedding=ref.Embedding, EmbeddingVoxel=mk_ev,
                                 CodeLibrary=ref.CodeLibrary)


def save(name, d):
    os.makedirs(cases.GOLDEN_DIR, exist_ok=True)
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}
    np.savez_compressed(os.path.join(cases.GOLDEN_DIR, name + ".npz"), **arrs)
    print("wrote %-28s %s" % (name, {k: tuple(a.shape) for k, a in list(arrs.items())[:4]}))


def _f64_oracle(fn, *a, **kw):
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            return fn(*a, **kw)
    finally:
        torch.set_default_dtype(old)


def _dbl(d):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}


def frames(ref, scene):
    """Image-scale goldens (cases.FRAME_CASES): one 160x120 frame of each bench workload rendered by the real reference
    (utils/metrics.py:5-15 defines PSNR over a frame).  Stored per case: the pixel maps of both passes, the fine depths of
    every sub-th ray (moved-ray count on the GPU), `_floor_<key>` = max-norm distance between the reference's fp32 maps
    and the float64 oracle on the same inputs (the reference's own fp32 noise floor), `_floor_l2_<key>` the same in
    relative L2, `_moved64` = the float64 oracle's moved-ray count on the stored subset."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from oracle import objnerf_oracle as O
    for case, fc in cases.FRAME_CASES.items():
        out = {}
        if fc["kind"] == "single":
            rays, ids, kw, sname = cases.frame_inputs(case)
            sc = scene(sname)
            codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
            r = dict(ref.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, chunk=32768, **kw))
            okw = {k: v for k, v in kw.items()}
            f64 = _f64_oracle(O.render_rays, _dbl(H.state(sc.models["coarse"])), _dbl(H.state(sc.models["fine"])),
                              _dbl(H.oracle_grid(sc.embeddings["xyz"])), rays.double(), embedding_instance=codes.double(), **okw)
            maps = cases.FRAME_MAPS
            sub = torch.arange(0, rays.shape[0], cases.FRAME["sub"])
            zc_sub = r["z_vals_coarse"][sub]
        else:
            sname = cases.frame_inputs(case)[3]
            sc = scene(sname)
            bm = cases.BENCH_MULTI
            focal, poses, box, _ = cases.frame_inputs(case)
            w_, h_ = cases.FRAME["W"], cases.FRAME["H"]
            directions = ref.get_ray_directions(h_, w_, focal)
            helper = ref_import.make_box(box)
            sets = []
            for k, Toc in enumerate(poses):
                rays_o, rays_d = ref.get_rays(directions, torch.from_numpy(np.asarray(Toc)).float())
                if k == 0:
                    pre = synth.SCANNET_LIKE
                    nf = [pre["near"] * torch.ones_like(rays_o[:, :1]), pre["far"] * torch.ones_like(rays_o[:, :1])]
                else:
                    mask, bn, bf = helper.get_ray_bbox_intersections(rays_o, rays_d, box["scale_factor"], bbox_enlarge=bm["bbox_enlarge"])
                    bn[~mask] = torch.zeros_like(bn[~mask])
                    bf[~mask] = torch.zeros_like(bf[~mask])
                    nf = [bn, bf]
                    out["_hit_%d" % k] = np.packbits(mask.numpy())
                sets.append(torch.cat([rays_o, rays_d] + nf, 1))
            # the tests regenerate these ray sets with the oracle's generate_rays: it has to be bit-equal to the reference's
            for a, b in zip(sets, cases.frame_multi_sets(O.generate_rays, case)):
                assert torch.equal(a, b), "oracle generate_rays differs from the reference's ray / box code"
            # The sets themselves travel with the golden (round 6): the reference's get_rays is an fp32 matmul + a vector norm,
            # which round differently on different host CPUs -- sets regenerated on the GPU box are NOT the rays this frame was
            # rendered from (directions an ulp apart on some rays).  Directions + (near, far) per set; origins are one point each.
            for k, st in enumerate(sets):
                out["_set%d_o" % k] = st[0, 0:3].clone()
                assert torch.equal(st[:, 0:3], st[0:1, 0:3].expand(st.shape[0], 3))
                out["_set%d_d" % k] = st[:, 3:6].clone()
                out["_set%d_nf" % k] = st[:, 6:8].clone()
            r = dict(ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets], bm["obj_ids"],
                                           N_samples=bm["N_samples"], N_importance=bm["N_importance"], perturb=0, noise_std=0,
                                           chunk=32768, white_back=False, background_skip_bbox={4: helper}))
            for typ in ("coarse", "fine"):
                zz = r["z_vals_" + typ]
                out["_tied_" + typ] = ((zz[:, 1:] == zz[:, :-1]) & (zz[:, 1:] != 0)).any(1).sum()
            f64 = _f64_oracle(O.render_rays_multi, _dbl(H.state(sc.models["coarse"])), _dbl(H.state(sc.models["fine"])),
                              _dbl(H.oracle_grid(sc.embeddings["xyz"])), sc.code_library.embedding_instance.weight.detach().double(),
                              [s.double() for s in sets], bm["obj_ids"], N_samples=bm["N_samples"],
                              N_importance=bm["N_importance"], skip_boxes=[box])
            maps = ["rgb", "depth", "opacity"]
            sub = torch.arange(0, sets[0].shape[0], cases.FRAME["sub_multi"])
            # the joint coarse depths interleave the K sets: the spacing a moved sample is measured against is a set's own
            # (the background's; helpers.frame_report does the same)
            nf = sets[0][sub, 6:8]
            zc_sub = nf[:, :1] + (nf[:, 1:] - nf[:, :1]) * torch.linspace(0, 1, bm["N_samples"])
        for typ in ("coarse", "fine"):
            for m in maps:
                k = "%s_%s" % (m, typ)
                out[k] = r[k]
                out["_floor_" + k] = H.normwise(r[k], f64[k])
                out["_floor_l2_" + k] = H.rel_l2(r[k], f64[k])
        out["_sub"] = sub
        out["_z_vals_fine_sub"] = r["z_vals_fine"][sub]
        out["_moved64"] = int(H.moved_rays(f64["z_vals_fine"][sub], r["z_vals_fine"][sub], zc_sub).sum())
        out["_psnr64"] = -10.0 * math.log10(max(((r["rgb_fine"].double() - f64["rgb_fine"]) ** 2).mean().item(), 1e-30))
        print(case, "floors:", {k[7:]: "%.1e" % float(v) for k, v in out.items() if k.startswith("_floor_") and "l2" not in k},
              "moved64", out["_moved64"], "of", len(sub), "psnr64 %.1f" % out["_psnr64"])
        save(case, out)


def full_frame(ref, scene):
    """BASELINE configs[1] at 640x480: all 307,200 rays of bench.py's ToyDesk-2 frame through the real reference (minutes of
    CPU).  cases.FULL_FRAME describes what is kept."""
    rays, ids, kw, sname = cases.full_frame_inputs()
    sc = scene(sname)
    out = {}
    parts = {"rgb_fine": [], "depth_fine": []}
    step = 32768
    for lo in range(0, rays.shape[0], step):          # the callers' own ray-chunk loop (train.py:84-98)
        r_ = rays[lo:lo + step]
        codes = sc.code_library({"instance_ids": ids[lo:lo + step]})["embedding_instance"]
        r = ref.render_rays(sc.models, sc.embeddings, r_, embedding_instance=codes, chunk=32768, **kw)
        for k in parts:
            parts[k].append(r[k])
        print("full frame: %d / %d rays" % (min(lo + step, rays.shape[0]), rays.shape[0]), flush=True)
    rgb = torch.cat(parts["rgb_fine"], 0)
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    out["rgb_fine_u16"] = torch.round(rgb.double() * 65535.0).to(torch.int32).numpy().astype(np.uint16)
    out["depth_fine_sub"] = torch.cat(parts["depth_fine"], 0)[:: cases.FULL_FRAME["sub"]]
    out["_mean_rgb"] = rgb.double().mean()
    save("full_frame_toydesk2", out)


def coarse_f64(ref, scene):
    """Coarse-pass attribution vectors (round 6, tools/frame_parity.py --coarse): for the two single-ray-set 160x120 frames
    the float64 oracle's coarse maps themselves (frame_*.npz holds the reference's maps and only the DISTANCE to float64),
    and for the 640x480 configs[1] frame the coarse maps of every FULL_FRAME["sub"]-th pixel from the real reference (the
    coarse pass does not depend on N_importance: rendered with N_importance=0) next to the float64 oracle's.  With both,
    dist(ours, float64) can be set beside dist(reference, float64): which of the two fp32 computations is the noisier one."""
    import helpers as H
    from oracle import objnerf_oracle as O
    out = {}
    todo = [(c, cases.frame_inputs(c), None) for c in ("frame_toydesk2", "frame_scannet_multi")]
    todo.append(("full_frame_toydesk2_sub", cases.full_frame_inputs(), cases.FULL_FRAME["sub"]))
    for name, (rays, ids, kw, sname), sub in todo:
        if sub:
            rays, ids = rays[::sub].contiguous(), ids[::sub].contiguous()
        sc = scene(sname)
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        kw = dict(kw, N_importance=0)
        r = dict(ref.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, chunk=32768, **kw))
        f64 = _f64_oracle(O.render_rays, _dbl(H.state(sc.models["coarse"])), _dbl(H.state(sc.models["fine"])),
                          _dbl(H.oracle_grid(sc.embeddings["xyz"])), rays.double(), embedding_instance=codes.double(), **kw)
        for m in cases.FRAME_MAPS:
            k = m + "_coarse"
            out["%s__%s" % (name, k)] = r[k]
            out["%s__%s_f64" % (name, k)] = f64[k]
            print(name, k, "reference vs float64: max-norm %.2e, rel L2 %.2e" % (H.normwise(r[k], f64[k]), H.rel_l2(r[k], f64[k])))
        if not sub:        # the coarse maps of a full render (frame_*.npz) are the maps of this coarse-only render
            g = cases.load_golden(name)
            for m in cases.FRAME_MAPS:
                assert torch.equal(g[m + "_coarse"], r[m + "_coarse"]), (name, m)
    save("coarse_f64", out)


def multi_training_mode(ref, scene):
    """render_rays_multi in training mode (perturb != 0, noise_std != 0; multi_rendering.py:186-190, 126, 272-274) with the
    draws of cases.multi_randoms() injected in call order: randn_like (coarse compositing), rand x K (sample_pdf per set),
    randn_like (fine compositing)."""
    sc = scene("voxel")
    sets, boxes = cases.multi_inputs()
    m = cases.MULTI
    rnd = cases.multi_randoms()
    with ref_import.inject_randoms(rand=list(rnd["u_rand"]), randn_like=list(rnd["noise"])):
        out = ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets], m["obj_ids"],
                                    N_samples=m["N_samples"], N_importance=m["N_importance"], perturb=1.0, noise_std=1.0,
                                    chunk=32768, white_back=False, background_skip_bbox={4: ref_import.make_box(boxes[0])})
    assert not torch.equal(out["z_vals_fine"], cases.load_golden("multi_scannet_dup")["z_vals_fine"])
    save("multi_train_random", dict(out))


def other_architectures(ref, scene):
    """config.model shapes other than the shipped default (cases.ARCH_SCENES) through the REAL reference: render_rays in
    eval mode and with the training-time flags, render_rays_multi, and the two MLP forwards on materialised embeddings."""
    for name in cases.ARCH_SCENES:
        sc = scene(name)
        rays, ids, ptm, sets, boxes = cases.arch_inputs(name)
        ar = cases.ARCH_RENDER
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        out = {}
        r = ref.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, chunk=32768, N_samples=ar["N_samples"],
                            N_importance=ar["N_importance"], perturb=0, noise_std=0, is_eval=True)
        out.update({"eval_" + k: v for k, v in r.items()})
        r = ref.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, chunk=32768, N_samples=ar["N_samples"],
                            N_importance=ar["N_importance"], perturb=0, noise_std=0, is_eval=False, frustum_bound_th=0.025,
                            rays_in_bbox=True, white_back=True, pass_through_mask=ptm)
        out.update({"flags_" + k: v for k, v in r.items()})
        if cases.ARCH_SCENES[name][0]:      # the reference's render_rays_multi unpacks (scene, object) features: voxel mode only
            r = ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets], cases.MULTI["obj_ids"],
                                      N_samples=ar["N_samples"], N_importance=ar["N_importance"], perturb=0, noise_std=0, chunk=32768,
                                      white_back=False, background_skip_bbox={4: ref_import.make_box(boxes[0])})
            out.update({"multi_" + k: v for k, v in r.items()})
        # the forwards on the embeddings of 150 points (ObjectNeRF.forward / forward_instance, sigma_only too)
        pts = cases.voxel_points(150)
        e = sc.embeddings["xyz"](pts.clone())
        ex, ov = e if isinstance(e, tuple) else (e, None)
        ed = sc.embeddings["dir"](torch.nn.functional.normalize(pts.flip(-1), dim=-1))
        m = sc.models["fine"]
        code = sc.code_library.embedding_instance(torch.full((pts.shape[0],), 3, dtype=torch.long))
        o = m({"emb_xyz": ex, "emb_dir": ed})
        oi = m.forward_instance({"emb_xyz": ex, "emb_dir": ed, "obj_voxel": ov, "obj_code": code})
        out.update(fwd_emb_xyz=ex, fwd_emb_dir=ed, fwd_sigma=o["sigma"], fwd_rgb=o["rgb"], fwd_inst_sigma=oi["inst_sigma"],
                   fwd_inst_rgb=oi["inst_rgb"])
        if ov is not None:
            out["fwd_obj_voxel"] = ov
        save(name, out)


def sigma_grids(ref, scene):
    """tools/extract_mesh.py:62-113 issued on a 32^3 lattice with the REAL reference modules (the script itself is a
    __main__ with a checkpoint and mcubes; its query loop is the part restated here, line by line): nerf_fine's density
    for the scene (obj_id 0) and for object 4, voxel and plain embedding."""
    x, y, z = cases.sigma_grid_axes()
    xyz_ = torch.FloatTensor(np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3))                  # :66
    chunk = 32768                                                                                # default_conf.yml:41
    out = {}
    for sname in ("voxel", "plain"):
        sc = scene(sname)
        emb, fine, lib = sc.embeddings["xyz"], sc.models["fine"], sc.code_library
        use_voxel = sname == "voxel"
        for obj_id in (0, cases.SIGMA_GRID["obj_id"]):
            chunks = []
            for i in range(0, xyz_.shape[0], chunk):                                             # :80
                obj_voxel_embedded = None
                if use_voxel:
                    xyz_embedded, obj_voxel_embedded = emb(xyz_[i:i + chunk])                    # :84-87
                else:
                    xyz_embedded = emb(xyz_[i:i + chunk])                                        # :88-91
                input_dict = {"emb_xyz": xyz_embedded, "obj_voxel": obj_voxel_embedded}
                if obj_id > 0:
                    n_local = xyz_embedded.shape[0]
                    input_dict["obj_code"] = lib.embedding_instance(torch.ones((n_local)).long() * obj_id)    # :99-101
                    chunks.append(fine.forward_instance(input_dict, sigma_only=True)["inst_sigma"])      # :102-104
                else:
                    chunks.append(fine.forward(input_dict, sigma_only=True)["sigma"])                    # :106-108
            sigma = torch.cat(chunks, 0)                                                         # :111
            assert sigma.shape == (xyz_.shape[0], 1)
            out["%s_obj%d" % (sname, obj_id)] = sigma[:, -1]                                     # :113
    save("stage_sigma_grid", out)


def write_input_digests():
    """tests/golden/input_digests.json: bit digests of every synthetic input the goldens were made from, as generated HERE.
    The tests regenerate the inputs on their own host and compare (tests/test_golden_inputs.py, and on the GPU box
    tests/test_gpu_frames.py): a golden is only a golden for the inputs it was rendered from."""
    import json
    d = cases.input_digests()
    path = os.path.join(cases.GOLDEN_DIR, "input_digests.json")
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, "(%d digests)" % len(d))


def main():
    torch.set_num_threads(8)
    ref = ref_import.load_reference()
    rt = ref_types(ref)
    scenes = {}

    def scene(name):
        if name not in scenes:
            scenes[name] = cases.scene_for(rt, name)
        return scenes[name]

    if "--frames" in sys.argv:       # only the image-scale cases (minutes of CPU: the other files are left untouched)
        with torch.no_grad():
            frames(ref, scene)
        return
    if "--sigma-grid" in sys.argv:   # only the density-grid query
        with torch.no_grad():
            sigma_grids(ref, scene)
        return
    if "--arch" in sys.argv:         # only the non-default architectures
        with torch.no_grad():
            other_architectures(ref, scene)
        return
    if "--multi-train" in sys.argv:
        with torch.no_grad():
            multi_training_mode(ref, scene)
        other_architectures(ref, scene)
        return
    if "--digests" in sys.argv:     # only tests/golden/input_digests.json
        write_input_digests()
        return
    if "--coarse-f64" in sys.argv:  # only the coarse-pass attribution vectors (a minute of CPU)
        with torch.no_grad():
            coarse_f64(ref, scene)
        return
    if "--full-frame" in sys.argv:   # only the 640x480 frame (about five minutes of CPU)
        with torch.no_grad():
            full_frame(ref, scene)
        return

    with torch.no_grad():
        frames(ref, scene)
        full_frame(ref, scene)
        coarse_f64(ref, scene)
        sigma_grids(ref, scene)
        # ---- render_rays end to end ----
        for case, c in cases.RENDER_CASES.items():
            sc = scene(c["scene"])
            rays, ids, ptm, randoms = cases.render_inputs(case)
            codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
            kw = dict(c["kw"])
            kw.setdefault("perturb", 0)
            kw.setdefault("noise_std", 0)
            if ptm is not None:
                kw["pass_through_mask"] = ptm
            if randoms is None:
                out = ref.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, chunk=32768, **kw)
            else:
                fi = kw.get("forward_instance", True)
                nz = randoms["noise"]
                with ref_import.inject_randoms(rand_like=[randoms["perturb_rand"]], rand=[randoms["u_rand"]],
                                               randn_like=[nz[0], nz[1], nz[2], nz[3]] if fi else [nz[0], nz[2]]):
                    out = ref.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, chunk=32768, **kw)
            out = dict(out)
            out["_rays"] = rays
            out["_codes"] = codes
            save("render_" + case, out)

        # ---- stages ----
        pe = {}
        for k, (x, nf) in cases.pe_inputs().items():
            pe[k] = ref.Embedding(x.shape[1], nf)(x)
        save("stage_pe", pe)

        pts = cases.voxel_points()
        for sname in ("voxel", "sparse"):
            ev = scene(sname).embeddings["xyz"]
            s_ftr, o_ftr = ev(pts.clone())
            save("stage_voxel_embed_" + sname, dict(scene_ftr=s_ftr, obj_ftr=o_ftr, idx_map_sum=ev.voxel_idx_map.sum(),
                                                    occupied=ev.voxel_occupancy.sum(), shape=ev.voxel_shape))

        for sname in ("voxel", "plain"):
            m = scene(sname).models["coarse"]
            inp = cases.mlp_inputs(sname == "voxel")
            o = m({"emb_xyz": inp["emb_xyz"], "emb_dir": inp["emb_dir"]})
            oi = m.forward_instance(dict(inp))
            so = m({"emb_xyz": inp["emb_xyz"], "emb_dir": inp["emb_dir"]}, sigma_only=True)
            assert list(so.keys()) == ["sigma"]
            save("stage_mlp_" + sname, dict(sigma=o["sigma"], rgb=o["rgb"], inst_sigma=oi["inst_sigma"], inst_rgb=oi["inst_rgb"]))

        bins, w, u = cases.pdf_inputs()
        det = ref.sample_pdf(bins, w, 64, det=True)
        with ref_import.inject_randoms(rand=[u]):
            rnd = ref.sample_pdf(bins, w, u.shape[1], det=False)
        save("stage_sample_pdf", dict(det=det, rnd=rnd))

        # ---- render_rays_multi ----
        sc = scene("voxel")
        sets, boxes = cases.multi_inputs()
        ref_boxes = {4: ref_import.make_box(boxes[0])}
        m = cases.MULTI
        out = ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets], m["obj_ids"],
                                    N_samples=m["N_samples"], N_importance=m["N_importance"], perturb=0, noise_std=0,
                                    chunk=32768, white_back=False, background_skip_bbox=ref_boxes)
        save("multi_scannet_dup", dict(out))
        out = ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets], m["obj_ids"],
                                    N_samples=m["N_samples"], N_importance=0, perturb=0, noise_std=0,
                                    chunk=32768, white_back=True, background_skip_bbox=None)
        save("multi_coarse_only_white", dict(out))
        # the 10-column ray sets (fine depths clipped to the far end of a per-ray interval, multi_rendering.py:277-285)
        sets10, _ = cases.multi_inputs_clip()
        out = ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets10], m["obj_ids"],
                                    N_samples=m["N_samples"], N_importance=m["N_importance"], perturb=0, noise_std=0,
                                    chunk=32768, white_back=False, background_skip_bbox=ref_boxes)
        assert not torch.equal(out["z_vals_fine"], cases.load_golden("multi_scannet_dup")["z_vals_fine"])    # the clip bites
        save("multi_scannet_clip10", dict(out))
        multi_training_mode(ref, scene)
        # ---- bench.py --config 4: the editing demo's ray sets, generated by the reference's own ray / box code ----
        sc = scene("scannet_800k")
        bm = cases.BENCH_MULTI
        focal, poses, box = cases.bench_multi_geometry()
        pix = cases.bench_multi_pixels()
        w_, h_ = bm["frame"]
        directions = ref.get_ray_directions(h_, w_, focal)
        helper = ref_import.make_box(box)
        sets = []
        for k, Toc in enumerate(poses):
            rays_o, rays_d = ref.get_rays(directions, torch.from_numpy(np.asarray(Toc)).float())
            rays_o, rays_d = rays_o[pix], rays_d[pix]
            if k == 0:
                pre = synth.SCANNET_LIKE
                nf = [pre["near"] * torch.ones_like(rays_o[:, :1]), pre["far"] * torch.ones_like(rays_o[:, :1])]
            else:
                mask, bn, bf = helper.get_ray_bbox_intersections(rays_o, rays_d, box["scale_factor"], bbox_enlarge=bm["bbox_enlarge"])
                bn[~mask] = torch.zeros_like(bn[~mask])
                bf[~mask] = torch.zeros_like(bf[~mask])
                nf = [bn, bf]
                assert 0 < int(mask.sum()) < mask.numel(), "config-4 golden: object set %d hits %d of %d" % (k, int(mask.sum()), mask.numel())
            sets.append(torch.cat([rays_o, rays_d] + nf, 1))
        out = ref.render_rays_multi(sc.models, sc.embeddings, sc.code_library, [s.clone() for s in sets], bm["obj_ids"],
                                    N_samples=bm["N_samples"], N_importance=bm["N_importance"], perturb=0, noise_std=0,
                                    chunk=32768, white_back=False, background_skip_bbox={4: helper})
        out = dict(out)
        for typ in ("coarse", "fine"):     # no exact cross-set depth ties away from z == 0 (their order is unspecified in the reference)
            zz = out["z_vals_" + typ]
            assert not ((zz[:, 1:] == zz[:, :-1]) & (zz[:, 1:] != 0)).any(), "config-4 golden: tied depths in the %s pass" % typ
        for k, s_ in enumerate(sets):
            out["_rays_%d" % k] = s_
        save("multi_bench_edit_demo", out)

        # ---- editor ray generation (row f2) ----
        h, w, focal, Toc, box = cases.raygen_inputs()
        rg = cases.RAYGEN
        directions = ref.get_ray_directions(h, w, focal)
        rays_o, rays_d = ref.get_rays(directions, Toc)
        bg = torch.cat([rays_o, rays_d, rg["near"] * torch.ones_like(rays_o[:, :1]), rg["far"] * torch.ones_like(rays_o[:, :1])], 1)
        helper = ref_import.make_box(box)
        mask, bn, bf = helper.get_ray_bbox_intersections(rays_o, rays_d, box["scale_factor"], bbox_enlarge=rg["bbox_enlarge"])
        bn[~mask] = torch.zeros_like(bn[~mask])          # editable_renderer.py:175-176
        bf[~mask] = torch.zeros_like(bf[~mask])
        obj = torch.cat([rays_o, rays_d, bn, bf], 1)
        assert 0 < int(mask.sum()) < mask.numel()
        save("stage_generate_rays", dict(background=bg, object=obj, hit=mask))

        xyz = cases.voxel_points(600).view(20, 30, 3)
        save("stage_points_in_boxes", dict(inside=ref.check_in_any_boxes(ref_boxes, xyz)))
    write_input_digests()


if __name__ == "__main__":
    main()
