"""GPU: image-scale parity.  A whole 160x120 frame of each bench workload (BASELINE configs 1, 2 and 4: same scenes,
weights, camera, code assignment as bench.py) rendered through the drop-in API, against the SAME frame rendered by the real
reference (tests/golden/frame_*.npz, oracle/make_golden.py::frames).  utils/metrics.py:5-15 defines PSNR as a mean over a
frame, so this -- not the 48-ray cases of test_gpu_render.py -- is where the BASELINE's "PSNR within 0.1 dB" is graded.

Per case:
  * PSNR(ours, reference) of rgb_fine over the frame >= 60 dB;
  * |PSNR(ours, T) - PSNR(reference, T)| <= 0.1 dB against a fixed synthetic target T;
  * every FINE-pass pixel map (rgb / depth / opacity + the three instance maps): max-norm distance from the reference's
    map <= 3x the distance between the reference's fp32 frame and the float64 oracle on the same inputs (its own fp32 noise
    floor, stored with the golden); the same in relative L2 (robust against a single moved ray);
  * every COARSE-pass map (round 6, no floor allowance): <= 2e-5 on the single-ray-set frames, <= 1x the floor on the multi-set frame;
  * rays whose importance samples moved (helpers.moved_rays, on the stored subset of rays) <= the float64 oracle's count
    + 0.1 % of the subset (at least 1)."""
import numpy as np
import pytest
import torch

import cases
import helpers as H
import object_nerf_amd as A

pytestmark = pytest.mark.gpu
DEV = "cuda"
_scenes = {}


def scene(name):
    if name not in _scenes:
        _scenes[name] = cases.scene_for(A, name, device=DEV)
    return _scenes[name]


@pytest.mark.parametrize("case", sorted(cases.FRAME_CASES))
def test_frame_matches_reference_frame(case):
    g = cases.load_golden(case)
    out = H.render_frame_case(case, scene(cases.frame_inputs(case)[3]))
    rep = H.frame_report(case, out, g)
    print(case, "PSNR %.1f dB (reference vs float64: %.1f), dPSNR %.4f, moved %d (float64: %d) of %d" % (
        rep["psnr"], rep["psnr64"], rep["dpsnr"], rep["moved"], rep["moved64"], rep["n_sub"]))
    assert rep["psnr"] >= 60.0
    assert rep["dpsnr"] <= 0.1
    single = cases.FRAME_CASES[case]["kind"] == "single"
    for k, (err, floor, e2, f2) in rep["rows"].items():
        if k.endswith("coarse"):
            # Round 6: NO floor allowance for the coarse pass.  Rounds 3-5 allowed max(3 x floor, 1e-4) because these keys sat at
            # 1.2-1.6e-4 (1.13 x the floor on the headline config) -- which was the GPU box's host regenerating ray directions an ulp
            # off the ones the reference had rendered (tests/test_golden_inputs.py), not the kernels.  On identical rays the
            # single-ray-set frames measure 2.6-3.8e-6 (profiles/r06_frame_parity.md): held to 2e-5, i.e. 5 x inside the
            # contract's 1e-4 and 8 x inside the reference's own fp32-vs-fp64 distance.  The multi-set frame's coarse maps
            # inherit the joint depth sort's discontinuities (the reference itself is 7e-3 from its float64 run there): never
            # further from the reference than the reference is from exact arithmetic (1 x floor, measured 0.10-0.13 x).
            tol = 2e-5 if single else floor
            assert err <= tol and e2 <= (2e-5 if single else f2), "%s/%s: max-norm %.3e, relative L2 %.3e (floors %.3e / %.3e)" % (
                case, k, err, e2, floor, f2)
            continue
        assert err <= max(H.FLOOR_FACTOR * floor, 2e-5), "%s/%s: max-norm %.3e, floor %.3e" % (case, k, err, floor)
        assert e2 <= max(H.FLOOR_FACTOR * f2, 2e-5), "%s/%s: relative L2 %.3e, floor %.3e" % (case, k, e2, f2)
    assert rep["moved"] <= rep["moved64"] + max(1, rep["n_sub"] // 1000), (rep["moved"], rep["moved64"])


def test_inputs_regenerate_bit_identically_on_this_host():
    """the GPU box's host CPU regenerates every synthetic input bit-for-bit as the host that made the goldens did
    (tests/test_golden_inputs.py explains; the same check, run HERE because this is the host that differed in rounds 3-5)"""
    from test_golden_inputs import check_input_digests
    check_input_digests()


def test_full_size_frame_matches_the_reference_frame():
    """BASELINE configs[1] at its OWN size: the 640x480 ToyDesk-2 frame bench.py times (same scene, weights, camera, code)
    against the same 307,200 pixels rendered by the real reference (tests/golden/full_frame_toydesk2.npz: rgb_fine as
    uint16 / 65535 -- quantisation 107 dB --, depth_fine of every 16th pixel in fp32).  "PSNR within 0.1 dB of the
    reference" (BASELINE north_star; utils/metrics.py:5-15), literally, at 640x480."""
    g = cases.load_golden("full_frame_toydesk2")
    rays, ids, kw, sname = cases.full_frame_inputs()
    sc = scene(sname)
    with torch.no_grad():
        codes = sc.code_library({"instance_ids": ids.to(DEV)})["embedding_instance"]
        out = A.render_rays(sc.models, sc.embeddings, rays.to(DEV), embedding_instance=codes, chunk=32768, **kw)
    ref_rgb = torch.from_numpy(g["rgb_fine_u16"].numpy().astype(np.float64) / 65535.0)
    assert ref_rgb.shape == (640 * 480, 3)
    ours = out["rgb_fine"].cpu().double()
    p = H.psnr(ours, ref_rgb)
    # a fixed synthetic ground truth T (there are no images on the box): reference render + N(0, 0.05), clamped
    target = (ref_rgb + 0.05 * torch.randn(ref_rgb.shape, generator=torch.Generator().manual_seed(7), dtype=torch.float64)).clamp(0, 1)
    dp = abs(H.psnr(ours, target) - H.psnr(ref_rgb, target))
    derr = H.normwise(out["depth_fine"][:: cases.FULL_FRAME["sub"]], g["depth_fine_sub"])
    small = cases.load_golden("frame_toydesk2")            # the 160x120 frame of the same camera carries the fp64 floors
    print("640x480 configs[1]: PSNR(ours, reference) %.1f dB, |dPSNR| vs a fixed target %.5f dB, depth_fine max-norm %.2e "
          "(160x120 floor %.2e), mean rgb ours %.6f / reference %.6f" % (p, dp, derr, float(small["_floor_depth_fine"]),
                                                                        float(ours.mean()), float(g["_mean_rgb"])))
    assert p >= 70.0                                         # 160x120: 74 dB; the reference against its own float64: 68 dB
    assert dp <= 0.1
    assert derr <= max(H.FLOOR_FACTOR * float(small["_floor_depth_fine"]), 2e-5)
    dl2 = H.rel_l2(out["depth_fine"][:: cases.FULL_FRAME["sub"]], g["depth_fine_sub"])
    assert dl2 <= H.FLOOR_FACTOR * float(small["_floor_l2_depth_fine"]), dl2
    # the coarse pass at 640x480 (round 6): every 16th pixel of all six coarse maps against the reference's, tests/golden/coarse_f64.npz
    gc = cases.load_golden("coarse_f64")
    for m in cases.FRAME_MAPS:
        k = m + "_coarse"
        ref = gc["full_frame_toydesk2_sub__" + k]
        err = H.normwise(out[k][:: cases.FULL_FRAME["sub"]], ref)
        assert err <= 2e-5, "640x480 %s: %.3e from the reference (its own fp32-vs-fp64 distance: %.3e)" % (
            k, err, H.normwise(ref, gc["full_frame_toydesk2_sub__" + k + "_f64"]))


def test_edit_demo_frame_from_device_generated_rays():
    """the same configs[4] frame with the three ray sets written by objnerf_generate_rays (row f2) instead of the
    reference's CPU ray / box code: identical hit masks, rays within 2e-6, and the frame still matches the reference's"""
    case = "frame_edit_demo"
    g = cases.load_golden(case)
    sc = scene(cases.frame_inputs(case)[3])
    out = H.render_frame_case(case, sc, device_rays=True)
    ref_sets = H.render_frame_case(case, sc)["_sets"]
    n = out["_sets"][0].shape[0]
    for k in (1, 2):
        ghit = torch.from_numpy(np.unpackbits(g["_hit_%d" % k].numpy())[:n].astype(bool))
        assert torch.equal((out["_sets"][k][:, 7] > 0).cpu(), ghit)
        assert H.normwise(out["_sets"][k], ref_sets[k]) < 2e-6
    assert H.psnr(out["rgb_fine"], g["rgb_fine"]) >= 60.0
