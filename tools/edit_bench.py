#!/usr/bin/env python
"""BASELINE configs[2] and configs[4] on one GPU (not the headline bench line; recorded in DESIGN.md):
  config3: render_rays, 640x480, 5 object codes per-ray, 64 + 128 samples, frustum bound, rays_in_bbox
  config5: the editing demo's shape -- ray sets [background, obj 4, obj 4'] generated on the device
           (objnerf_generate_rays with oriented boxes), render_rays_multi 64 + 64 with a removed-object box."""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402
from object_nerf_amd.multi_rendering import render_rays_multi  # noqa: E402
from object_nerf_amd.ray_utils import generate_rays  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], out


def main(W=640, H=480):
    dev = "cuda"
    pre = synth.SCANNET_LIKE
    sc = synth.build_scene(A, True, preset=pre, max_voxels=800_000, device=dev, n_importance=128)
    n = W * H
    with torch.no_grad():
        # ---- config 3
        rays = synth.camera_rays(W, H, near=pre["near"], far=pre["far"]).to(dev)
        ids = synth.per_ray_ids(n).to(dev)
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"].contiguous()
        t, r = timed(lambda: A.render_rays(sc.models, sc.embeddings, rays, N_samples=64, N_importance=128, perturb=0, noise_std=0,
                                           embedding_instance=codes, frustum_bound_th=pre["frustum_bound_th"], is_eval=True,
                                           rays_in_bbox=True))
        ev = n * 256
        print("config3 render_rays 64+128, 5 codes: %.1f ms/frame, %.2f M ray-samples/s, %.1f TFLOP/s"
              % (t * 1e3, ev / t / 1e6, ev * 1776128 / t / 1e12))

        # ---- config 5: device-side ray generation for [bg, obj, obj'] + multi compositing
        focal = (W / 2) / np.tan(math.radians(60.0) / 2)
        cy, sy = math.cos(math.radians(35.0)), math.sin(math.radians(35.0))
        cp, sp = math.cos(math.radians(75.0)), math.sin(math.radians(75.0))
        R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]) @ np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Twc = np.concatenate([R, np.array([[0.5], [0.5], [0.6]])], 1)
        box = synth.oriented_box([3.6, 3.9, 0.5], [1.0, 0.8, 1.0], 20.0, pre["scene_center"], pre["scale_factor"])

        def moved(dx, dy, yaw):
            c, s = math.cos(math.radians(yaw)), math.sin(math.radians(yaw))
            T = np.eye(4); T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]; T[:3, 3] = [dx, dy, 0]
            M = np.eye(4); M[:3] = Twc
            return (np.linalg.inv(T) @ M)[:3]

        def frame():
            sets = [generate_rays(H, W, focal, Twc, pre["near"], pre["far"]),
                    generate_rays(H, W, focal, moved(0.05, 0.2, 10.0), box=box, bbox_enlarge=0.06),
                    generate_rays(H, W, focal, moved(-0.05, -0.1, -10.0), box=box, bbox_enlarge=0.06)]
            return render_rays_multi(sc.models, sc.embeddings, sc.code_library, sets, [0, 4, 4], N_samples=64, N_importance=64,
                                     perturb=0, noise_std=0, background_skip_bbox={4: box}), sets
        t, (r, sets) = timed(frame)
        hit = [(s[:, 7] > 0).float().mean().item() for s in sets]
        # evaluated sample points: rays that missed their object's box are compacted away before the MLP kernel
        ev = n * 192 * sum(hit)
        flop = n * 192 * (1399808 * hit[0] + 376320 * (hit[1] + hit[2]))
        print("config5 render_rays_multi [0,4,4] 64+64 (+ device ray generation): %.1f ms/frame, %.2f M evaluated ray-samples/s (one "
              "branch each), %.1f TFLOP/s; box hit fraction %s; mean rgb %.4f"
              % (t * 1e3, ev / t / 1e6, flop / t / 1e12, ["%.2f" % h for h in hit], r["rgb_fine"].mean().item()))


if __name__ == "__main__":
    main()
