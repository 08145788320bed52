#!/usr/bin/env python
"""Clock / power look at the two workloads (developer tool): runs un-synchronised training steps (tools/train_bench.py's step)
or bench frames back to back while a thread samples the GPU's hwmon files (socket power, shader clock), and prints the
steady-state time per step / frame next to the sampled clock and power.  Answers "is the training step clock-limited?".
Usage (GPU box): python tools/power_probe.py train|render [seconds]"""
import glob
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "power1_cap"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out


class Sampler(threading.Thread):
    def __init__(self, files, period=0.01):
        super().__init__(daemon=True)
        self.files, self.period, self.rows, self.stop = files, period, [], False

    def run(self):
        while not self.stop:
            row = {"t": time.perf_counter()}
            for k, p in self.files.items():
                try:
                    row[k] = float(open(p).read().strip())
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            time.sleep(self.period)


def summarise(rows, t0, t1):
    rows = [r for r in rows if t0 + 0.3 * (t1 - t0) <= r["t"] <= t1]       # steady part
    out = []
    for k in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input"):
        v = sorted(r[k] for r in rows if k in r)
        if v:
            scale = 1e-6 if k.startswith(("power", "freq")) else 1e-3
            out.append("%s median %.0f (min %.0f, max %.0f) %s over %d samples" % (
                k, v[len(v) // 2] * scale, v[0] * scale, v[-1] * scale, "W" if k.startswith("power") else ("MHz" if k.startswith("freq") else "C"), len(v)))
    return out


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "train"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    dev = "cuda"
    files = hwmon_files()
    print("hwmon:", {k: v for k, v in files.items()})
    if "power1_cap" in files:
        print("power cap %.0f W" % (float(open(files["power1_cap"]).read()) * 1e-6))
    if what == "train":
        sc = synth.build_scene(A, True, preset=synth.SCANNET_LIKE, max_voxels=800_000, device=dev)
        rays_all = synth.camera_rays(640, 480).to(dev)
        params = [p for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"]) for p in m.parameters()]
        opt = torch.optim.Adam(params, lr=1e-3)
        g = torch.Generator(device=dev).manual_seed(0)
        n_rays = 2048
        target = torch.rand(n_rays, 3, device=dev, generator=g)
        ids = synth.per_ray_ids(n_rays).to(dev)
        mask = (ids == 1).view(-1, 1)

        def unit():
            idx = torch.randint(0, rays_all.shape[0], (n_rays,), device=dev, generator=g)
            rays = rays_all[idx].contiguous()
            opt.zero_grad(set_to_none=True)
            codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
            r = A.render_rays(sc.models, sc.embeddings, rays, N_samples=64, N_importance=64, perturb=1.0, noise_std=1.0,
                              embedding_instance=codes, frustum_bound_th=0.025, pass_through_mask=mask)
            loss = sum(((r["rgb_%s" % t] - target) ** 2).mean() + ((r["rgb_instance_%s" % t] - target) ** 2).mean()
                       + 0.1 * (r["depth_%s" % t] ** 2).mean() + (r["opacity_instance_%s" % t] ** 2).mean() for t in ("coarse", "fine"))
            loss.backward()
            opt.step()
        label = "training step (2048 rays x (64 + 128), no host synchronisation between steps)"
    else:
        sc = synth.build_scene(A, True, preset=synth.SCANNET_LIKE, max_voxels=800_000, device=dev)
        rays = synth.camera_rays(640, 480).to(dev)
        ids = synth.per_ray_ids(rays.shape[0]).to(dev)
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]

        def unit():
            with torch.no_grad():
                A.render_rays(sc.models, sc.embeddings, rays, N_samples=64, N_importance=64, embedding_instance=codes, frustum_bound_th=0.025)
        label = "640x480 frame, 64 + 64"
    for _ in range(3):
        unit()
    torch.cuda.synchronize()
    smp = Sampler(files)
    smp.start()
    time.sleep(0.2)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            unit()
        n += 4
        if what == "train" and n % 16 == 0:
            torch.cuda.synchronize()          # bound the launch queue (a few steps deep)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    smp.stop = True
    print("%s: %.2f ms per unit over %d units" % (label, (t1 - t0) / n * 1e3, n))
    for line in summarise(smp.rows, t0, t1):
        print("  " + line)


if __name__ == "__main__":
    main()
