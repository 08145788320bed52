#!/usr/bin/env python
"""Small-batch behaviour of render_rays (GPU box): the editor renders <= 4,096-ray chunks (test/config/*.yaml:4) and a
training step 2,048 rays, far from the 307,200-ray frame the headline is quoted on.  Per batch size: wall time per call
(host + device, calls issued back to back), device time per call (events around the loop), and host-only time per call
(the same loop with the device idle at the end) -> where the time goes.  Last column: the same call captured ONCE in a
hipGraph (torch.cuda.CUDAGraph: the library only enqueues on the caller's stream and never allocates or synchronises, so a
whole render_rays call is capturable) and replayed -- no host issue, no gaps between the call's eight kernels.
usage: python tools/small_batch.py [out.md]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402

DEV = "cuda"


def main(out=None):
    sc = synth.build_scene(A, use_voxel=True, preset=synth.TOYDESK2, max_voxels=800_000, device=DEV)
    rays_all = synth.preset_rays(synth.TOYDESK2, 640, 480).to(DEV)
    lines = ["| rays / call | calls | wall ms / call | device ms / call | host-issue ms / call | M ray-samples/s (wall) | of frame rate "
             "| hipGraph replay: device ms / call | of frame rate |",
             "|---|---|---|---|---|---|---|---|---|"]
    frame_rate = None
    sizes = (307200, 32768, 8192, 4096, 2048, 1024)
    if os.environ.get("SMALL_BATCH_ONLY"):        # e.g. under rocprofv3 --kernel-trace: one size only (the first is the "frame rate")
        sizes = tuple(int(x) for x in os.environ["SMALL_BATCH_ONLY"].split(","))
    for n in sizes:
        idx = torch.linspace(0, rays_all.shape[0] - 1, n).long().to(DEV)
        rays = rays_all[idx].contiguous()
        codes = sc.code_library({"instance_ids": torch.ones(n, dtype=torch.long, device=DEV)})["embedding_instance"].detach()
        kw = dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, embedding_instance=codes, is_eval=True)
        calls = 3 if n > 100000 else 30
        with torch.no_grad():
            for _ in range(2):
                A.render_rays(sc.models, sc.embeddings, rays, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(calls):
                A.render_rays(sc.models, sc.embeddings, rays, **kw)
            e1.record()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        dev_ms = e0.elapsed_time(e1) / calls
        rate = n * 192 * calls / wall / 1e6
        if frame_rate is None:
            frame_rate = rate
        graph_ms, graph_frac = "n/a", "n/a"
        if n <= 32768:
            try:
                with torch.no_grad():
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        A.render_rays(sc.models, sc.embeddings, rays, **kw)
                    torch.cuda.synchronize()
                    g.replay()
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(calls):
                        g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                gm = e0.elapsed_time(e1) / calls
                graph_ms, graph_frac = "%.3f" % gm, "%.3f" % (n * 192 / (gm * 1e-3) / 1e6 / frame_rate)
            except Exception as e:      # noqa: BLE001 -- report, keep the table
                graph_ms = "failed: %s" % (str(e).splitlines()[0][:80])
        lines.append("| %d | %d | %.3f | %.3f | %.3f | %.1f | %.2f | %s | %s |" % (n, calls, 1e3 * wall / calls, dev_ms, 1e3 * t_issue / calls, rate,
                                                                             rate / frame_rate, graph_ms, graph_frac))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write("# render_rays at small batch sizes (tools/small_batch.py, 64+64, scene+object, ToyDesk-2 preset)\n\n" + txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
