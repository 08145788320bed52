#!/usr/bin/env python
"""One 3840 x 2160 frame (8.3 M rays, 1.6 G sample points, ~50 GB of device buffers) through render_rays in ONE call:
64-bit indexing of the kernels and the 288 GB memory budget, checked by batch independence against a 4,096-ray slice from
the far end of the frame, and timed.  usage: python tools/big_frame.py [out.md]"""
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
    W, H = 3840, 2160
    sc = synth.build_scene(A, use_voxel=True, preset=synth.TOYDESK2, max_voxels=800_000, device=DEV)
    rays = synth.preset_rays(synth.TOYDESK2, W, H).to(DEV)
    n = rays.shape[0]
    codes = sc.code_library.embedding_instance.weight.detach()[1:2].expand(n, 64).contiguous()
    kw = dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, is_eval=True)
    with torch.no_grad():
        lo = n - 5000
        small = A.render_rays(sc.models, sc.embeddings, rays[lo:lo + 4096].contiguous(), embedding_instance=codes[lo:lo + 4096].contiguous(), **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = A.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for k in r:
        assert torch.equal(r[k][lo:lo + 4096], small[k]), "batch-dependent at the far end of the frame: " + k
    assert torch.isfinite(r["rgb_fine"]).all()
    peak_gb = torch.cuda.max_memory_allocated() / 1e9
    txt = ("| frame | rays | sample points | s | M ray-samples/s | peak device memory GB |\n|---|---|---|---|---|---|\n"
           "| %d x %d | %d | %d | %.2f | %.1f | %.1f |" % (W, H, n, n * 192, dt, n * 192 / dt / 1e6, peak_gb))
    print(txt)
    if out:
        open(out, "w").write("# One 4K frame in one render_rays call (tools/big_frame.py): every key of a 4,096-ray slice at the far end of "
                             "the frame is bit-equal to the same rays rendered alone\n\n" + txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
