#!/usr/bin/env python
"""Times objnerf_ray_bias alone at the headline frame's ray count (307,200 rays, both branches) on the library OBJNERF_LIB names:
the attribution probes of round 6 (libraries built from a temporary patch of csrc/ray_kernels.hip that compiled the kernel's stores,
its input loads or its MFMAs out -- the patch is not kept, profiles/r06_ray_bias_probe.txt holds what it measured) against the
shipped kernel.  tools/ray_bias_probe.py [n_rays]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import _lib, synth  # noqa: E402
import cases  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
dev = "cuda"
sc = cases.scene_for(A, "voxel", device=dev)
rays = synth.camera_rays(640, 480).to(dev)[:n].contiguous()
codes = torch.randn(n, 64, device=dev)
blob, aux = sc.models["coarse"].packed()
l = _lib.lib()
a = _lib.MlpArgs()
a.use_voxel, a.do_scene, a.do_object = 1, 1, 1
a.blob, a.aux = blob.data_ptr(), aux.data_ptr()
a.rays, a.n_rays, a.S = rays.data_ptr(), n, 64
a.codes, a.code_stride = codes.data_ptr(), 64
rb = torch.empty(l.objnerf_ray_bias_floats(n), device=dev)
st = _lib.stream_ptr()
for _ in range(5):
    _lib.check(l.objnerf_ray_bias(C.byref(a), _lib.ptr(rb), st), "ray_bias")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    _lib.check(l.objnerf_ray_bias(C.byref(a), _lib.ptr(rb), st), "ray_bias")
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print("%s: ray_bias %d rays: %.1f us per launch (%.2f TB/s of its %d MB output)" % (
    os.path.basename(os.environ.get("OBJNERF_LIB", "shipped")), n, us, rb.numel() * 4 / us / 1e6, rb.numel() * 4 >> 20))
