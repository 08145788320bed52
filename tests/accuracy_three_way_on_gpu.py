#!/usr/bin/env python
"""Three-way accuracy table (not a pytest file; run by hand on the GPU box): the HIP path, the reference's PyTorch path
run by PyTorch-ROCm on the same GPU, and the same path on the CPU (the oracle, bit-exact with the reference there), on
512 rays of the bench scene.  Shows that the HIP path is as close to the CPU reference as PyTorch's own GPU result is.
noise_std / perturb must be passed as 0 explicitly: render_rays' default noise_std is 1 (models/rendering.py:240)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A
from object_nerf_amd import synth
from oracle import objnerf_oracle as O
from tests.perf_torch_path_on_gpu import oracle_state

from object_nerf_amd import _lib
if os.environ.get("DIAG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["DIAG_LIB"])
    print("library:", _lib.LIB_PATH)
dev = "cuda"
print("matmul precision:", torch.get_float32_matmul_precision(), "allow_tf32:", torch.backends.cuda.matmul.allow_tf32)
x = torch.randn(4096, 256); w = torch.randn(256, 256); b = torch.randn(256)
ref = torch.addmm(b.double(), x.double(), w.double().t())
cpu = torch.addmm(b, x, w.t())
gpu = torch.addmm(b.to(dev), x.to(dev), w.to(dev).t()).cpu()
print("addmm 4096x256x256 rel err: cpu %.2e  gpu %.2e" % (((cpu - ref).abs().max() / ref.abs().max()).item(), ((gpu - ref).abs().max() / ref.abs().max()).item()))

sc = synth.build_scene(A, True, preset=synth.TOYDESK_LIKE, device=dev)
rays = synth.camera_rays(640, 480, near=0.05, far=1.5).to(dev)[::601][:512].contiguous()
ids = synth.per_ray_ids(rays.shape[0]).to(dev)
codes = sc.code_library({"instance_ids": ids})["embedding_instance"].detach()
kw = dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, frustum_bound_th=0.025, is_eval=True)
with torch.no_grad():
    hip = A.render_rays(sc.models, sc.embeddings, rays, embedding_instance=codes, **kw)
    g, pc, pf = oracle_state(sc, dev)
    with torch.device(dev):
        tg = O.render_rays(pc, pf, g, rays, embedding_instance=codes, **kw)
    g, pc, pf = oracle_state(sc, "cpu")
    tc = O.render_rays(pc, pf, g, rays.cpu(), embedding_instance=codes.cpu(), **kw)
def psnr(a, b):
    return (-10 * torch.log10(((a.double().cpu() - b.double().cpu()) ** 2).mean().clamp_min(1e-30))).item()
for k in ("rgb_coarse", "rgb_fine", "depth_fine", "rgb_instance_fine", "opacity_instance_fine"):
    print("%-22s PSNR hip-vs-cpu %.1f  torchgpu-vs-cpu %.1f  hip-vs-torchgpu %.1f" % (k, psnr(hip[k], tc[k]), psnr(tg[k], tc[k]), psnr(hip[k], tg[k])))
d = (tg["rgb_fine"].cpu() - tc["rgb_fine"]).abs().max(1)[0]
print("torchgpu-vs-cpu rgb_fine: rays with |diff| > 1e-3: %d of %d; max %.3e" % ((d > 1e-3).sum().item(), d.numel(), d.max().item()))
d = (hip["rgb_fine"].cpu() - tc["rgb_fine"]).abs().max(1)[0]
print("hip-vs-cpu      rgb_fine: rays with |diff| > 1e-3: %d of %d; max %.3e" % ((d > 1e-3).sum().item(), d.numel(), d.max().item()))
