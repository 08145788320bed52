import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["OBJNERF_PATH"] = "layerwise"
import object_nerf_amd as A
from object_nerf_amd import synth
import cases
sc = cases.scene_for(A, "voxel", device="cuda")
rays = synth.camera_rays(320, 240).to("cuda")
n = rays.shape[0]
with torch.no_grad():
    codes = sc.code_library({"instance_ids": synth.per_ray_ids(n).to("cuda")})["embedding_instance"]
    kw = dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, embedding_instance=codes, is_eval=True)
    for _ in range(3):
        A.render_rays(sc.models, sc.embeddings, rays, **kw)
    torch.cuda.synchronize()
