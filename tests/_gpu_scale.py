import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import helpers as H
import object_nerf_amd as A
from object_nerf_amd import synth
dev='cuda'
t0=time.time()
sc = H.scene(True, dev)
print('scene', time.time()-t0, flush=True)
rays_all = synth.camera_rays(640, 480).to(dev)
for n in (1024, 8192, 32768, 131072, 307200):
    rays = rays_all[:n].contiguous()
    with torch.no_grad():
        ids = torch.full((n,), 1, dtype=torch.long, device=dev)
        codes = sc.code_library({'instance_ids': ids})['embedding_instance'].contiguous()
        for rep in range(2):
            torch.cuda.synchronize(); t0=time.time()
            r = A.render_rays(sc.models, sc.embeddings, rays, N_samples=64, N_importance=64, perturb=0, noise_std=0, embedding_instance=codes, is_eval=True)
            torch.cuda.synchronize(); dt=time.time()-t0
            print(n, 'rays %.4fs  %.2f Mevals/s' % (dt, n*192/dt/1e6), flush=True)
