import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import helpers as H
from oracle import objnerf_oracle as O
import object_nerf_amd as A
from object_nerf_amd import synth
torch.manual_seed(0)
dev = 'cuda'
for use_voxel in (False, True):
    sc = H.scene(use_voxel, dev)
    # --- teacher-forced MLP branches
    n = 300
    g = torch.Generator().manual_seed(3)
    inx = 271 if use_voxel else 63
    exyz = torch.randn(n, inx, generator=g); edir = torch.randn(n, 27, generator=g)
    ovox = torch.randn(n, 104, generator=g); code = torch.randn(n, 64, generator=g)
    m = sc.models['coarse']
    P = H.state(m)
    with torch.no_grad():
        out = m({'emb_xyz': exyz.to(dev), 'emb_dir': edir.to(dev)})
        outi = m.forward_instance({'emb_xyz': exyz.to(dev), 'emb_dir': edir.to(dev), 'obj_voxel': ovox.to(dev) if use_voxel else None, 'obj_code': code.to(dev)})
    sg, c = O.mlp_scene(P, exyz, edir)
    isg, ic = O.mlp_object(P, exyz, edir, ovox if use_voxel else None, code)
    print('voxel' if use_voxel else 'plain', 'MLP scene sigma %.2e rgb %.2e | obj sigma %.2e rgb %.2e' % (H.normwise(out['sigma'], sg), H.normwise(out['rgb'], c), H.normwise(outi['inst_sigma'], isg), H.normwise(outi['inst_rgb'], ic)))
    # --- end-to-end
    rays = H.test_rays()
    nr = rays.shape[0]
    ids = synth.per_ray_ids(nr)
    with torch.no_grad():
        codes = sc.code_library({'instance_ids': ids.to(dev)})['embedding_instance']
        r = A.render_rays(sc.models, sc.embeddings, rays.to(dev), N_samples=64, N_importance=64, perturb=0, noise_std=0, embedding_instance=codes, is_eval=True)
        torch.cuda.synchronize()
    grid = H.oracle_grid(sc.embeddings['xyz']) if use_voxel else None
    ro = O.render_rays(P, H.state(sc.models['fine']), grid, rays, N_samples=64, N_importance=64, embedding_instance=codes.cpu(), is_eval=True)
    for k in sorted(ro):
        print('   %-26s %.3e' % (k, H.normwise(r[k], ro[k])))
