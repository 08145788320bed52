import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, cases, helpers as H
import object_nerf_amd as A
import test_gpu_render as T
def l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a-b).norm() / b.norm().clamp_min(1e-30)).item()
for case in sorted(cases.RENDER_CASES):
    c = cases.RENDER_CASES[case]
    sc = T.scene(c['scene']); use_voxel = cases.SCENES[c['scene']][0]
    g = cases.load_golden('render_' + case)
    rays, ids, ptm, randoms = cases.render_inputs(case)
    kw = dict(c['kw']); kw.setdefault('perturb', 0); kw.setdefault('noise_std', 0)
    with torch.no_grad():
        codes = sc.code_library({'instance_ids': ids.to('cuda')})['embedding_instance']
        rd = None
        if randoms: rd = dict(perturb_rand=randoms['perturb_rand'].cuda(), u_rand=randoms['u_rand'].cuda(), noise=[t.cuda() for t in randoms['noise']])
        out = A.render_rays(sc.models, sc.embeddings, rays.cuda(), embedding_instance=codes, pass_through_mask=ptm.cuda() if ptm is not None else None, _randoms=rd, **kw)
    f64 = T.oracle_f64(sc, use_voxel, rays, g['_codes'], ptm, randoms, c['kw'])
    print(case)
    for k in sorted(g):
        if k.startswith('_'): continue
        print('   %-24s max: err %.2e floor %.2e ratio %5.1f | l2: err %.2e floor %.2e ratio %5.1f' % (k, H.normwise(out[k], g[k]), H.normwise(g[k], f64[k]), H.normwise(out[k], g[k])/max(H.normwise(g[k], f64[k]),1e-12), l2(out[k], g[k]), l2(g[k], f64[k]), l2(out[k], g[k])/max(l2(g[k], f64[k]),1e-12)))
