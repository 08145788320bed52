"""GPU: the training path -- fp32 MFMA GEMM, layer-wise MLP forward/backward,
compositing backward, voxel-embedding backward and the end-to-end gradients of render_rays -- against
PyTorch autograd through the CPU oracle (oracle/objnerf_oracle.py is plain differentiable torch)."""
import ctypes as C

import pytest
import torch

import cases
import helpers as H
import object_nerf_amd as A
from object_nerf_amd import _lib, synth
from oracle import objnerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(300, 256, 271), (1000, 3, 128), (1, 256, 5000), (257, 129, 33), (64, 527, 70000)])
def test_gemm_matches_torch(akc, bkc, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    Ap = torch.randn(M, K, generator=g)        # logical A' (M x K), B' (K x N)
    Bp = torch.randn(K, N, generator=g)
    A_st = (Ap if akc else Ap.t()).contiguous().to(DEV)
    B_st = (Bp.t() if bkc else Bp).contiguous().to(DEV)
    want = Ap.double() @ Bp.double()
    l = _lib.lib()
    for split in (1, 7):
        Cm = torch.full((M, N), 0.5, device=DEV)
        _lib.check(l.objnerf_gemm(_lib.ptr(A_st), A_st.shape[1], akc, _lib.ptr(B_st), B_st.shape[1], bkc, _lib.ptr(Cm), N, M, N, K,
                                  1, 0, None, split, _lib.stream_ptr()), "gemm")
        err = ((Cm.cpu().double() - 0.5 - want).abs().max() / want.abs().max()).item()
        assert err < 2e-5 * max(1.0, (K / 1000) ** 0.5), (split, err)
    # epilogue: bias + LeakyReLU, overwrite
    bias = torch.randn(N, generator=g)
    Cm = torch.empty(M, N, device=DEV)
    bd = bias.to(DEV)
    _lib.check(l.objnerf_gemm(_lib.ptr(A_st), A_st.shape[1], akc, _lib.ptr(B_st), B_st.shape[1], bkc, _lib.ptr(Cm), N, M, N, K,
                              0, 2, _lib.ptr(bd), 1, _lib.stream_ptr()), "gemm")
    ref = torch.nn.functional.leaky_relu(want + bias.double(), 0.01)
    assert ((Cm.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-5 * max(1.0, (K / 1000) ** 0.5)


def _loss(res, seed=0):
    """a fixed random linear functional of every differentiable output"""
    g = torch.Generator().manual_seed(seed)
    tot = 0.0
    for k in sorted(res):
        if k.startswith(("weights_", "z_vals_")):
            continue
        wgt = torch.randn(res[k].shape, generator=g).to(res[k].device)
        tot = tot + (res[k] * wgt).sum()
    return tot


@pytest.mark.parametrize("case", ["voxel_train", "plain_train", "voxel_eval_flags", "voxel_random"])
def test_render_rays_gradients_match_oracle_autograd(case):
    cfgs = {
        "voxel_train": dict(scene="voxel", kw=dict(is_eval=False, frustum_bound_th=0.025, rays_in_bbox=False), ptm=True),
        "plain_train": dict(scene="plain", kw=dict(is_eval=False, frustum_bound_th=-1.0, white_back=True)),
        "voxel_eval_flags": dict(scene="sparse", kw=dict(is_eval=True, use_zero_as_last_delta=True, use_disp=True, rays_in_bbox=True)),
        "voxel_random": dict(scene="voxel", kw=dict(is_eval=False, frustum_bound_th=0.025, perturb=1.0, noise_std=1.0), rnd=True),
    }
    c = cfgs[case]
    S, I, n = 16, 16, 24
    sc = cases.scene_for(A, c["scene"], device=DEV)
    use_voxel = cases.SCENES[c["scene"]][0]
    rays = H.test_rays(n, stride=97)
    ids = synth.per_ray_ids(n, seed=5)
    ptm = (torch.arange(n) % 3 == 0).view(n, 1) if c.get("ptm") else None
    randoms = None
    if c.get("rnd"):
        g = torch.Generator().manual_seed(2)
        randoms = dict(perturb_rand=torch.rand(n, S, generator=g), u_rand=torch.rand(n, I, generator=g),
                       noise=[torch.randn(n, S, generator=g), torch.randn(n, S, generator=g),
                              torch.randn(n, S + I, generator=g), torch.randn(n, S + I, generator=g)])
    kw = dict(N_samples=S, N_importance=I, perturb=0, noise_std=0)
    kw.update(c["kw"])

    # ---- HIP training path
    for m in (sc.models["coarse"], sc.models["fine"], sc.code_library, sc.embeddings["xyz"] if use_voxel else None):
        if m is not None:
            m.zero_grad()
    codes = sc.code_library({"instance_ids": ids.to(DEV)})["embedding_instance"]
    rd = None
    if randoms:
        rd = dict(perturb_rand=randoms["perturb_rand"].to(DEV), u_rand=randoms["u_rand"].to(DEV),
                  noise=[t.to(DEV) for t in randoms["noise"]])
    res = A.render_rays(sc.models, sc.embeddings, rays.to(DEV), embedding_instance=codes,
                        pass_through_mask=ptm.to(DEV) if ptm is not None else None, _randoms=rd, **kw)
    assert all(res[k].requires_grad for k in res if k.startswith("rgb_"))
    _loss(res).backward()

    # ---- oracle autograd on the CPU
    pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sc.models["coarse"].state_dict().items()}
    pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sc.models["fine"].state_dict().items()}
    ctab = sc.code_library.embedding_instance.weight.detach().cpu().clone().requires_grad_(True)
    grid = None
    if use_voxel:
        grid = H.oracle_grid(sc.embeddings["xyz"])
        grid["table"] = grid["table"].clone().requires_grad_(True)
    # teacher-forced fine depths: both sides differentiate at the same sample points (the sampler itself carries no
    # gradient, rendering.py:307, and its fp32 sensitivity is graded separately in test_gpu_render.py)
    ro = O.render_rays(pc, pf, grid, rays, embedding_instance=ctab[ids], pass_through_mask=ptm, randoms=randoms,
                       z_fine_override=res["z_vals_fine"].detach().cpu(), **{k: v for k, v in kw.items()})
    _loss({k: v for k, v in ro.items()}).backward()

    # forward values of the training path agree with the oracle as well
    for k in ro:
        assert H.normwise(res[k], ro[k]) < 1e-4, k
    worst = 0.0
    for typ, mod, ref in (("coarse", sc.models["coarse"], pc), ("fine", sc.models["fine"], pf)):
        for name, p in mod.named_parameters():
            assert p.grad is not None, name
            if ref[name].grad is None:
                assert p.grad.abs().max().item() == 0, name
                continue
            e = rel_l2(p.grad, ref[name].grad)
            worst = max(worst, e)
            assert e < 2e-4, "%s %s: rel L2 grad error %.3e" % (typ, name, e)
    e = rel_l2(sc.code_library.embedding_instance.weight.grad, ctab.grad)
    assert e < 2e-4, "codes: %.3e" % e
    if use_voxel:
        tg = sc.embeddings["xyz"].embedding_space_ftr.weight.grad
        assert tg is not None
        e = rel_l2(tg, grid["table"].grad)
        assert e < 2e-4, "voxel table: %.3e" % e
    print(case, "worst parameter-gradient rel L2 error %.2e" % worst)


def test_training_step_updates_weights_and_repacks():
    """an optimizer step on the HIP gradients changes the render, through the automatic weight repack"""
    sc = cases.scene_for(A, "plain", device=DEV)
    rays = H.test_rays(32).to(DEV)
    ids = synth.per_ray_ids(32).to(DEV)
    params = [p for m in (sc.models["coarse"], sc.models["fine"], sc.code_library) for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3)
    target = torch.full((32, 3), 0.25, device=DEV)

    def step():
        opt.zero_grad()
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        r = A.render_rays(sc.models, sc.embeddings, rays, N_samples=16, N_importance=16, perturb=0, noise_std=0,
                          embedding_instance=codes)
        loss = ((r["rgb_fine"] - target) ** 2).mean() + ((r["rgb_coarse"] - target) ** 2).mean()
        loss.backward()
        opt.step()
        return loss.item()
    losses = [step() for _ in range(8)]
    assert losses[-1] < losses[0]
    with torch.no_grad():      # inference path sees the updated parameters (re-packed weight stream)
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        r_inf = A.render_rays(sc.models, sc.embeddings, rays, N_samples=16, N_importance=16, perturb=0, noise_std=0,
                              embedding_instance=codes)
    codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
    r_tr = A.render_rays(sc.models, sc.embeddings, rays, N_samples=16, N_importance=16, perturb=0, noise_std=0,
                         embedding_instance=codes)
    assert H.normwise(r_inf["rgb_coarse"], r_tr["rgb_coarse"]) < 1e-4      # GEMM path == fused MFMA path
