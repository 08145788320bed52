"""Shared test plumbing: scene construction for the drop-in types and the oracle's view of it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402


def state(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def oracle_grid(ev):
    """plain-tensor view of an EmbeddingVoxel for oracle.objnerf_oracle"""
    return dict(voxel_idx_map=ev.voxel_idx_map.cpu(), table=ev.embedding_space_ftr.weight.detach().cpu(),
                voxel_offset=ev.voxel_offset.cpu(), voxel_size=ev.voxel_size.cpu(), voxel_shape=ev.voxel_shape.cpu())


def test_rays(n=96, w=64, h=48, stride=29, **kw):
    rays_all = synth.camera_rays(w, h, **kw)
    idx = torch.arange(0, rays_all.shape[0], stride)[:n]
    return rays_all[idx].contiguous()


def scene(use_voxel=True, device="cpu", max_voxels=120000, n_points=200_000, preset=synth.SCANNET_LIKE):
    return synth.build_scene(A, use_voxel, preset=preset, max_voxels=max_voxels, n_points=n_points, device=device)


def normwise(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def cdf_position(bins, weights, z, eps=1e-5):
    """F(z): position of samples z (N,I) on the float64 piecewise-linear CDF that sample_pdf inverts
    (bins (N,nb), weights (N,nb-1)); used to grade samplers in the well-conditioned domain."""
    bins, w, z = bins.double(), weights.double() + eps, z.double()
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    idx = (torch.searchsorted(bins.contiguous(), z.contiguous(), right=True) - 1).clamp(0, bins.shape[1] - 2)
    b0, b1 = torch.gather(bins, 1, idx), torch.gather(bins, 1, idx + 1)
    c0, c1 = torch.gather(cdf, 1, idx), torch.gather(cdf, 1, idx + 1)
    t = ((z - b0) / (b1 - b0).clamp_min(1e-300)).clamp(0, 1)
    return c0 + t * (c1 - c0)


def sampler_residual(bins, weights, u, z, eps=1e-5):
    """Per-sample grade of an inverse-CDF sampler, robust to both kinds of ill-conditioning:
    flat CDF (tiny pdf: z is sensitive, F(z) is not) and steep CDF (narrow heavy bin: F(z) is
    sensitive to one ulp of z, z is not).  Returns min(|F64(z) - u|, |z - z64| / max|bins|)."""
    from oracle import objnerf_oracle as O
    q = cdf_position(bins, weights, z, eps)
    z64 = O.sample_pdf(bins.double(), weights.double(), u.shape[1], det=False, eps=eps, u=u.double())
    rz = (z.double() - z64).abs() / bins.double().abs().max()
    return torch.minimum((q - u.double()).abs(), rz)
