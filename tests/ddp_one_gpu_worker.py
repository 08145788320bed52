"""Worker of tests/test_gpu_dist.py::test_ddp_wrapper_equals_gradient_sync (not collected by pytest).

    python tests/ddp_one_gpu_worker.py RANK WORLD PORT

What Lightning instantiates for the reference's `accelerator="ddp"` (train.py:261-262) is
`torch.nn.parallel.DistributedDataParallel` around the LightningModule whose forward calls render_rays with its own
sub-modules (train.py:45-98).  Here: the drop-in modules (ObjectNeRF x 2, CodeLibrary, EmbeddingVoxel) inside one nn.Module,
wrapped in DDP over gloo with every rank on cuda:0 (RCCL refuses two ranks on one device), ONE training step on a
rank-specific batch -- the differentiable HIP render_rays is one autograd node that hands every parameter gradient to DDP's
reducer at once, with the dense voxel-table gradient in its buckets -- against the same step on an identical copy of the
modules followed by object_nerf_amd.distributed.GradientSync.  Averaged gradients must be bit-equal wherever the backward
is bit-reproducible (both MLPs, the codes; the voxel table's scatter uses fp32 atomics: 1e-6), then one Adam step on each
leaves the parameters equal to the same bound, and a SECOND DDP step runs (the reducer was left in a consistent state)."""
import copy
import os
import sys

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import helpers as H  # noqa: E402
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402
from object_nerf_amd.distributed import GradientSync  # noqa: E402


class System(nn.Module):
    """the part of train.py::ObjectNeRFSystem that owns parameters: models, embeddings, code library + forward = render_rays"""

    def __init__(self, sc):
        super().__init__()
        self.models = nn.ModuleDict({"coarse": sc.models["coarse"], "fine": sc.models["fine"]})
        self.embedding_xyz = sc.embeddings["xyz"]
        self.embedding_dir = sc.embeddings["dir"]
        self.code_library = sc.code_library

    def forward(self, rays, ids):
        codes = self.code_library({"instance_ids": ids})["embedding_instance"]
        emb = {"xyz": self.embedding_xyz, "dir": self.embedding_dir}
        return A.render_rays({"coarse": self.models["coarse"], "fine": self.models["fine"]}, emb, rays, N_samples=32,
                             N_importance=32, perturb=0, noise_std=0, embedding_instance=codes, frustum_bound_th=0.025)


def loss_of(r):
    return sum((r["rgb_%s" % t] ** 2).mean() + (r["rgb_instance_%s" % t] ** 2).mean() + 0.1 * (r["depth_%s" % t] ** 2).mean()
               + (r["opacity_instance_%s" % t] ** 2).mean() for t in ("coarse", "fine"))


def main(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sc = cases.scene_for(A, "voxel", device="cuda")             # seeded: the same weights on every rank
    sys_ddp = System(sc)
    sys_own = copy.deepcopy(sys_ddp)
    n = 96
    rays = H.test_rays(n * world, w=128, h=96, stride=7)[rank::world].contiguous().cuda()      # a different batch per rank
    ids = synth.per_ray_ids(n, seed=11 + rank).cuda()
    ddp = nn.parallel.DistributedDataParallel(sys_ddp, device_ids=[0])
    loss_of(ddp(rays, ids)).backward()
    loss_of(sys_own(rays, ids)).backward()
    names = [k for k, _ in sys_own.named_parameters()]
    params = [p for _, p in sys_own.named_parameters()]
    table = sys_own.embedding_xyz.embedding_space_ftr.weight
    GradientSync(params, active_rows={table: sys_own.embedding_xyz.active_rows()}).sync()
    torch.cuda.synchronize()
    got = dict(sys_ddp.named_parameters())
    n_equal = 0
    for k, p in zip(names, params):
        a, b = got[k].grad, p.grad
        assert a is not None and b is not None, k
        assert a.abs().max().item() > 0 or "embedding_instance" in k or b.abs().max().item() == 0, k
        if p is table:                                          # fp32 atomics in the scatter: not bit-reproducible run to run
            assert H.normwise(a, b) < 1e-5, (k, H.normwise(a, b))
        else:
            assert torch.equal(a, b), "rank %d: %s differs between DDP and GradientSync (%.3e)" % (rank, k, H.normwise(a, b))
            n_equal += 1
    # every rank holds the same averaged gradient: a digest of rank 0's travels to the others
    digest = torch.stack([got[k].grad.double().sum() for k in names if got[k] is not got.get("embedding_xyz.embedding_space_ftr.weight")]).cpu()
    ref = digest.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(digest, ref), "ranks disagree on the averaged gradients"
    # one optimizer step each, then a second DDP step: the reducer is re-armed, the new weights are rendered (no stale stream)
    o1 = torch.optim.Adam(sys_ddp.parameters(), lr=1e-3, fused=True)
    o2 = torch.optim.Adam(sys_own.parameters(), lr=1e-3, fused=True)
    o1.step(); o2.step()
    for k, p in zip(names, params):
        if p is not table:
            assert torch.equal(got[k].detach(), p.detach()), k
    o1.zero_grad(set_to_none=True)
    l1 = loss_of(ddp(rays, ids))
    l1.backward()
    torch.cuda.synchronize()
    assert all(p.grad is not None for p in sys_ddp.parameters())
    dist.barrier()
    if rank == 0:
        print("ddp ok: %d tensors bit-equal, table within 1e-5, second step loss %.6f" % (n_equal, l1.item()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
