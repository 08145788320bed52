"""CPU, world_size 2 over gloo: the ray-sharding + pixel all-gather wrapper (the N>1 path of
bench.py / SURVEY.md §8e).  The renderer itself needs a GPU, so a per-ray stand-in function plays
its role here; what is tested is the partitioning, padding of short shards and the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_nerf_amd.distributed import (GradientSync, gather_pixels, render_rays_multi_sharded, render_rays_sharded,
                                         shard_rays)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(rays, embedding_instance=None, scale=1.0, **kw):
    # any function that is independent per ray
    rgb = torch.stack([rays[:, 0] * scale, rays[:, 3] + embedding_instance[:, 0], rays[:, 7]], -1)
    return {"rgb_fine": rgb, "depth_fine": rays[:, 6] * 2, "weights_fine": rays[:, :4]}


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        rays = torch.randn(n, 8, generator=g)
        codes = torch.randn(n, 64, generator=g)
        full = _fake_render(rays, codes, scale=3.0)
        out = render_rays_sharded(_fake_render, rays, {"embedding_instance": codes}, scale=3.0,
                                  gather_keys=("rgb_fine", "depth_fine"))
        ok = torch.equal(out["rgb_fine"], full["rgb_fine"]) and torch.equal(out["depth_fine"], full["depth_fine"])
        ok = ok and "weights_fine" not in out
        r_loc, ex = shard_rays(rays, {"embedding_instance": codes, "flag": 1.5})
        ok = ok and ex["flag"] == 1.5 and ex["embedding_instance"].shape[0] == r_loc.shape[0]
        ok = ok and torch.equal(gather_pixels(r_loc[:, 0], n), rays[:, 0])
        # multi-object compositor: all ray sets cut at the same bounds
        sets = [rays, rays * 2.0 + 1.0, rays - 3.0]

        def fake_multi(rays_list, obj_instance_ids, **kw):
            acc = sum(r[:, :3] * float(i + 1) for i, r in zip(obj_instance_ids, rays_list))
            return {"rgb_fine": acc, "depth_fine": rays_list[0][:, 6] + rays_list[-1][:, 7]}
        want = fake_multi(sets, [0, 4, 2])
        got = render_rays_multi_sharded(fake_multi, sets, obj_instance_ids=[0, 4, 2], gather_keys=("rgb_fine", "depth_fine"))
        ok = ok and torch.equal(got["rgb_fine"], want["rgb_fine"]) and torch.equal(got["depth_fine"], want["depth_fine"])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7, 1])
def test_sharded_render_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        shapes = [(256, 271), (256,), (64, 64), (1000, 24), (3, 128), (1,)]
        params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
        frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)
        grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(world)]     # same on every rank
        for i, p in enumerate(params):
            p.grad = grads[rank][i].clone()
        params[2].grad = None if rank == 1 else params[2].grad        # e.g. codes no local ray used
        # 64 KiB buckets: forces several buckets, one parameter larger than a bucket
        sync = GradientSync(params + [frozen], bucket_bytes=64 << 10)
        sync.sync()
        ok = len(sync.buckets) >= 3 and frozen.grad is None
        for i, p in enumerate(params):
            want = sum(grads[r][i] for r in range(world)) / world
            if i == 2:
                want = grads[0][i] / world
            ok = ok and torch.allclose(p.grad, want, atol=1e-6) and p.grad.shape == p.shape
        sync.sync()                                                     # second step reuses the flat buffers
        ok = ok and torch.allclose(params[0].grad, sum(grads[r][0] for r in range(world)) / world, atol=1e-6)
        # voxel-table handling: only the rows the index map can reach (a prefix) travel
        table = torch.nn.Parameter(torch.zeros(1000, 24))
        tg = [torch.randn(300, 24, generator=g) for _ in range(world)]
        table.grad = torch.zeros(1000, 24)
        table.grad[:300] = tg[rank]
        table.grad[900] = float(rank + 1)          # outside the active prefix: stays local (never happens in the renderer)
        s2 = GradientSync([params[1], table], active_rows={table: 300})
        ok = ok and s2.message_bytes() == [4 * (256 + 300 * 24)]
        s2.sync()
        ok = ok and torch.allclose(table.grad[:300], sum(tg) / world, atol=1e-6)
        ok = ok and bool((table.grad[900] == float(rank + 1)).all()) and bool((table.grad[300:900] == 0).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradient_sync_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_gradient_sync_without_process_group_is_a_noop():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradientSync([p]).sync()
    assert torch.equal(p.grad, torch.full((3,), 2.0))
