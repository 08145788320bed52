"""CPU, world_size 2 over gloo: the ray-sharding + pixel all-gather wrapper (the N>1 path of
bench.py / SURVEY.md §8e).  The renderer itself needs a GPU, so a per-ray stand-in function plays
its role here; what is tested is the partitioning, padding of short shards and the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_nerf_amd.distributed import (GradientSync, RayShards, default_block, gather_pixel_maps, gather_pixels,
                                         render_rays_multi_sharded, render_rays_sharded, shard_rays)


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
        # every split of the same frame gives the same pixels: contiguous bands, block-cyclic with a ragged last block,
        # blocks larger than the frame (most ranks empty), and rays handed over already dealt (rays_are_local)
        for sh in (RayShards(n, world), RayShards(n, world, 3), RayShards(n, world, 1), RayShards(n, world, n + 5),
                   RayShards(n, world, default_block(n, world))):
            ok = ok and sum(sh.counts) == n and sh.per == max(sh.counts)
            got = render_rays_multi_sharded(fake_multi, sets, obj_instance_ids=[0, 4, 2], gather_keys=("rgb_fine", "depth_fine"),
                                            shards=sh)
            ok = ok and torch.equal(got["rgb_fine"], want["rgb_fine"]) and torch.equal(got["depth_fine"], want["depth_fine"])
            mine = [sh.take(r, rank) for r in sets]
            got = render_rays_multi_sharded(fake_multi, mine, obj_instance_ids=[0, 4, 2], gather_keys=("rgb_fine",),
                                            shards=sh, rays_are_local=True)
            ok = ok and torch.equal(got["rgb_fine"], want["rgb_fine"])
            out = render_rays_sharded(_fake_render, rays, {"embedding_instance": codes}, scale=3.0,
                                      gather_keys=("rgb_fine", "depth_fine"), shards=sh)
            ok = ok and torch.equal(out["rgb_fine"], full["rgb_fine"]) and torch.equal(out["depth_fine"], full["depth_fine"])
            # packed maps of mixed rank (a (n,) and a (n, 3) map in one message)
            loc = {"a": sh.take(rays[:, 5], rank), "b": sh.take(rays[:, :3], rank)}
            gm = gather_pixel_maps(loc, n, sh)
            ok = ok and torch.equal(gm["a"], rays[:, 5]) and torch.equal(gm["b"], rays[:, :3])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_sharded(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("n", [10, 7, 1])
def test_sharded_render_world2(n):
    _run_sharded(2, n)


@pytest.mark.parametrize("n", [37, 5])
def test_sharded_render_world8_ragged_and_empty_shards(n):
    """8 ranks (the node size the driver scales to): n = 37 -> ragged bands (5,5,5,5,5,5,5,2); n = 5 -> three ranks with
    nothing to render at all.  Every split reproduces the single-process frame bit for bit."""
    _run_sharded(8, n)


def test_ray_shards_layouts():
    """the dealing itself, no process group: counts, the block-cyclic order, whole-row shards as objnerf_generate_rays_rows
    writes them (ray_utils.row_share), restore() = inverse permutation of the padded all-gather buffer"""
    from object_nerf_amd.ray_utils import row_share
    for n, world, block in [(307200, 8, None), (307200, 8, 2560), (307201, 3, 1000), (10, 8, None), (10, 8, 3), (0, 4, 5),
                            (5, 8, None), (307200, 8, default_block(307200, 8))]:
        sh = RayShards(n, world, block)
        x = torch.arange(n, dtype=torch.float32)[:, None]
        g = torch.full((world * sh.per, 1), -1.0)
        for r in range(world):
            loc = sh.take(x, r)
            assert loc.shape[0] == sh.counts[r] and torch.equal(loc[:, 0].long(), sh.local_index(r))
            g[r * sh.per:r * sh.per + sh.counts[r]] = loc
        assert sum(sh.counts) == n and torch.equal(sh.restore(g), x)
    assert RayShards(307200, 8, default_block(307200, 8)).counts == [38400] * 8        # equal shares when the sizes allow
    H, W = 480, 640
    for world, rb in [(8, 4), (8, None), (3, 8), (7, None), (8, 64)]:
        sh = RayShards.rows(H, W, world, rb)
        for r in range(world):
            row0, n_rows, blk, stride = row_share(H, r, world, rb)
            lr = torch.arange(n_rows)
            y = row0 + (lr // blk) * blk * stride + lr % blk                            # the kernel's row map
            want = (y[:, None] * W + torch.arange(W)[None]).reshape(-1)
            assert sh.counts[r] == n_rows * W and torch.equal(sh.local_index(r), want)


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
            if i == 2:                                                  # rank 1 contributed nothing
                want = (sum(grads[r][i] for r in range(world)) - grads[1][i]) / world
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
        # the debug switch names the silent divergence above; a changed occupancy is announced with set_active_rows
        s3 = GradientSync([table], active_rows={table: 300}, check_inactive_rows=True)
        try:
            s3.sync()
            ok = False
        except RuntimeError as e:
            ok = ok and "active_rows" in str(e)
        table.grad[900] = 0.0
        table.grad[:400] = float(rank + 1)
        s3.set_active_rows(table, 400)
        s3.sync()
        ok = ok and bool((table.grad[:400] == (1 + world) / 2.0).all()) and s3.message_bytes() == [4 * 400 * 24]
        # a gradient that is not contiguous (a transposed view) is still averaged in place, whatever the layout
        pt = torch.nn.Parameter(torch.zeros(6, 4))
        pt.grad = torch.full((4, 6), float(rank + 1)).t()
        ok = ok and not pt.grad.is_contiguous()
        GradientSync([pt]).sync()
        ok = ok and bool((pt.grad == (1 + world) / 2.0).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_sync(world):
    """data-parallel training (the reference's Lightning DDP, train.py:202-220): bucketed gradient averaging incl. the
    voxel table's active-row prefix, at 2 ranks and at the 8 ranks of one MI355X node"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def _bcast_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from object_nerf_amd.distributed import broadcast_parameters, gather_pixel_maps
        g = torch.Generator().manual_seed(100 + rank)                  # every rank starts from DIFFERENT values
        shapes = [(256, 271), (256,), (64, 64), (1000, 24), (3, 128), (1,)]
        params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
        idx_map = torch.randint(-1, 1000, (6, 7, 5), generator=g, dtype=torch.int64)      # a buffer of another dtype
        flag = torch.tensor(rank == 0)                                                   # 0-d bool buffer
        g1 = torch.Generator().manual_seed(101)                         # what rank 1 holds
        want = [torch.randn(*s, generator=g1) for s in shapes]
        want_idx = torch.randint(-1, 1000, (6, 7, 5), generator=g1, dtype=torch.int64)
        moved = broadcast_parameters(params + [idx_map, flag], src=1, bucket_bytes=64 << 10)   # several buckets per dtype
        ok = moved == 4 * sum(p.numel() for p in params) + 8 * idx_map.numel() + 1
        ok = ok and all(torch.equal(p.detach(), w) for p, w in zip(params, want)) and torch.equal(idx_map, want_idx)
        ok = ok and bool(flag.item()) is False and all(p.requires_grad for p in params)
        # a rank that passes another list is told so instead of exchanging garbage
        try:
            broadcast_parameters(params[: (3 if rank == 0 else 2)], src=0)
            ok = False
        except RuntimeError as e:
            ok = ok and "different tensor lists" in str(e)
        # one packed pixel message carries one dtype: a mixed request is refused, not cast
        try:
            gather_pixel_maps({"rgb": torch.zeros(4, 3), "ids": torch.zeros(4, dtype=torch.int64)}, 4 * world)
            ok = False
        except RuntimeError as e:
            ok = ok and "share dtype" in str(e)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_broadcast_parameters(world):
    """the initial synchronisation of data-parallel training (Lightning DDP broadcasts the wrapped module's state from rank 0,
    train.py:261-262): parameters and buffers of every dtype, bucketed, from any source rank"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_broadcast_parameters_without_process_group_is_a_noop():
    from object_nerf_amd.distributed import broadcast_parameters
    p = torch.nn.Parameter(torch.ones(3))
    assert broadcast_parameters([p]) == 0 and torch.equal(p.detach(), torch.ones(3))


def test_gradient_sync_without_process_group_is_a_noop():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradientSync([p]).sync()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


class _Probe(torch.autograd.Function):
    """y = 3 x whose backward records which buckets' exchanges have already been started (the "coarse node" of the test)"""
    seen = None

    @staticmethod
    def forward(ctx, x, sync):
        ctx.sync = sync
        return x * 3.0

    @staticmethod
    def backward(ctx, g):
        _Probe.seen = [w is not None for w in ctx.sync._works]
        return g * 3.0, None


def _attach_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        fine = [torch.nn.Parameter(torch.randn(40, 7, generator=g)), torch.nn.Parameter(torch.randn(40, generator=g))]
        coarse = [torch.nn.Parameter(torch.randn(30, 5, generator=g))]
        shared = [torch.nn.Parameter(torch.randn(12, 24, generator=g))]           # gets a gradient from BOTH nodes (the voxel table)
        x = torch.randn(5, generator=torch.Generator().manual_seed(100 + rank))   # a different batch per rank
        sync = GradientSync(None, groups=[fine, coarse, shared], active_rows={shared[0]: 8}).attach()
        chk = [(0, bool(len(sync.buckets) == 3 and [len(b) for b in sync.buckets] == [2, 1, 1]))]

        def step(skip_fine_bias=False):
            # recorded first -> runs LAST in the backward: the "coarse node"
            yc = _Probe.apply(coarse[0], sync).sum() * x[0] + (shared[0][:8] * x[1]).sum()
            # recorded last -> its backward runs FIRST: the "fine node"
            yf = (fine[0] * x[2]).sum() + (shared[0][:8] * x[3]).sum()
            if not skip_fine_bias:
                yf = yf + (fine[1] * x[4]).sum()
            for p in fine + coarse + shared:
                p.grad = None
            (yc + yf).backward()
        step()
        # when the coarse node ran, the fine bucket was already travelling; the coarse and the shared bucket were not ...
        chk.append((1, bool(_Probe.seen == [True, False, False])))
        # ... and by the end of the backward every bucket is on its way (the shared parameter's gradient is final last)
        chk.append((2, bool([w is not None for w in sync._works] == [True, True, True])))
        sync.sync()
        chk.append((3, bool(sync.early_launches == 3 and all(w is None for w in sync._works))))
        xs = [torch.randn(5, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        mean = lambda k: sum(float(v[k]) for v in xs) / world                   # noqa: E731
        chk.append((4, bool(torch.allclose(fine[0].grad, torch.full((40, 7), mean(2)), atol=1e-6))))
        chk.append((5, bool(torch.allclose(fine[1].grad, torch.full((40,), mean(4)), atol=1e-6))))
        chk.append((6, bool(torch.allclose(coarse[0].grad, torch.full((30, 5), 3.0 * mean(0)), atol=1e-6))))
        chk.append((7, bool(torch.allclose(shared[0].grad[:8], torch.full((8, 24), mean(1) + mean(3)), atol=1e-6))))
        chk.append((8, bool(bool((shared[0].grad[8:] == 0).all()))))
        # a parameter of the first bucket without a gradient on ONE rank: that rank starts nothing early, the collectives still
        # pair up in bucket order on every rank and the missing gradient counts as zeros
        step(skip_fine_bias=(rank == 1))
        early = sum(w is not None for w in sync._works)
        chk.append((9, bool(early == (0 if rank == 1 else 3))))
        sync.sync()
        chk.append((10, bool(torch.allclose(fine[1].grad, torch.full((40,), (sum(float(v[4]) for v in xs) - float(xs[1][4])) / world), atol=1e-6))))
        chk.append((11, bool(torch.allclose(coarse[0].grad, torch.full((30, 5), 3.0 * mean(0)), atol=1e-6))))
        sync.detach()
        q.put((rank, all(c for _, c in chk), [i for i, c in chk if not c]))
    finally:
        dist.destroy_process_group()


def test_gradient_sync_attach_starts_buckets_in_order_during_the_backward():
    """round 6: with the coarse and the fine pass as two autograd nodes, GradientSync.attach() starts the fine model's bucket when
    the fine node's backward has returned -- before the coarse node runs -- and keeps the collectives in bucket order on every
    rank even when a rank has no gradient for some parameter"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_attach_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
