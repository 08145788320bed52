"""Multi-GPU rendering: one process per GPU, rays sharded, pixels all-gathered.

The reference renders on one GPU (its only collective is Lightning's DDP gradient all-reduce,
train.py:261-262).  Rays are fully independent, so a frame shards into contiguous row bands
with NO collective inside the renderer; the single exchange step is one
`all_gather_into_tensor` of the pixel outputs the caller needs (12 B/ray for rgb) -- on 8 MI355X
over xGMI that is <0.5 MB per rank, latency-bound, so it is issued once per frame, never per
chunk, and only for pixels (never the 1 KB/ray weights_/z_vals_ rows).

Training (SURVEY.md §8 row f1 on N GPUs) is plain data parallelism like the reference's Lightning DDP
(train.py:261-262): every rank draws its own ray batch, runs the differentiable `render_rays`, and
`GradientSync` averages the gradients with a few large flat all-reduces.

Backend: torch.distributed "nccl" (= RCCL on ROCm) for GPU tensors; the same functions work on
"gloo" with CPU tensors, which is how tests/test_distributed_gloo.py covers world_size 2 and 8.  A gloo group
handed DEVICE tensors (several ranks sharing one GPU, where RCCL refuses duplicate devices: `bench.py --one-gpu`,
tests/test_gpu_dist.py) stages the message through host memory -- gloo has no device all-gather.
"""
from typing import Dict, Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: ceil(n/world) rays per rank, last ranks may be short."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays)


def shard_rays(rays: torch.Tensor, extras: Dict[str, torch.Tensor] = None, rank: int = None, world: int = None):
    """This rank's slice of the rays and of every per-ray tensor in `extras`."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(rays.shape[0], rank, world)
    ex = {}
    for k, v in (extras or {}).items():
        ex[k] = v[lo:hi] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == rays.shape[0] else v
    return rays[lo:hi], ex


class RayShards:
    """How the n rays (pixels) of one frame are dealt to `world` ranks.

    block=None  contiguous bands of ceil(n/world) rays (`shard_bounds`): right when every ray costs the same
                (`render_rays`: BASELINE configs[3]).
    block=b     blocks of b consecutive rays dealt round-robin (block-cyclic): the cost-balanced split for
                `render_rays_multi` (BASELINE configs[4]) -- rays that miss their object's box are culled before the MLP
                kernel and the hits are spatially clustered (the moved object covers a fifth of the frame), so contiguous
                bands carry very different amounts of work and the single all-gather waits for the heaviest; with blocks
                of a few image rows every rank gets the same share of every region.  Rays stay whole (all K ray sets of a
                pixel on one rank), results come back in frame order through one index_select after the ONE all-gather.
    """

    def __init__(self, n: int, world: int, block: int = None, band: int = None):
        """band: rays per contiguous band when block is None (default ceil(n/world); `rows` passes whole image rows)"""
        self.n, self.world, self.block = int(n), int(world), (None if block is None else int(block))
        if self.block is not None and self.block < 1:
            raise ValueError("RayShards: block must be >= 1")
        if self.block is None:
            self.band = int(band) if band is not None else (self.n + self.world - 1) // self.world
            self.counts = [self._band(r)[1] - self._band(r)[0] for r in range(self.world)]
        else:
            nblk = (self.n + self.block - 1) // self.block
            self.counts = [sum(min(self.block, self.n - b * self.block) for b in range(r, nblk, self.world)) for r in range(self.world)]
        self.per = max(self.counts) if self.counts else 0          # rows per rank in the (padded) collective
        self._idx = {}

    @staticmethod
    def rows(H: int, W: int, world: int, row_block: int = None) -> "RayShards":
        """shards of an H x W frame in whole image rows (what `ray_utils.row_share` / objnerf_generate_rays_rows produce)"""
        if row_block is None:
            return RayShards(H * W, world, band=((H + world - 1) // world) * W)
        return RayShards(H * W, world, block=row_block * W)

    def _band(self, rank):
        lo = min(rank * self.band, self.n)
        return lo, min(lo + self.band, self.n)

    def local_index(self, rank: int, device=None) -> torch.Tensor:
        """frame positions of rank's rays, in the order the rank renders them (cached per device)"""
        key = ("local", rank, str(device))
        if key not in self._idx:
            if self.block is None:
                lo, hi = self._band(rank)
                idx = torch.arange(lo, hi)
            else:
                nblk = (self.n + self.block - 1) // self.block
                idx = torch.cat([torch.arange(b * self.block, min((b + 1) * self.block, self.n)) for b in range(rank, nblk, self.world)]
                                or [torch.zeros(0, dtype=torch.long)])
            self._idx[key] = idx.to(device) if device is not None else idx
        return self._idx[key]

    def take(self, t: torch.Tensor, rank: int) -> torch.Tensor:
        """this rank's rows of a per-ray tensor (n, ...)"""
        if self.block is None:
            lo, hi = self._band(rank)
            return t[lo:hi]
        if self._even():      # whole blocks, the same number for every rank: a strided view, one copy
            return t.reshape(self.n // (self.block * self.world), self.world, self.block, *t.shape[1:])[:, rank].reshape(-1, *t.shape[1:])
        return t.index_select(0, self.local_index(rank, t.device))

    def _even(self):
        return self.block is not None and self.n > 0 and self.n % (self.block * self.world) == 0

    def restore(self, gathered: torch.Tensor) -> torch.Tensor:
        """(world * per, C) all-gather result (rank r's rows at [r*per, r*per + counts[r])) -> (n, C) in frame order"""
        if self.block is None and self.band == self.per:
            return gathered[: self.n]                        # bands are already in frame order: only the tail is padding
        if self._even():      # rank-major blocks -> frame order: one strided copy instead of an index_select
            g = gathered.reshape(self.world, self.per // self.block, self.block, *gathered.shape[1:])
            return g.transpose(0, 1).reshape(self.n, *gathered.shape[1:])
        key = ("inverse", str(gathered.device))
        if key not in self._idx:
            inv = torch.empty(self.n, dtype=torch.long)
            for r in range(self.world):
                inv[self.local_index(r)] = r * self.per + torch.arange(self.counts[r])
            self._idx[key] = inv.to(gathered.device)
        return gathered.index_select(0, self._idx[key])


def default_block(n: int, world: int) -> int:
    """block-cyclic granularity when the caller gives none: ~64 blocks per rank, a multiple of 64 rays, and -- when possible --
    a block count divisible by the world size so that every rank gets the same number of rays"""
    b = max(64, (n // (world * 64)) // 64 * 64)
    for cand in range(b, max(63, b // 2), -64):
        if n % cand == 0 and (n // cand) % world == 0:
            return cand
    return b


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """device tensor on a gloo group: the collective runs on a host copy"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_into(out: torch.Tensor, send: torch.Tensor, group=None) -> None:
    if _host_staged(send, group):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, send.cpu(), group=group)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, send, group=group)


def gather_pixels(local: torch.Tensor, n_total: int = None, shards: RayShards = None) -> torch.Tensor:
    """All-gathers per-ray pixel rows.  `local` is (n_local, C) (or (n_local,)); short / empty shards are padded to the
    longest one for the collective and the padding dropped afterwards.  Returns (n_total, C) in frame order."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    squeeze = local.dim() == 1
    cols = 1
    for d in local.shape[1:]:
        cols *= int(d)
    x = local.reshape(local.shape[0], cols)                  # explicit width: reshape(0, -1) is ambiguous
    if shards is None and n_total is not None:
        shards = RayShards(n_total, world)
    per = x.shape[0] if shards is None else shards.per
    if x.shape[0] == per and x.is_contiguous():
        send = x
    else:
        send = x.new_zeros(per, cols)
        send[: x.shape[0]].copy_(x)
    out = x.new_empty(world * per, cols)
    _all_gather_into(out, send)
    if shards is not None:
        out = shards.restore(out)
    return out.reshape(-1) if squeeze else out


def gather_pixel_maps(local: Dict[str, torch.Tensor], n_total: int = None, shards: RayShards = None) -> Dict[str, torch.Tensor]:
    """All-gathers several per-ray maps ((n_local,) or (n_local, C) each) in ONE collective: the maps are packed side by
    side into one (per, sum C) message -- rgb + depth + opacity = 20 B/ray -- because on xGMI a ring all-gather is
    latency-bound at this size (<1 MB per rank) and three launches cost three latencies.  The message buffer is
    allocated at its padded size and the maps are copied straight into its columns (no cat, no pad pass).
    Returns {key: (n_total, ...)} in frame order."""
    keys = list(local)
    if not dist.is_initialized() or not keys:
        return dict(local)
    world = dist.get_world_size()
    cols = [int(torch.Size(local[k].shape[1:]).numel()) for k in keys]          # 1 for (n_local,) maps
    first = local[keys[0]]
    n_local = first.shape[0]
    for k in keys:          # one packed message: a map of another dtype / device would be cast or copied silently
        if local[k].dtype != first.dtype or local[k].device != first.device or local[k].shape[0] != n_local:
            raise RuntimeError("gather_pixel_maps: map %r is %s on %s with %d rows, %r is %s on %s with %d rows -- the maps of "
                               "one message share dtype, device and row count (gather other dtypes in a call of their own)"
                               % (k, local[k].dtype, local[k].device, local[k].shape[0], keys[0], first.dtype, first.device, n_local))
    if shards is None and n_total is not None:
        shards = RayShards(n_total, world)
    per = n_local if shards is None else shards.per
    packed = first.new_empty(per, sum(cols))
    off = 0
    for k, c in zip(keys, cols):
        packed[:n_local, off:off + c].copy_(local[k].reshape(n_local, c))
        off += c
    if n_local < per:
        packed[n_local:].zero_()
    full = first.new_empty(world * per, packed.shape[1])
    _all_gather_into(full, packed)
    if shards is not None:
        full = shards.restore(full)
    out, off = {}, 0
    for k, c in zip(keys, cols):
        piece = full[:, off:off + c]
        out[k] = piece.reshape(-1) if local[k].dim() == 1 else piece.reshape(full.shape[0], *local[k].shape[1:])
        off += c
    return out


def _rank_world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def render_rays_sharded(render_fn, rays: torch.Tensor, per_ray: Dict[str, torch.Tensor] = None,
                        gather_keys: Iterable[str] = ("rgb_fine", "depth_fine", "opacity_fine"), on_rendered=None,
                        shards: RayShards = None, as_rank: Tuple[int, int] = None, **kwargs):
    """Renders this rank's share of `rays` with `render_fn(rays=..., **per_ray_slices, **kwargs)` and returns
    {key: full-frame tensor} for `gather_keys` (identical on every rank); one collective per call.  Default split:
    contiguous bands (every ray of `render_rays` costs the same).
    on_rendered(local_results): optional hook between the render and the collective (bench.py records an event there).
    as_rank = (rank, world): render that rank's share without a process group (tools/band_replay.py replays the ranks of
    an N-GPU run one after the other on one GPU); the local maps are returned, nothing is gathered."""
    n = rays.shape[0]
    rank, world = as_rank if as_rank is not None else _rank_world()
    sh = shards if shards is not None else RayShards(n, world)
    ex = {}
    for k, v in (per_ray or {}).items():
        ex[k] = sh.take(v, rank) if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v
    res = render_fn(rays=sh.take(rays, rank), **ex, **kwargs)
    if on_rendered is not None:
        on_rendered(res)
    local = {k: res[k] for k in gather_keys if k in res}
    return local if as_rank is not None else gather_pixel_maps(local, n, sh)


def render_rays_multi_sharded(render_fn, rays_list, gather_keys: Iterable[str] = ("rgb_fine", "depth_fine", "opacity_fine"),
                              on_rendered=None, shards: RayShards = None, rays_are_local: bool = False,
                              as_rank: Tuple[int, int] = None, **kwargs):
    """The same for `render_rays_multi` (BASELINE configs[4], the editing demo on N GPUs): every ray set of the list
    describes the same pixels through a different object transform, so all sets are dealt with the same `RayShards` and
    each rank composites its pixels over all sets; one pixel all-gather per frame.  Default split: block-cyclic
    (`default_block`), because the culled object ray sets make the cost per pixel non-uniform (see RayShards).
    rays_are_local: `rays_list` already holds only this rank's rays in `shards.local_index(rank)` order (each rank
    generated its own rows, `ray_utils.generate_rays(rows=ray_utils.row_share(...))`); `shards` is then required.
    as_rank: as in render_rays_sharded."""
    rank, world = as_rank if as_rank is not None else _rank_world()
    if rays_are_local:
        if shards is None:
            raise RuntimeError("render_rays_multi_sharded: rays_are_local needs the RayShards the rays were generated for")
        n, sh = shards.n, shards
        if any(r.shape[0] != sh.counts[rank] for r in rays_list):
            raise RuntimeError("render_rays_multi_sharded: a local ray set does not have this rank's %d rays" % sh.counts[rank])
        local = list(rays_list)
    else:
        n = rays_list[0].shape[0]
        if any(r.shape[0] != n for r in rays_list):
            raise RuntimeError("render_rays_multi_sharded: every ray set must have the same number of rays")
        sh = shards if shards is not None else RayShards(n, world, default_block(n, world) if world > 1 else None)
        local = [sh.take(r, rank) for r in rays_list]
    res = render_fn(rays_list=local, **kwargs)
    if on_rendered is not None:
        on_rendered(res)
    maps = {k: res[k] for k in gather_keys if k in res}
    return maps if as_rank is not None else gather_pixel_maps(maps, n, sh)


@torch.no_grad()
def broadcast_parameters(tensors: Iterable[torch.Tensor], src: int = 0, group=None, bucket_bytes: int = 64 << 20) -> int:
    """Every rank's tensors become rank `src`'s, in place: the initial synchronisation of data-parallel training (what
    Lightning's DDP does when it wraps the module, train.py:261-262 -- `GradientSync` keeps replicas equal only if they
    START equal).  Pass parameters AND buffers (`list(m.parameters()) + list(m.buffers())`: the voxel index map and the
    grid geometry are buffers).  Like the gradient exchange: few large messages -- tensors are packed per dtype into flat
    buckets of up to `bucket_bytes`, one broadcast each, all in flight before the first wait.  Every rank must pass
    tensors of the same shapes and dtypes in the same order (checked: a one-number digest of the layout is compared
    first).  No-op without a process group or in a 1-rank group.  Returns the number of bytes that travelled."""
    ts = [t for t in tensors]
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not ts:
        return 0
    # layout digest: ranks that disagree on the list would otherwise exchange garbage (or hang on message sizes)
    import zlib
    desc = ";".join("%s:%s" % (str(t.dtype), "x".join(str(int(d)) for d in t.shape)) for t in ts).encode()
    ddev = "cpu" if dist.get_backend(group) == "gloo" else ts[0].device       # RCCL moves device tensors only
    mine = torch.tensor([zlib.crc32(desc), len(ts)], dtype=torch.int64, device=ddev)
    ref = mine.clone()
    dist.broadcast(ref, src=src, group=group)
    agree = (ref == mine).all().to(torch.int64).reshape(1)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=group)
    if int(agree.item()) != 1:
        raise RuntimeError("broadcast_parameters: the ranks pass different tensor lists (count / shapes / dtypes)")
    by_dtype: Dict[torch.dtype, List[int]] = {}
    for i, t in enumerate(ts):
        by_dtype.setdefault(t.dtype, []).append(i)
    pending, total = [], 0
    for dt, idx in by_dtype.items():
        cur, cur_bytes, buckets = [], 0, []
        for i in idx:
            nb = ts[i].numel() * ts[i].element_size()
            if cur and cur_bytes + nb > bucket_bytes:
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += nb
        if cur:
            buckets.append(cur)
        for b in buckets:
            dev = ts[b[0]].device
            staged = _host_staged(ts[b[0]], group)
            flat = torch.empty(sum(ts[i].numel() for i in b), dtype=dt, device="cpu" if staged else dev)
            off = 0
            for i in b:
                n = ts[i].numel()
                flat[off:off + n].copy_(ts[i].reshape(-1))
                off += n
            total += flat.numel() * flat.element_size()
            pending.append((dist.broadcast(flat, src=src, group=group, async_op=True), flat, b))
    for work, flat, b in pending:
        work.wait()
        off = 0
        for i in b:
            n = ts[i].numel()
            ts[i].copy_(flat[off:off + n].view(ts[i].shape))
            off += n
    return total


class GradientSync:
    """Averages `.grad` of a fixed parameter list over the process group: the collective behind data-parallel
    training (what Lightning's DDP does for the reference, train.py:261-262).

    The differentiable `render_rays` is ONE autograd node, so all gradients of a step appear together at the end of
    its backward: there is nothing to overlap bucket by bucket, and the right shape for xGMI (point-to-point links,
    ring all-reduce bound by the per-link rate, ~10 us of latency per launch) is few, large messages.  Parameters are
    therefore packed, in order, into flat fp32 buckets of up to `bucket_bytes` (default 64 MiB: both MLPs and the
    codes share one bucket, the 77 MB voxel table gets its own), each reduced by one asynchronous all-reduce; the
    reduces are all in flight before the first is waited for.  Parameters whose gradient is None on this rank
    (e.g. an object code no local ray used) contribute zeros, so ranks never disagree on the message layout.
    """

    def __init__(self, params: Sequence[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None,
                 active_rows: Dict[torch.nn.Parameter, int] = None, reduce_at_world1: bool = False,
                 check_inactive_rows: bool = False, groups: Sequence[Sequence[torch.nn.Parameter]] = None):
        """active_rows: {parameter: n} -- only the first n rows of that (2-D) parameter can ever receive a gradient, so only
        `grad[:n]` travels.  This is the sparse handling of the voxel feature table: the renderer only
        reads rows the index map points at, i.e. rows < number of occupied voxels (`EmbeddingVoxel.active_rows()`); the
        reference's `max_voxels` = 800000 rows (76.8 MB of gradient) are mostly padding -- 103.6k rows = 9.9 MB in the
        ScanNet-like scene, 10.8k rows = 1 MB in the ToyDesk-2-like one.  Contiguous prefix: no index exchange, no
        host synchronisation.  reduce_at_world1: issue the collectives even in a 1-rank group (tests of the RCCL path on
        one GPU); by default a 1-rank sync is a no-op.  check_inactive_rows: debug switch -- every sync asserts that the rows
        beyond the active prefix carry no gradient (a host synchronisation per parameter: off by default); call
        `set_active_rows` after the voxel index map changes (e.g. a checkpoint with another occupancy was loaded)."""
        self.check_inactive_rows = bool(check_inactive_rows)
        # groups (round 6): parameter lists in the order their gradients become final during the backward -- with the two-node
        # differentiable render_rays: [fine model], [coarse model], [codes, voxel table].  Buckets never straddle a group, and the
        # bucket ORDER is the exchange order on every rank (see attach()).  `params` may then be None.
        self._group_ends = None
        if groups is not None:
            flat = [p for g_ in groups for p in g_ if p.requires_grad]
            if params is not None and {id(p) for p in params if p.requires_grad} != {id(p) for p in flat}:
                raise ValueError("GradientSync: `groups` must partition `params`")
            ends, c = set(), 0
            for g_ in groups:
                c += sum(1 for p in g_ if p.requires_grad)
                ends.add(c)
            self._group_ends = ends
            params = flat
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.reduce_at_world1 = bool(reduce_at_world1)
        ar = {id(p): int(n) for p, n in (active_rows or {}).items()}
        self.rows: List[int] = []                    # rows of params[i] that travel (-1 = the whole tensor)
        for p in self.params:
            n = ar.get(id(p), -1)
            if n >= 0 and (p.dim() != 2 or n > p.shape[0]):
                raise ValueError("active_rows: parameter must be 2-D with at least that many rows")
            self.rows.append(n)
        self.buckets: List[List[int]] = []          # indices into self.params
        cur, cur_bytes = [], 0
        for i, p in enumerate(self.params):
            nbytes = self._numel(i) * 4
            if cur and (cur_bytes + nbytes > bucket_bytes or (self._group_ends is not None and i in self._group_ends)):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat: List[torch.Tensor] = [None] * len(self.buckets)
        self._hooks = []
        self._works: List = [None] * len(self.buckets)       # attach(): the all-reduce of bucket b once it has been started
        self._ready = [0] * len(self.buckets)
        self._bucket_of = {}
        for b, idx in enumerate(self.buckets):
            for i in idx:
                self._bucket_of[i] = b
        self.early_launches = 0                              # buckets whose exchange started from a gradient hook (last step)

    def set_active_rows(self, param: torch.nn.Parameter, n: int) -> None:
        """new travelling prefix of `param` (same value on every rank); the flat buffers are re-sized on the next sync"""
        i = next(j for j, p in enumerate(self.params) if p is param)
        if n < 0 or n > param.shape[0]:
            raise ValueError("active_rows: parameter has %d rows" % param.shape[0])
        self.rows[i] = int(n)
        self._flat = [None] * len(self.buckets)

    def _numel(self, i: int) -> int:
        p = self.params[i]
        return p.numel() if self.rows[i] < 0 else self.rows[i] * p.shape[1]

    def message_bytes(self) -> List[int]:
        """bytes of each all-reduce message"""
        return [4 * sum(self._numel(i) for i in idx) for idx in self.buckets]

    def _buffer(self, b: int) -> torch.Tensor:
        idx = self.buckets[b]
        n = sum(self._numel(i) for i in idx)
        dev = self.params[idx[0]].device
        if _host_staged(self.params[idx[0]], self.group):      # several ranks on one GPU over gloo: the message lives on the host
            dev = torch.device("cpu")
        if self._flat[b] is None or self._flat[b].device != dev or self._flat[b].numel() != n:
            self._flat[b] = torch.empty(n, dtype=torch.float32, device=dev)
        return self._flat[b]

    def _travelling(self, i: int, t: torch.Tensor) -> torch.Tensor:
        """the part of params[i]-shaped tensor t that is exchanged, flattened"""
        return t.reshape(-1) if self.rows[i] < 0 else t[: self.rows[i]].reshape(-1)

    def attach(self) -> "GradientSync":
        """Start each bucket's all-reduce AS SOON AS its gradients are final instead of at sync(): a post-accumulate-grad hook per
        parameter counts the bucket's arrivals.  With the two-node differentiable render_rays (object_nerf_amd/autograd.py) the fine
        model's gradients arrive when the fine node's backward returns, i.e. their exchange travels while the coarse node's
        backward runs -- what torch's DDP reducer does for the reference (train.py:261-262).  Buckets are started strictly in
        bucket order (bucket b waits for b - 1), so every rank issues the same collectives in the same order whatever arrives
        when; a bucket with a parameter that received no gradient on this rank is started by sync().  sync() stays mandatory: it
        starts what is left, waits, averages and writes the gradients back."""
        if self._hooks:
            return self
        for i, p in enumerate(self.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i)))
        return self

    def detach(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _active(self) -> bool:
        if not dist.is_initialized():
            return False
        return dist.get_world_size(self.group) > 1 or self.reduce_at_world1

    def _on_grad(self, i: int) -> None:
        if not self._active():
            return
        b = self._bucket_of[i]
        self._ready[b] += 1
        # in order: start every complete bucket whose predecessors have all been started
        nb = 0
        while nb < len(self.buckets) and self._works[nb] is not None:
            nb += 1
        while nb < len(self.buckets) and self._ready[nb] >= len(self.buckets[nb]):
            self._start(nb)
            self.early_launches += 1
            nb += 1

    @torch.no_grad()
    def _start(self, b: int) -> None:
        """pack bucket b's gradients into its flat message and issue the asynchronous all-reduce"""
        idx = self.buckets[b]
        flat = self._buffer(b)
        off = 0
        for i in idx:
            p = self.params[i]
            n = self._numel(i)
            view = flat[off:off + n]
            if p.grad is None:
                view.zero_()
            else:
                view.copy_(self._travelling(i, p.grad))
                if self.check_inactive_rows and self.rows[i] >= 0 and bool(p.grad[self.rows[i]:].any()):
                    raise RuntimeError("GradientSync: a row beyond active_rows = %d received a gradient (the voxel index "
                                       "map changed after construction?): ranks would diverge" % self.rows[i])
            off += n
        self._works[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @torch.no_grad()
    def sync(self) -> None:
        """In place: every rank's `.grad` becomes the mean over ranks.  No-op without a process group."""
        if not self._active():
            return
        world = dist.get_world_size(self.group)
        started_early = sum(w is not None for w in self._works)
        for b in range(len(self.buckets)):          # whatever the hooks have not started (all of it without attach())
            if self._works[b] is None:
                self._start(b)
        works = self._works
        inv = 1.0 / world
        for b, idx in enumerate(self.buckets):
            works[b].wait()
            flat = self._flat[b]
            off = 0
            for i in idx:
                p = self.params[i]
                n = self._numel(i)
                g = flat[off:off + n]
                if g.device != p.device:
                    g = g.to(p.device)
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                if p.grad.is_contiguous():
                    torch.mul(g, inv, out=self._travelling(i, p.grad))     # a view of p.grad: written in place
                elif self.rows[i] < 0:
                    p.grad.copy_((g * inv).view_as(p.grad))                # any layout: reshape() would have been a copy
                else:
                    p.grad[: self.rows[i]].copy_((g * inv).view(self.rows[i], -1))
                off += n
        self._works = [None] * len(self.buckets)
        self._ready = [0] * len(self.buckets)
        self.early_launches = started_early
