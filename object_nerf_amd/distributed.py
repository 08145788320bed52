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
"gloo" with CPU tensors, which is how tests/test_distributed_gloo.py covers world_size 2.
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


def gather_pixels(local: torch.Tensor, n_total: int = None) -> torch.Tensor:
    """All-gathers per-ray pixel rows.  `local` is (n_local, C) (or (n_local,)); every rank must
    hold the same n_local except possibly trailing short/empty shards, which are padded to
    ceil(n_total/world) for the collective and trimmed afterwards.  Returns (n_total, C)."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    squeeze = local.dim() == 1
    cols = 1
    for d in local.shape[1:]:
        cols *= int(d)
    x = local.reshape(local.shape[0], cols).contiguous()     # explicit width: reshape(0, -1) is ambiguous
    per = x.shape[0] if n_total is None else (n_total + world - 1) // world
    if x.shape[0] < per:
        x = torch.cat([x, x.new_zeros(per - x.shape[0], x.shape[1])], 0)
    out = x.new_empty(world * per, x.shape[1])
    dist.all_gather_into_tensor(out, x)
    if n_total is not None:
        out = out[:n_total]
    return out.reshape(-1) if squeeze else out


def gather_pixel_maps(local: Dict[str, torch.Tensor], n_total: int = None) -> Dict[str, torch.Tensor]:
    """All-gathers several per-ray maps ((n_local,) or (n_local, C) each) in ONE collective: the maps are packed side by
    side into one (n_local, sum C) message -- rgb + depth + opacity = 20 B/ray -- because on xGMI a ring all-gather is
    latency-bound at this size (<1 MB per rank) and three launches cost three latencies.  Returns {key: (n_total, ...)}."""
    keys = list(local)
    if not dist.is_initialized() or not keys:
        return dict(local)
    cols = [int(torch.Size(local[k].shape[1:]).numel()) for k in keys]          # 1 for (n_local,) maps
    n_local = local[keys[0]].shape[0]
    packed = torch.cat([local[k].reshape(n_local, c) for k, c in zip(keys, cols)], 1) if len(keys) > 1 else \
        local[keys[0]].reshape(n_local, cols[0])
    full = gather_pixels(packed, n_total)
    out, off = {}, 0
    for k, c in zip(keys, cols):
        piece = full[:, off:off + c]
        out[k] = piece.reshape(-1) if local[k].dim() == 1 else piece.reshape(full.shape[0], *local[k].shape[1:])
        off += c
    return out


def render_rays_sharded(render_fn, rays: torch.Tensor, per_ray: Dict[str, torch.Tensor] = None,
                        gather_keys: Iterable[str] = ("rgb_fine", "depth_fine", "opacity_fine"), on_rendered=None, **kwargs):
    """Renders this rank's band of `rays` with `render_fn(rays=..., **per_ray_slices, **kwargs)`
    and returns {key: full-frame tensor} for `gather_keys` (identical on every rank); one collective per call.
    on_rendered(local_results): optional hook between the render and the collective (bench.py records an event there)."""
    n = rays.shape[0]
    r_local, ex = shard_rays(rays, per_ray)
    res = render_fn(rays=r_local, **ex, **kwargs)
    if on_rendered is not None:
        on_rendered(res)
    return gather_pixel_maps({k: res[k] for k in gather_keys if k in res}, n)


def render_rays_multi_sharded(render_fn, rays_list, gather_keys: Iterable[str] = ("rgb_fine", "depth_fine", "opacity_fine"),
                              on_rendered=None, **kwargs):
    """The same for `render_rays_multi` (BASELINE configs[4], the editing demo on N GPUs): every ray set of the list
    describes the same pixels through a different object transform, so all sets are cut at the same row bounds and
    each rank composites its band of pixels over all sets; one pixel all-gather per frame, as above."""
    n = rays_list[0].shape[0]
    if any(r.shape[0] != n for r in rays_list):
        raise RuntimeError("render_rays_multi_sharded: every ray set must have the same number of rays")
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(n, rank, world)
    res = render_fn(rays_list=[r[lo:hi] for r in rays_list], **kwargs)
    if on_rendered is not None:
        on_rendered(res)
    return gather_pixel_maps({k: res[k] for k in gather_keys if k in res}, n)


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
                 check_inactive_rows: bool = False):
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
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat: List[torch.Tensor] = [None] * len(self.buckets)

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
        if self._flat[b] is None or self._flat[b].device != dev or self._flat[b].numel() != n:
            self._flat[b] = torch.empty(n, dtype=torch.float32, device=dev)
        return self._flat[b]

    def _travelling(self, i: int, t: torch.Tensor) -> torch.Tensor:
        """the part of params[i]-shaped tensor t that is exchanged, flattened"""
        return t.reshape(-1) if self.rows[i] < 0 else t[: self.rows[i]].reshape(-1)

    @torch.no_grad()
    def sync(self) -> None:
        """In place: every rank's `.grad` becomes the mean over ranks.  No-op without a process group."""
        if not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1 and not self.reduce_at_world1:
            return
        works = []
        for b, idx in enumerate(self.buckets):
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
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        inv = 1.0 / world
        for b, idx in enumerate(self.buckets):
            works[b].wait()
            flat = self._flat[b]
            off = 0
            for i in idx:
                p = self.params[i]
                n = self._numel(i)
                g = flat[off:off + n]
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                if p.grad.is_contiguous():
                    torch.mul(g, inv, out=self._travelling(i, p.grad))     # a view of p.grad: written in place
                elif self.rows[i] < 0:
                    p.grad.copy_((g * inv).view_as(p.grad))                # any layout: reshape() would have been a copy
                else:
                    p.grad[: self.rows[i]].copy_((g * inv).view(self.rows[i], -1))
                off += n
