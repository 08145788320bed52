"""Oracle-backed renderer with the call surface bench.py's workloads use (TEST INFRASTRUCTURE).

Two users, both allowed to touch oracle/ (nothing in the product package imports this file):
  * bench.py's `cpu_baseline` leg: times the CPU restatement of the reference's PyTorch path
    (oracle/objnerf_oracle.py, bit-exact with the reference on CPU) on a bounded sample of the SAME workload the GPU
    just rendered, and gives the pixels PSNR is computed against;
  * tests/test_bench_gloo.py: runs bench.py's N > 1 code path (sharding, pixel all-gather, timing, JSON) with
    world_size 2 over gloo on CPU, where the HIP renderer cannot run.

The scene objects are the drop-in modules (plain nn.Modules holding the parameters); the oracle reads their state dicts.
"""
import torch

from oracle import objnerf_oracle as O


def _state(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def _grid(ev):
    return dict(voxel_idx_map=ev.voxel_idx_map.cpu(), table=ev.embedding_space_ftr.weight.detach().cpu(),
                voxel_offset=ev.voxel_offset.cpu(), voxel_size=ev.voxel_size.cpu(), voxel_shape=ev.voxel_shape.cpu())


class OracleRenderer:
    name = "cpu-oracle"
    device = torch.device("cpu")

    def __init__(self):
        self._cache = {}

    def _scene(self, sc):
        key = id(sc)
        if key not in self._cache:
            ev = sc.embeddings["xyz"]
            self._cache[key] = (_state(sc.models["coarse"]), _state(sc.models["fine"]) if "fine" in sc.models else None,
                                _grid(ev) if hasattr(ev, "voxel_idx_map") else None,
                                sc.code_library.embedding_instance.weight.detach().cpu())
        return self._cache[key]

    def render_rays(self, sc, rays, **kw):
        pc, pf, grid, _ = self._scene(sc)
        kw = dict(kw)
        kw["embedding_instance"] = kw["embedding_instance"].cpu()
        if rays.shape[0] == 0:
            # an idle rank of a sharded frame: the reference (and so its restatement) cannot take an empty batch
            # (torch.cat of no chunks, rendering.py:132); the product returns empty maps -- do the same here
            one = torch.tensor([[0.5, 0.5, 0.6, 0.0, 0.0, -1.0, 0.15, 3.0]])
            kw["embedding_instance"] = torch.zeros(1, kw["embedding_instance"].shape[1])
            with torch.no_grad():
                return {k: v[:0] for k, v in O.render_rays(pc, pf, grid, one, **kw).items()}
        with torch.no_grad():
            return O.render_rays(pc, pf, grid, rays.cpu(), **kw)

    def render_rays_multi(self, sc, rays_list, obj_instance_ids, background_skip_bbox=None, **kw):
        pc, pf, grid, table = self._scene(sc)
        boxes = list(background_skip_bbox.values()) if background_skip_bbox else None
        if rays_list[0].shape[0] == 0:                 # idle rank, as above
            one = torch.tensor([[0.5, 0.5, 0.6, 0.0, 0.0, -1.0, 0.15, 3.0]])
            with torch.no_grad():
                r = O.render_rays_multi(pc, pf, grid, table, [one for _ in rays_list], list(obj_instance_ids), skip_boxes=boxes, **kw)
            return {k: v[:0] for k, v in r.items()}
        with torch.no_grad():
            return O.render_rays_multi(pc, pf, grid, table, [r.cpu() for r in rays_list], list(obj_instance_ids),
                                       skip_boxes=boxes, **kw)

    def generate_rays(self, H, W, focal, c2w, near=0.0, far=0.0, box=None, bbox_enlarge=0.0, rows=None):
        c = torch.as_tensor(c2w, dtype=torch.float32)[:3, :4]
        full = O.generate_rays(H, W, focal, c, near, far, box, bbox_enlarge)
        if rows is None:
            return full
        row0, n_rows, blk, stride = rows              # the row map of objnerf_generate_rays_rows (include/objnerf_hip.h)
        lr = torch.arange(n_rows)
        y = row0 + (lr // blk) * blk * stride + lr % blk
        return full.view(H, W, 8)[y].reshape(-1, 8)

    def sync(self):
        pass
