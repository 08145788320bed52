"""Drop-in `Embedding` / `EmbeddingVoxel` (reference: models/embedding_helper.py:40-74, 77-479).

`forward` of both types runs on HIP kernels (csrc/ray_kernels.hip: pos_encode_kernel,
voxel_embed_kernel).  Inside `render_rays` these modules are NOT called at all: the renderer
hands the grid state (`EmbeddingVoxel.grid_struct()`) to the fused kernel, which embeds in
registers.  The stand-alone forwards exist because callers use them directly
(tools/extract_mesh.py:85-108) and for stage-level parity tests.

State kept bit-compatible with the reference so checkpoints round-trip:
buffers voxel_size, bounds, voxel_offset, voxel_shape, voxel_count, voxel_occupancy,
voxel_idx_map (int64) and the parameter embedding_space_ftr.weight.

Out of scope: progressive-training utilities (self_pruning_empty_voxels,
voxel_subdivision), the unused dense/ray-box helpers.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from . import _lib


class Embedding(nn.Module):
    """(x, sin(2^k x), cos(2^k x), ...), reference embedding_helper.py:40-74."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        self.logscale = bool(logscale)
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
        # plain attribute, like the reference (embedding_helper.py:52-55)
        if logscale:
            self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)
        else:
            self.freq_bands = torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)
        self._bands_dev = None

    @_lib.on_device_of(lambda self, x: x)
    def forward(self, x):
        _lib.require_cuda(x, "Embedding input")
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("object_nerf_amd.Embedding is forward-only (no PyTorch fallback)")
        shp = x.shape
        c = shp[-1]
        xf = _lib.as_f32(x).reshape(-1, c)
        out = torch.empty(xf.shape[0], c * (2 * self.N_freqs + 1), dtype=torch.float32, device=x.device)
        bands = None
        # (modules pickled whole before `logscale` / `_bands_dev` existed have neither attribute: they are 2^k modules)
        if not getattr(self, "logscale", True):        # the bands travel as a device table; 2^k bands are generated in the kernel
            cached = getattr(self, "_bands_dev", None)
            if cached is None or cached.device != x.device:
                cached = self._bands_dev = self.freq_bands.to(torch.float32).to(x.device).contiguous()
            bands = cached
        _lib.check(_lib.lib().objnerf_pos_encode_freqs(_lib.ptr(xf), xf.shape[0], c, self.N_freqs, _lib.ptr(bands), _lib.ptr(out),
                                                       _lib.stream_ptr()), "pos_encode")
        return out.reshape(*shp[:-1], out.shape[-1])


def build_voxel_state(pcd_xyz_world, scene_center, scale_factor, voxel_size_world, neighbor_marks):
    """Point cloud -> (voxel_size, bounds, voxel_offset, voxel_shape, voxel_occupancy, voxel_idx_map).

    One-off CPU setup restating EmbeddingVoxel.set_pointclouds + generate_voxel_idx_map
    (embedding_helper.py:86-200): normalise, quantise with round(), mark a
    neighbor_marks^3 neighbourhood (the reference uses an all-ones Conv3d with zero padding; a
    max-pool over the same window is the same boolean), enumerate occupied voxels in
    torch.nonzero (row-major) order.
    """
    pts = torch.as_tensor(np.asarray(pcd_xyz_world), dtype=torch.float64)
    pts = ((pts - torch.as_tensor(np.asarray(scene_center), dtype=torch.float64)) / scale_factor).float()
    voxel_size = torch.scalar_tensor(voxel_size_world / scale_factor)
    bounds = torch.stack([pts.min(dim=0)[0], pts.max(dim=0)[0]])
    voxel_offset = -bounds[0]
    voxel_shape = torch.tensor([int(((bounds[1][i] - bounds[0][i]) / voxel_size).int().item()) + 3 for i in range(3)])
    occ = torch.zeros(tuple(voxel_shape.tolist()), dtype=torch.bool)
    q = ((pts + voxel_offset) / voxel_size).round().long()
    bad = ((q < 0).sum(1) > 0) | ((q >= voxel_shape).sum(1) > 0)
    q = q[~bad]
    occ[q[:, 0], q[:, 1], q[:, 2]] = True
    k = int(neighbor_marks)
    pad = (k - 1) // 2
    dil = torch.nn.functional.max_pool3d(occ[None, None].float(), kernel_size=k, stride=1, padding=pad)
    occ = dil[0, 0] > 0
    if tuple(occ.shape) != tuple(voxel_shape.tolist()):
        raise RuntimeError("neighbor_marks must be odd (the reference asserts the shape is unchanged)")
    idx_map = -torch.ones(tuple(voxel_shape.tolist()), dtype=torch.long)
    nz = torch.nonzero(occ)
    idx_map[nz[:, 0], nz[:, 1], nz[:, 2]] = torch.arange(nz.shape[0])
    return voxel_size, bounds, voxel_offset, voxel_shape, occ, idx_map


class EmbeddingVoxel(nn.Module):
    def __init__(self, channels, N_freqs, max_voxels, dataset_extra_config):
        super().__init__()
        if channels <= 8 or N_freqs < 0:
            raise ValueError("EmbeddingVoxel: channels = scene channels + 8 object channels (instance_ftr_C, embedding_helper.py)")
        # 16 + 8 channels with 6 frequencies is what the fused kernels embed in registers; other sizes are embedded by the
        # generic kernels (objnerf_voxel_features + objnerf_pos_encode_block) and rendered on the layer-wise path
        self.fused_layout = (channels == 24 and N_freqs == 6)
        self.N_freqs = int(N_freqs)
        self.embedding_final = Embedding(channels, N_freqs)
        self.embedding_space_ftr = nn.Embedding(max_voxels, channels)
        self.set_pointclouds(dataset_extra_config)
        self.channels = channels
        self.embedding_xyz_classical = Embedding(3, 10)
        self._idx32 = None
        self._idx32_key = None

    def set_pointclouds(self, dataset_extra_config):
        self.conf = dataset_extra_config
        conf = dataset_extra_config
        if "pcd_xyz" in conf:            # synthetic / in-memory cloud (tests, bench)
            pcd_xyz = np.asarray(conf["pcd_xyz"])
        else:                            # embedding_helper.py:88-94
            import open3d as o3d
            pcd_xyz = np.asarray(o3d.io.read_point_cloud(conf["pcd_path"]).points)
        vs, bounds, off, shape, occ, idx_map = build_voxel_state(
            pcd_xyz, conf["scene_center"], conf["scale_factor"], conf["voxel_size"], conf["neighbor_marks"])
        if int(occ.sum()) > self.embedding_space_ftr.num_embeddings:
            raise RuntimeError("more occupied voxels than max_voxels")   # embedding_helper.py:196
        self.register_buffer("voxel_size", vs)
        self.register_buffer("bounds", bounds)
        self.register_buffer("voxel_offset", off)
        self.register_buffer("voxel_shape", shape)
        self.register_buffer("voxel_count", torch.scalar_tensor(shape.prod()))
        self.register_buffer("voxel_occupancy", occ)
        self.register_buffer("voxel_idx_map", idx_map)
        self.instance_ftr_C = 8

    # ---- device view for the kernels --------------------------------------------------------
    def grid_struct(self):
        """objnerf_voxel_grid for the current device state.  The int32 copy of the int64 index map and the host copies
        of the grid scalars are cached per buffer version (all four buffers are part of the key); the fp32-contiguous
        view of the table is kept on `self` so that a converted copy (a .half()/.double() table) outlives the launch
        that reads it.  `invalidate_grid_cache()` after in-place writes through `.data` (they do not bump `_version`)."""
        m = self.voxel_idx_map
        _lib.require_cuda(m, "EmbeddingVoxel buffers")
        key = tuple((b.data_ptr(), b._version) for b in (m, self.voxel_shape, self.voxel_offset, self.voxel_size))
        if self._idx32 is None or key != self._idx32_key:
            self._idx32 = m.to(torch.int32).contiguous()
            self._idx32_key = key
            # scalars are read back once per buffer version, not per call
            self._host = (tuple(int(v) for v in self.voxel_shape.tolist()),
                          tuple(float(v) for v in self.voxel_offset.tolist()), float(self.voxel_size.item()))
        table = self.embedding_space_ftr.weight
        self._table32 = _lib.as_f32(table.detach())      # same storage for an fp32 contiguous table, else a kept copy
        g = _lib.VoxelGrid()
        g.idx_map = self._idx32.data_ptr()
        g.table = self._table32.data_ptr()
        shape, off, vs = self._host
        if tuple(self._idx32.shape) != shape:
            raise RuntimeError("voxel_idx_map shape does not match voxel_shape")
        for i in range(3):
            g.shape[i] = shape[i]
            g.offset[i] = off[i]
        g.voxel_size = vs
        g.n_rows = table.shape[0]
        return g

    def active_rows(self):
        """Rows of embedding_space_ftr.weight the index map can point at (= occupied voxels, enumerated from 0 by
        generate_voxel_idx_map, embedding_helper.py:187-200): only these rows are read by the renderer and only they
        can receive a gradient -- distributed.GradientSync(active_rows=...) exchanges just this prefix."""
        return int(self.voxel_idx_map.max().item()) + 1

    def invalidate_grid_cache(self):
        self._idx32 = None
        self._idx32_key = None

    @_lib.on_device_of(lambda self, xyz: xyz)
    def forward(self, xyz):
        _lib.require_cuda(xyz, "EmbeddingVoxel input")
        if torch.is_grad_enabled() and (xyz.requires_grad or self.embedding_space_ftr.weight.requires_grad):
            raise NotImplementedError(
                "object_nerf_amd.EmbeddingVoxel.forward is an inference entry point (the differentiable path is render_rays, "
                "object_nerf_amd/autograd.py, which embeds inside the kernel); call it under torch.no_grad()")
        x = _lib.as_f32(xyz).reshape(-1, 3)
        n = x.shape[0]
        g = self.grid_struct()
        if not self.fused_layout:
            # any channel count / frequency count: raw trilinear features, then each positional encoding written straight
            # into its column block of cat([PE(scene), PE10(xyz)]) and PE(object) (embedding_helper.py:325-329, 403-409)
            l = _lib.lib()
            C_, F = self.channels, self.N_freqs
            cs, co = C_ - self.instance_ftr_C, self.instance_ftr_C
            raw = torch.empty(n, C_, dtype=torch.float32, device=x.device)
            scene = torch.empty(n, cs * (2 * F + 1) + 63, dtype=torch.float32, device=x.device)
            obj = torch.empty(n, co * (2 * F + 1), dtype=torch.float32, device=x.device)
            if n == 0:
                return scene, obj
            sp = _lib.stream_ptr()
            _lib.check(l.objnerf_voxel_features(C.byref(g), C_, _lib.ptr(x), n, _lib.ptr(raw), C_, sp), "voxel_features")
            _lib.check(l.objnerf_pos_encode_block(C.c_void_p(raw.data_ptr()), C_, n, cs, F, None, C.c_void_p(scene.data_ptr()),
                                                  scene.shape[1], sp), "pos_encode_block")
            _lib.check(l.objnerf_pos_encode_block(C.c_void_p(x.data_ptr()), 3, n, 3, 10, None,
                                                  C.c_void_p(scene.data_ptr() + 4 * cs * (2 * F + 1)), scene.shape[1], sp), "pos_encode_block")
            _lib.check(l.objnerf_pos_encode_block(C.c_void_p(raw.data_ptr() + 4 * cs), C_, n, co, F, None, C.c_void_p(obj.data_ptr()),
                                                  obj.shape[1], sp), "pos_encode_block")
            return scene, obj
        scene = torch.empty(n, 271, dtype=torch.float32, device=x.device)
        obj = torch.empty(n, 104, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().objnerf_voxel_embed(C.byref(g), _lib.ptr(x), n, _lib.ptr(scene), _lib.ptr(obj),
                                                  _lib.stream_ptr()), "voxel_embed")
        return scene, obj
