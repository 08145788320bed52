#!/usr/bin/env python
"""the density-grid query of the reference's tools/extract_mesh.py:62-113 -- N^3 lattice points
(N_grid = 512 by default there), nerf_fine's sigma for the scene (obj_id 0) or one object (obj_id > 0) -- timed on the device
in the two forms the product offers:

  fused   ONE enqueue: `ObjectNeRF.query_sigma(embedding_xyz, lattice=(x, y, z))` -- the MLP kernel generates the lattice
          points, embeds them in registers and stops after the density head (objnerf_mlp_args.lat_*, sigma_only);
  memory  the script's own loop on the drop-in types: chunks of 32 * 2014 points, per chunk `embedding_xyz(xyz)` ->
          `nerf_fine.forward({"emb_xyz", "obj_voxel"}, sigma_only=True)` / `forward_instance(...)`: embeddings written and
          read back (271 + 104 floats per point), three launches per chunk.

GEMM work per point up to the density head (2 FLOP per MAC; voxel mode): scene 597,760 MAC (8 layers + head), object
439*128 + 128*128 + 567*128 + 128*128 + 128 = 161,664 MAC.
usage: python tools/mesh_query_bench.py [N_grid] [out.md]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import synth  # noqa: E402

DEV = "cuda"
PEAK = 157.3e12
MAC = {0: 597_760, 4: 439 * 128 + 128 * 128 + 567 * 128 + 128 * 128 + 128}


def query_memory(sc, xyz_, chunk, obj_id):
    out_chunks = []
    emb, fine, codes = sc.embeddings["xyz"], sc.models["fine"], sc.code_library
    with torch.no_grad():
        for i in range(0, xyz_.shape[0], chunk):                       # extract_mesh.py:80-111
            xyz_embedded, obj_voxel_embedded = emb(xyz_[i:i + chunk])
            input_dict = {"emb_xyz": xyz_embedded, "obj_voxel": obj_voxel_embedded}
            if obj_id > 0:
                n = xyz_embedded.shape[0]
                input_dict["obj_code"] = codes.embedding_instance(torch.ones(n, device=DEV).long() * obj_id)
                out = fine.forward_instance(input_dict, sigma_only=True)["inst_sigma"]
            else:
                out = fine.forward(input_dict, sigma_only=True)["sigma"]
            out_chunks.append(out)                                     # (the script moves each chunk to the host here)
    return torch.cat(out_chunks, 0)


def query_fused(sc, axes, obj_id):
    fine = sc.models["fine"]
    with torch.no_grad():
        code = sc.code_library.embedding_instance.weight[obj_id] if obj_id > 0 else None
        return fine.query_sigma(sc.embeddings["xyz"], lattice=axes, obj_code=code)


def main(N=512, out=None):
    sc = synth.build_scene(A, use_voxel=True, preset=synth.SCANNET_LIKE, max_voxels=800_000, device=DEV)
    x = np.linspace(-1.5, 1.5, N)
    axes = (x, x, x)
    chunk = 32 * 2014
    lines = ["| query | form | points | launches | s | M points/s | TFLOP/s | of fp32-MFMA peak |", "|---|---|---|---|---|---|---|---|"]
    xyz_ = None
    for name, obj_id in (("scene sigma (obj_id 0)", 0), ("object sigma (obj_id 4)", 4)):
        query_fused(sc, (x[:64], x[:64], x[:64]), obj_id)                # warm-up: packs the weights
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            sf = query_fused(sc, axes, obj_id)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[1]
        assert sf.shape == (N ** 3, 1) and torch.isfinite(sf).all()
        tf = 2.0 * MAC[obj_id] * N ** 3 / dt
        lines.append("| %s | fused, one enqueue | %d | 1 | %.3f | %.1f | %.1f | %.3f |" % (name, N ** 3, dt, N ** 3 / dt / 1e6, tf / 1e12, tf / PEAK))
        if xyz_ is None:
            xyz_ = torch.from_numpy(np.stack(np.meshgrid(x, x, x), -1).reshape(-1, 3)).float().to(DEV)      # extract_mesh.py:62-66
        query_memory(sc, xyz_[: 4 * chunk], chunk, obj_id)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sm = query_memory(sc, xyz_, chunk, obj_id)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nch = (N ** 3 + chunk - 1) // chunk
        tf = 2.0 * MAC[obj_id] * N ** 3 / dt
        lines.append("| %s | memory, the script's chunk loop | %d | %d | %.3f | %.1f | %.1f | %.3f |" % (
            name, N ** 3, nch * (3 if obj_id == 0 else 4), dt, N ** 3 / dt / 1e6, tf / 1e12, tf / PEAK))
        err = ((sf - sm).abs().max() / sm.abs().max()).item()
        lines.append("| | fused vs memory form: max abs difference / max abs sigma = %.2e | | | | | | |" % err)
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write("# extract_mesh.py's density-grid query (tools/mesh_query_bench.py, N_grid = %d; memory form in chunks of 32*2014)\n\n" % N
                             + txt + "\n")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 512, sys.argv[2] if len(sys.argv) > 2 else None)
