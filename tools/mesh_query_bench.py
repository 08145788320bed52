#!/usr/bin/env python
"""the density-grid query of the reference's tools/extract_mesh.py:63-113, issued the way that script
issues it -- N^3 grid points (N_grid = 512 by default there), chunks of 32 * 2014 points, per chunk
`embedding_xyz(xyz)` -> `nerf_fine.forward({"emb_xyz", "obj_voxel"}, sigma_only=True)["sigma"]` (scene) or
`forward_instance(..., sigma_only=True)["inst_sigma"]` (object id > 0) -- on the drop-in types, timed on the device.
The sigma-only kernel variant stops after the density head: 597,760 MAC per point (scene, voxel mode) instead of 699,904.
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


def query(sc, xyz_, chunk, obj_id):
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


def main(N=512, out=None):
    sc = synth.build_scene(A, use_voxel=True, preset=synth.SCANNET_LIKE, max_voxels=800_000, device=DEV)
    x = np.linspace(-1.5, 1.5, N)
    xyz_ = torch.from_numpy(np.stack(np.meshgrid(x, x, x), -1).reshape(-1, 3)).float().to(DEV)      # extract_mesh.py:62-66
    chunk = 32 * 2014
    lines = ["| query | points | chunks | s | M points/s | TFLOP/s | of fp32-MFMA peak |", "|---|---|---|---|---|---|---|"]
    for name, obj_id, mac in (("scene sigma (obj_id 0)", 0, 597_760), ("object sigma (obj_id 4)", 4, 0)):
        if mac == 0:
            # object branch up to its density head: L_O1..L_O4 + sigma head
            mac = (104 + 64) * 128 + 128 * 128 + (104 + 64 + 128) * 128 + 128 * 128 + 128
        sigma = query(sc, xyz_[: 4 * chunk], chunk, obj_id)              # warm-up: packs the weights
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sigma = query(sc, xyz_, chunk, obj_id)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert sigma.shape == (N ** 3, 1) and torch.isfinite(sigma).all()
        tf = 2.0 * mac * N ** 3 / dt
        lines.append("| %s | %d | %d | %.3f | %.1f | %.1f | %.2f |" % (name, N ** 3, (N ** 3 + chunk - 1) // chunk, dt, N ** 3 / dt / 1e6,
                                                                    tf / 1e12, tf / PEAK))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write("# extract_mesh.py's density-grid query on the drop-in types (tools/mesh_query_bench.py, N_grid = %d, chunk = 32*2014)\n\n" % N
                             + txt + "\n")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 512, sys.argv[2] if len(sys.argv) > 2 else None)
