#!/usr/bin/env python
"""A/B timing of MLP-kernel build variants (object_nerf_amd/tune/libobjnerf_<tag>.so, built by
`make -C object_nerf_amd/csrc variant TAG=.. DEFS=..`): interleaved rounds in ONE process, median ms
of the fused scene+object voxel kernel on a quarter frame, plus a checksum to catch broken variants.
Usage: python tools/tune_mlp.py tagA tagB ...   (tag 'ship' = the shipped libobjnerf_hip.so)"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import object_nerf_amd as A  # noqa: E402
from object_nerf_amd import _lib, synth  # noqa: E402


def load(tag):
    path = _lib.LIB_PATH if tag == "ship" else os.path.join(ROOT, "object_nerf_amd", "tune", "libobjnerf_%s.so" % tag)
    l = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, args
    return l


def main(tags, n_rays=76800, S=128, rounds=5):
    dev = "cuda"
    sc = synth.build_scene(A, True, preset=synth.TOYDESK_LIKE, device=dev)
    rays = synth.camera_rays(640, 480, near=0.05, far=1.5).to(dev)[:: 307200 // n_rays][:n_rays].contiguous()
    z = (rays[:, 6:7] + (rays[:, 7:8] - rays[:, 6:7]) * torch.linspace(0, 1, S, device=dev)).contiguous()
    codes = sc.code_library.embedding_instance.weight.detach()[1].contiguous()
    m = sc.models["fine"]
    params = m._param_list()
    out = [torch.empty(n_rays, S, device=dev), torch.empty(n_rays, S, 3, device=dev),
           torch.empty(n_rays, S, device=dev), torch.empty(n_rays, S, 3, device=dev)]
    grid = sc.embeddings["xyz"].grid_struct()
    libs, packs = {}, {}
    for t in tags:
        l = load(t)
        nb, na = l.objnerf_blob_floats(1), l.objnerf_aux_floats()
        bi, ai = torch.empty(nb, dtype=torch.int32), torch.empty(na, dtype=torch.int32)
        assert l.objnerf_pack_index(1, C.c_void_p(bi.data_ptr()), C.c_void_p(ai.data_ptr())) == 0
        bi, ai = bi.to(dev), ai.to(dev)
        blob, aux = torch.empty(nb, device=dev), torch.empty(na, device=dev)
        table = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        assert l.objnerf_pack_weights(1, _lib.ptr(bi), _lib.ptr(ai), table, _lib.ptr(blob), _lib.ptr(aux), _lib.stream_ptr()) == 0
        libs[t], packs[t] = l, (blob, aux)

    def args_for(t):
        a = _lib.MlpArgs()
        a.use_voxel, a.do_scene, a.do_object = 1, 1, 1
        a.blob, a.aux = packs[t][0].data_ptr(), packs[t][1].data_ptr()
        a.rays, a.z_vals, a.n_rays, a.S = rays.data_ptr(), z.data_ptr(), n_rays, S
        a.codes, a.code_stride, a.grid = codes.data_ptr(), 0, grid
        a.sigma, a.rgb, a.inst_sigma, a.inst_rgb = (o.data_ptr() for o in out)
        return a

    times = {t: [] for t in tags}
    errs, ref_out = {}, None
    sums = {}
    for r in range(rounds + 1):
        for t in tags:
            a = args_for(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = libs[t].objnerf_mlp_eval(C.byref(a), _lib.stream_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert rc == 0, libs[t].objnerf_last_error()
            if r > 0:
                times[t].append(dt * 1e3)
            sums[t] = (out[0].double().sum().item(), out[1].double().sum().item(), out[2].double().sum().item(), out[3].double().sum().item())
            if r == rounds:          # element-wise distance of this variant's outputs from the first tag's (normwise, per output)
                if t == tags[0]:
                    ref_out = [o.clone() for o in out]
                errs[t] = tuple(((o - q).abs().max() / q.abs().max()).item() for o, q in zip(out, ref_out))
    evals = n_rays * S
    base = None
    for t in tags:
        med = sorted(times[t])[len(times[t]) // 2]
        tf = evals * 1776128 / (med * 1e-3) / 1e12
        base = base or med
        print("%-14s median %8.3f ms  min %8.3f  %6.1f TFLOP/s  %.3f of peak  x%.3f vs first  checksum %.6e %.6e %.6e %.6e"
              "  normwise err vs first (sigma, rgb, isigma, irgb) %.1e %.1e %.1e %.1e"
              % (t, med, min(times[t]), tf, tf / 157.3, base / med, *sums[t], *errs[t]), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["ship"])
