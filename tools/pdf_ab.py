#!/usr/bin/env python
"""Bitwise A/B of the fine-depth sampler (objnerf_sample_pdf_merge / objnerf_sample_pdf) between two library builds, plus
timing.  Usage: python tools/pdf_ab.py <libA.so> <libB.so>"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_nerf_amd import _lib  # noqa: E402


def load(path):
    l = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, args
    return l


def main(pa, pb):
    dev = "cuda"
    la, lb = load(pa), load(pb)
    g = torch.Generator(device=dev).manual_seed(0)
    for (n, S, I, mode) in [(307200, 64, 64, "det"), (307200, 64, 128, "det"), (20000, 64, 64, "rand"), (5000, 200, 300, "rand"),
                            (3000, 1025, 1023, "det"), (4000, 33, 31, "rand"), (4000, 64, 64, "unsorted-coarse"), (4000, 5, 1, "det")]:
        z = torch.sort(torch.rand(n, S, device=dev, generator=g), -1)[0] * 3 + 0.1
        if mode == "unsorted-coarse":
            z = z.flip(-1).contiguous()
        w = torch.rand(n, S, device=dev, generator=g) ** 4
        w[::7] = 0                      # rays with zero weights everywhere (eps path)
        z[1::11, :] = 0.0               # near = far = 0 rays
        if mode == "det":
            u, us = torch.linspace(0, 1, I, device=dev), 0
        else:
            u, us = torch.rand(n, I, device=dev, generator=g), I
        outs = []
        for l in (la, lb):
            zf = torch.full((n, S + I), -1.0, device=dev)
            zs = torch.full((n, I), -1.0, device=dev)
            args = (_lib.ptr(z), _lib.ptr(w), _lib.ptr(u), us, n, S, I, 1e-5, _lib.ptr(zs), _lib.ptr(zf), _lib.stream_ptr())
            assert l.objnerf_sample_pdf_merge(*args) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                l.objnerf_sample_pdf_merge(*args)
            e1.record()
            torch.cuda.synchronize()
            outs.append((zf, zs, e0.elapsed_time(e1) / 5))
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        print("n=%6d S=%4d I=%4d %-15s bitwise-equal=%s  A %.3f ms  B %.3f ms" % (n, S, I, mode, same, outs[0][2], outs[1][2]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
