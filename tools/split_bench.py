#!/usr/bin/env python
"""Split-K sweep of the wgrad GEMM (dW = dY^T X, K = sample points) on the shapes of one training step; the
choice in csrc/train.hip::lin_wgrad comes from this table.  Usage (GPU box): python tools/split_bench.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_nerf_amd import _lib
l = _lib.lib()
dev = "cuda"
for P in (131072, 262144):
    for (M, N) in ((256, 256), (128, 128), (256, 271), (128, 64), (64, 128), (3, 128), (128, 27)):
        a = torch.randn(P, M, device=dev); b = torch.randn(P, N, device=dev); c = torch.zeros(M, N, device=dev)
        line = "P=%d %dx%d:" % (P, M, N)
        for split in (64, 128, 256, 512, 1024, 2048):
            def run():
                assert l.objnerf_gemm(_lib.ptr(a), M, 0, _lib.ptr(b), N, 0, _lib.ptr(c), N, M, N, P, 1, 0, None, split, _lib.stream_ptr()) == 0
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            line += "  s%d %.0fus" % (split, e0.elapsed_time(e1) * 100)
        print(line)
