#!/usr/bin/env python
"""Throughput of the training path's fp32 MFMA GEMM (csrc/gemm.h) on the three products of a 256-wide Linear layer
over one training batch's sample points (2048 rays x 192 = 393,216 points), against the 157.3 TFLOP/s fp32 MFMA peak.
Usage: python tools/gemm_bench.py [lib tag ...]   ('ship' = libobjnerf_hip.so, else object_nerf_amd/tune/libobjnerf_<tag>.so)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_nerf_amd import _lib  # noqa: E402
from tools.tune_mlp import load  # noqa: E402

PEAK = 157.3e12


def main(tags):
    dev = "cuda"
    P = 2048 * 192
    shapes = []
    for w in (256, 128):
        shapes += [("fwd  Y=X W^T  w=%d" % w, 1, 1, P, w, w, 1), ("dgrad dX=dY W w=%d" % w, 1, 0, P, w, w, 1),
                   ("wgrad dW=dY^T X w=%d" % w, 0, 0, w, w, P, max(2, 1024 // ((w // 128) ** 2)))]
    shapes += [("fwd  in=95 -> 256", 1, 1, P, 256, 95, 1), ("wgrad 256 x 95", 0, 0, 256, 95, P, 512),
               # the embedding-gradient product of the first scene layer (dX = dY W, 271 input columns: 3 column tiles
               # that all stream the same P x 256 panel of dY) and its weight gradient, at the fine pass's point count
               ("dgrad dX=dY W 256->271", 1, 0, 2048 * 128, 271, 256, 1), ("wgrad 256 x 271", 0, 0, 256, 271, 2048 * 128, 256),
               # round 5: the embedding-gradient products of the training step (N = 208 / 104 / 64 output columns) and the
               # layer-wise path's first layers (K = 271 / 439 input columns)
               ("dgrad 256 -> 208", 1, 0, 2048 * 128, 208, 256, 1), ("dgrad 128 -> 104", 1, 0, 2048 * 128, 104, 128, 1),
               ("dgrad 128 -> 64", 1, 0, 2048 * 128, 64, 128, 1),
               ("fwd  in=271 -> 256", 1, 1, P, 256, 271, 1), ("fwd  in=439 -> 128", 1, 1, P, 128, 439, 1),
               ("fwd  in=283 -> 128", 1, 1, P, 128, 283, 1)]
    libs = {t: load(t) for t in tags}
    for name, akc, bkc, M, N, K, split in shapes:
        a = torch.randn((M, K) if akc else (K, M), device=dev)
        b = torch.randn((N, K) if bkc else (K, N), device=dev)
        c = torch.zeros(M, N, device=dev)
        ref = (a if akc else a.t()).double()[:4096] @ (b.t() if bkc else b).double() if M > 4096 else None
        line = "%-24s" % name
        for t, l in libs.items():
            def run():
                rc = l.objnerf_gemm(_lib.ptr(a), a.shape[1], akc, _lib.ptr(b), b.shape[1], bkc, _lib.ptr(c), N, M, N, K,
                                    1 if split > 1 else 0, 0, None, split, _lib.stream_ptr())
                assert rc == 0
            c.zero_()
            run()
            err = ""
            if ref is not None:
                err = " err %.1e" % ((c[:4096].double() - ref).abs().max() / ref.abs().max()).item()
            # the error check above leaves the GPU idle for a while (CPU float64 product): give the clocks ~30 ms of this
            # kernel before timing, or whichever library is measured first looks ~10 % slower than the others
            for _ in range(60):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            tf = 2.0 * M * N * K / ms / 1e9
            line += "  | %s %.3f ms %6.1f TF/s (%.2f)%s" % (t, ms, tf, tf * 1e12 / PEAK, err)
        print(line)


if __name__ == "__main__":
    main(sys.argv[1:] or ["ship"])
