#!/usr/bin/env python
"""Achieved HBM rate of the sampling / compositing / PDF kernels of one bench.py trace (these stages are
HBM-bound, the MLP kernel is not).  Algorithmic bytes per launch for the bench workload (640x480 rays, 64 coarse + 64 fine):
  sample_coarse      read 32 B/ray (the ray row), write 4 S B/ray (z)
  composite          read z, sigma, rgb of both branches = 36 B/eval, write weights 4 B/eval, 40 B/ray of maps
  sample_pdf_merge   read 2 x 4 S B/ray (z, weights), write 4 (S + I) B/ray
  composite_finish   (round 3: the eval-mode compositing lives in the MLP kernel's epilogue; this is its per-ray second half)
                     read + write the weights 8 B/eval, read 64 B of segment records per 32 evals, write 40 B/ray of maps
  ray_bias           (round 3) read the ray's direction and code (12 + 256 B/ray), write 1792 B/ray
Usage: python tools/hbm_rates.py <rocpd results.db> [n_rays=307200] [S=64] [I=64]"""
import sqlite3
import sys

PEAK, ACHIEVABLE = 8.0e12, 6.3e12      # MI355X_MICROARCH.md: HBM3E spec / measured achievable


def main(db, n=307200, S=64, I=64):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(end - start), min(end - start), max(end - start) from kernels "
                       "where name like '%sample_coarse%' or name like '%composite_kernel%' or name like '%sample_pdf_merge%' "
                       "or name like '%composite_finish%' or name like '%ray_bias%' "
                       "group by name").fetchall()
    print("| kernel | calls | avg us | algorithmic MB / launch | GB/s | of 8 TB/s | note |")
    print("|---|---|---|---|---|---|---|")
    for name, calls, avg, mn, mx in rows:
        if "sample_coarse" in name:
            b, note = n * (32 + 4 * S), ""
        elif "composite_finish" in name:
            b, note = n * (10 * (S + S + I) / 2.0 + 40), "mean of the coarse (S) and fine (S + I) launch; half of each wave idles on the weights"
        elif "ray_bias_weights" in name:
            continue
        elif "ray_bias" in name:
            b, note = n * (268 + 1792), "lane = ray, weights as wave-uniform operands, 64-byte stores"
        elif "sample_pdf_merge" in name:
            b, note = n * (8 * S + 4 * (S + I)), "not bandwidth-bound: one wave per ray (float64 prefix scan of the cdf, per-lane binary searches, merge)"
        else:
            # two launches per frame: S and S + I samples; report on the mean
            b, note = n * (40 * (S + S + I) / 2.0 + 40), "mean of the coarse (S) and fine (S + I) launch"
        rate = b / (avg * 1e-9)
        print("| %s | %d | %.1f | %.1f | %.0f | %.2f | %s |" % (name.split("(")[0].replace("objnerf::", ""), calls, avg / 1e3, b / 1e6,
                                                               rate / 1e9, rate / PEAK, note))


if __name__ == "__main__":
    main(sys.argv[1], *[int(x) for x in sys.argv[2:5]])
