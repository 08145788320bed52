#!/usr/bin/env python
"""What will 8 ranks do?  Replays, on ONE GPU and one after the other, the share of a strong-scaling frame that each of the
WORLD ranks of `bench.py --gpus WORLD` would render (BASELINE configs[3]: the sharded ScanNet-multi frame; configs[4]: the
editing demo), for both ways of dealing the image rows -- contiguous bands and row blocks round-robin -- and reports

    per-rank render ms (whole share, hipEvents) and MLP-kernel ms (the library's own launch timing),
    spread          max / mean - 1 of the per-rank render time: what the single all-gather waits for,
    predicted strong-scaling efficiency = mean / (max + gather_ms): N ranks finish when the slowest has rendered and the one
                    pixel all-gather has run; `mean` is the work perfectly divided.

gather_ms is measured here as far as one GPU allows -- the packed message assembly, a world-1 RCCL all_gather_into_tensor of
one rank's message and the inverse permutation of the FULL-frame buffer -- plus a modelled wire time for the 8-rank ring over
xGMI ((WORLD-1) x message bytes at 50 GB/s per direction and 10 us per hop; MI355X_MICROARCH.md: 153 GB/s per link peak).

Run on the GPU box:  python tools/band_replay.py gpurun_out/r03_band_replay.md [--world 8] [--steps 3]
Every rank's pixels are also checked to be bit-equal to the same pixels of the unsharded frame."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out", nargs="?", default="gpurun_out/r03_band_replay.md")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--row-block", type=int, default=4)
    a = ap.parse_args()
    import ctypes as C
    import torch.distributed as dist
    from object_nerf_amd import _lib
    from object_nerf_amd.distributed import RayShards, gather_pixel_maps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    R = bench.HipRenderer(dev)
    lib = _lib.lib()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    lines = ["# Band replay: the %d ranks' shares of one strong-scaling frame, rendered one after the other on ONE MI355X (round 3)" % a.world, "",
             "`tools/band_replay.py`. render ms = hipEvents around the rank's whole share (ray generation for configs[4], "
             "both passes, compositing), MLP ms = the fused kernel alone; spread = max / mean - 1; predicted efficiency = "
             "mean / (max + gather). Median of %d timed steps per rank after one warm-up." % a.steps, ""]
    summ = ["| config | split | mean render ms | max render ms | spread | gather ms (measured part + modelled wire) | predicted %d-rank efficiency | pixels bit-equal to the unsharded frame |" % a.world,
            "|---|---|---|---|---|---|---|---|"]
    for cfg in (3, 4):
        scene = None
        whole = None
        for mode in ("contiguous", "cyclic"):
            args = bench.parse(["--config", str(cfg), "--shard", mode, "--row-block", str(a.row_block), "--cpu-rays", "0", "--pmc", "off"])
            if whole is None:
                w1 = bench.Workload(cfg, args, R, 0, 1, "strong", None, scene=scene)
                scene = w1.sc
                w1.step()
                whole = {k: w1.last[k].clone() for k in w1.gather_keys}
                R.sync()
            shards = RayShards.rows(args.height, args.width, a.world, a.row_block if mode == "cyclic" else None)
            rows, equal = [], True
            for r in range(a.world):
                wl = bench.Workload(cfg, args, R, r, a.world, "strong", None, as_rank=True, scene=scene)
                wl.step()
                R.sync()
                wl.marks.clear()
                lib.objnerf_timing_enable(1)
                for _ in range(a.steps):
                    wl.step()
                R.sync()
                launches, kms = C.c_int64(0), C.c_double(0.0)
                lib.objnerf_timing_read(C.byref(launches), C.byref(kms))
                lib.objnerf_timing_enable(0)
                # median over the timed steps: one step that catches a clock dip must not pass for imbalance
                per_step = sorted(x.elapsed_time(y) for x, y, _ in wl.marks)
                render_ms = per_step[len(per_step) // 2]
                idx = shards.local_index(r, dev)
                for k in wl.gather_keys:
                    equal = equal and torch.equal(wl.last[k], whole[k][idx])
                rows.append((r, wl.n_local, wl.evals_rank, render_ms, kms.value / a.steps))
                last_local = {k: wl.last[k] for k in wl.gather_keys}
            # the collective's cost as far as one GPU shows it: message assembly + world-1 all-gather + full-size restore
            per = shards.per
            cols = sum(int(torch.Size(v.shape[1:]).numel()) for v in last_local.values())
            full = torch.empty(a.world * per, cols, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gather_pixel_maps(last_local, None)
                shards.restore(full)
            torch.cuda.synchronize()
            measured = (time.perf_counter() - t0) / 20 * 1e3
            msg = per * cols * 4
            wire = (a.world - 1) * (msg / 50e9 + 10e-6) * 1e3
            gather = measured + wire
            ms = [x[3] for x in rows]
            mean, mx = sum(ms) / len(ms), max(ms)
            eff = mean / (mx + gather)
            lines += ["## configs[%d], %s%s" % (cfg, mode, " (%d-row blocks round-robin)" % a.row_block if mode == "cyclic" else " bands"), "",
                      "| rank | rays | evaluated sample points | render ms | MLP kernel ms |", "|---|---|---|---|---|"]
            for r, n, ev, rm, km in rows:
                lines.append("| %d | %d | %.3e | %.2f | %.2f |" % (r, n, ev, rm, km))
            lines += ["", "mean %.2f ms, max %.2f ms, spread %.1f %%; message %d B per rank; gather %.3f ms measured (assembly + world-1 "
                      "all-gather + restore of the %d-row frame buffer) + %.3f ms modelled wire = %.3f ms; predicted efficiency %.3f"
                      % (mean, mx, 100 * (mx / mean - 1), msg, measured, a.world * per, wire, gather, eff), ""]
            summ.append("| %d | %s | %.2f | %.2f | %.1f %% | %.3f | %.3f | %s |" % (cfg, mode, mean, mx, 100 * (mx / mean - 1), gather, eff, equal))
    lines += ["## Summary", ""] + summ
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(summ))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
