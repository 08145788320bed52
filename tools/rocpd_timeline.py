#!/usr/bin/env python
"""One steady-state training step of a rocprofv3 --kernel-trace rocpd database in LAUNCH ORDER: start offset, duration and the idle gap
in front of every kernel, then the sums -- where a step's wall time goes that the per-kernel totals do not show (gaps between tiny
kernels).  A step = the kernels between two consecutive launches of the marker kernel (default: the first mlp_kernel of a step, i.e.
every second one with two passes per step).  tools/rocpd_timeline.py DB [marker-substring] [launches of the marker per step] [step index]"""
import sqlite3
import sys


def main(path, marker="mlp_kernel", per_step=2, which=-2):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    starts = marks[::per_step]
    if len(starts) < 3:
        sys.exit("not enough steps in the trace")
    a, b = starts[which], starts[which + 1] if which + 1 != 0 else len(rows)
    # a step begins with the first kernel after the previous step's optimizer: walk back from the marker over the kernels that
    # precede it inside the same step (sampling, packing, fills) -- up to the last multi_tensor_apply (Adam) launch
    has_opt = any("multi_tensor_apply" in r[0] for r in rows)

    def step_begin(i):
        j = i
        if not has_opt:                 # an inference trace: the marker itself opens the step
            return j
        while j > 0 and "multi_tensor_apply" not in rows[j - 1][0]:
            j -= 1
        return j
    a, b = step_begin(a), step_begin(b)
    seg = rows[a:b]
    t0 = seg[0][1]
    print("# one step: %d kernels, %.3f ms wall (first start to last end)" % (len(seg), (seg[-1][2] - t0) / 1e6))
    print("| # | start ms | dur us | gap us | kernel |")
    print("|---|---|---|---|---|")
    prev_end = t0
    busy = gaps = 0.0
    small_n = 0
    small_busy = small_gap = 0.0
    for i, (name, s, e) in enumerate(seg):
        gap = max(0.0, (s - prev_end) / 1e3)
        dur = (e - s) / 1e3
        busy += dur
        gaps += gap
        if dur < 30.0:
            small_n += 1
            small_busy += dur
            small_gap += gap
        print("| %d | %.3f | %.1f | %.1f | %s |" % (i, (s - t0) / 1e6, dur, gap, name[:100]))
        prev_end = max(prev_end, e)
    print()
    print("kernel time %.3f ms, idle gaps %.3f ms; kernels under 30 us: %d, %.3f ms of kernel time + %.3f ms of gaps in front of them"
          % (busy / 1e3, gaps / 1e3, small_n, small_busy / 1e3, small_gap / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *[int(x) for x in sys.argv[3:5]])
