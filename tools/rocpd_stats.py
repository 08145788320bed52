#!/usr/bin/env python
"""Summarises a rocprofv3 rocpd database (ROCm 7.2 default --kernel-trace output) into the
per-kernel table `rocprofv3 --stats` prints for CSV output: calls, total/avg/min/max ns, %."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % | vgpr | agpr | sgpr | lds | scratch | grid | wg |", "|" + "---|" * 14]
    for r in rows:
        extra = list(r[6:]) + [""] * (7 - len(r[6:]))
        lines.append("| %s | %d | %.3f | %.4f | %.4f | %.4f | %.2f | %s |" % (
            r[0][:90], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / tot, " | ".join(str(e) for e in extra)))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
