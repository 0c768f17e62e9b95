#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats rocpd database (*_results.db) as text:
per-kernel calls / total / average / min / max duration and share, optionally split by grid.
  python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--by-grid] [--frames N]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--by-grid", action="store_true")
    ap.add_argument("--frames", type=int, default=0, help="forwards in the trace: adds per-forward columns")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    grp = "name, grid_x, grid_y" if a.by_grid else "name"
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by sum(end-start) desc" % (grp, grp)).fetchall()
    tot = sum(r[-4] for r in rows)
    print("# rocprofv3 kernel-trace summary of %s; total kernel time %.3f ms" % (a.db.split("/")[-1], tot / 1e6))
    hdr = "%-64s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share")
    if a.frames:
        hdr += " %10s %10s" % ("calls/fwd", "us/fwd")
    print(hdr)
    for r in rows:
        name = r[0].replace("lspf2f::", "").replace("(IgemmParams)", "").replace("void ", "")
        if a.by_grid:
            name = "%s grid(%d,%d)" % (name[:44], r[1], r[2])
            r = (r[0],) + r[3:]
        line = "%-64s %8d %12.1f %10.2f %10.2f %10.2f %6.1f%%" % (name[:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                  r[5] / 1e3, 100.0 * r[2] / tot)
        if a.frames:
            line += " %10.2f %10.1f" % (r[1] / a.frames, r[2] / 1e3 / a.frames)
        print(line)


if __name__ == "__main__":
    main()
