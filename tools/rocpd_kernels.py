#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 `--kernel-trace` result database (rocpd SQLite format, the default output of
ROCm 7.2): calls, total / average / min duration, optionally restricted to a grid size.
usage: rocpd_kernels.py <results.db> [csv_out]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
    agg = defaultdict(list)
    for name, s, e, gx, gy, gz in rows:
        short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
        agg[(short, gx, gy, gz)].append((e - s) / 1e3)
    out = []
    for (name, gx, gy, gz), d in agg.items():
        out.append((sum(d), name, gx, gy, gz, len(d), sum(d) / len(d), min(d)))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    lines = ["total_us,pct,calls,avg_us,min_us,grid,kernel"]
    for t, name, gx, gy, gz, n, avg, mn in out:
        lines.append(f"{t:.1f},{100 * t / tot:.1f},{n},{avg:.2f},{mn:.2f},{gx}x{gy}x{gz},\"{name}\"")
    txt = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
