#!/usr/bin/env python3
"""Launch-by-launch timeline of the LAST of `nsamples` identical samples in a rocprofv3 kernel trace:
ktimeline.py <dir> <nsamples> -> start offset, duration, gap to the previous kernel's end, name, grid (all us)"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n[: n.index("(")] if "(" in n else n
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
rows.sort()
ns = int(sys.argv[2])
# the timed samples lie between the last two marker kernels of tools/cyclebench.py (torch.cumsum: a scan kernel)
marks = [i for i, r in enumerate(rows) if "scan" in r[2].lower() or "cumsum" in r[2].lower()]
if len(marks) >= 2:
    rows = rows[marks[-2] + 1 : marks[-1]]
per, extra = divmod(len(rows), ns)  # a call's head and tail (layout changes of b and y) are not part of a sample
names = [r[2] + r[3] for r in rows]
o = next(o for o in range(extra + 1) if names[o : o + per] == names[o + per : o + 2 * per])
rows = rows[: o + ns * per]
last = rows[-per:]
t0, prev_end = last[0][0], rows[-per - 1][1]
tot_k = tot_gap = 0.0
for s, e, n, gx, gy, gz in last:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {n[:60]:60s} {gx:>8s} {gy:>5s} {gz:>5s}")
    tot_k += (e - s) / 1e3
    tot_gap += max(0, s - prev_end) / 1e3
    prev_end = e
print(f"launches {per}  kernel time {tot_k:.1f} us  gaps {tot_gap:.1f} us  span {(last[-1][1] - t0) / 1e3:.1f} us")
