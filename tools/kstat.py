#!/usr/bin/env python3
"""Average duration per (kernel, grid) from a rocprofv3 kernel-trace directory: kstat.py <dir> [substring ...]"""
import collections
import csv
import glob
import sys

d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n[: n.index("(")] if "(" in n else n
        d[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
want = sys.argv[2:]
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if (not want or any(s in k[0] for s in want)) and sum(v) > 500:
        print(f"{k[0][:52]:52s} {k[1]:>7s} {k[2]:>5s} {k[3]:>5s} n={len(v):4d} avg={sum(v) / len(v):8.1f} us  min={min(v):8.1f}")
