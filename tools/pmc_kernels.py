#!/usr/bin/env python3
"""HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB counters; gfx950 FETCH_SIZE counts half of a streaming
read, see tools/summarize_profile.py) of every kernel in two rocprofv3 --pmc output directories:
pmc_kernels.py <fetch_dir> <write_dir> [substring ...]"""
import collections
import csv
import glob
import sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
want = sys.argv[3:]
for key in sorted(fe, key=lambda k: -sum(fe[k])):
    name = key[0].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name[: name.index("(")] if "(" in name else name
    if want and not any(w in name for w in want):
        continue
    f = sum(fe[key]) / len(fe[key]) * 1024 * 2
    w = sum(wr[key]) / len(wr[key]) * 1024 if key in wr else float("nan")
    if f + w > 5e6:
        print(f"{name[:50]:50s} grid={key[1]:>10s} n={len(fe[key]):4d} read={f / 1e6:9.1f} MB write={w / 1e6:9.1f} MB total={(f + w) / 1e6:9.1f} MB")
