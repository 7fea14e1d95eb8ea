#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE) of one
bench.py command into a small JSON + CSV summary that is committed under profiles/.

usage: summarize_profile.py <trace_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix> [n] [timed_launches]

timed_launches: the last that many launches of the dominant kernel are bench.py's timed region (2 per step); their
average is reported next to the all-launch average of --stats, which also contains the warm-up launches.

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so the read side is doubled.  The
correction is calibrated in the same run on grid_to_cvec_kernel, which reads a known 8*n^3 bytes once."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load_counter(d, counter):
    agg = defaultdict(list)
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    trace, fetch, write, out = sys.argv[1:5]
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    timed = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    durations = defaultdict(list)
    if timed:
        for f in glob.glob(f"{trace}/**/*_kernel_trace.csv", recursive=True):
            for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
                durations[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    stats = []
    for f in glob.glob(f"{trace}/**/*_kernel_stats.csv", recursive=True):
        stats += list(csv.DictReader(open(f)))
    fe, wr = load_counter(fetch, "FETCH_SIZE"), load_counter(write, "WRITE_SIZE")
    rows = []
    for r in stats:
        name = r["Name"]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short[: short.index("(")] if "(" in short else short
        f_kib = sum(fe[name]) / len(fe[name]) if fe.get(name) else None
        w_kib = sum(wr[name]) / len(wr[name]) if wr.get(name) else None
        rows.append({"kernel": short, "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"]), "FETCH_SIZE_KiB_raw": f_kib, "WRITE_SIZE_KiB": w_kib})
        if timed and len(durations.get(name, [])) >= timed and "grid_color_sweep" in name:
            last = durations[name][-timed:]
            rows[-1]["timed_region_launches"] = timed
            rows[-1]["timed_region_avg_us"] = sum(last) / len(last)
    # calibration of the gfx950 FETCH_SIZE correction on a kernel with a known read volume
    cal = next((x for x in rows if "grid_to_cvec" in x["kernel"] and x["FETCH_SIZE_KiB_raw"]), None)
    corr = 2.0
    calib = None
    if cal:
        known = 8.0 * n ** 3
        calib = {"kernel": cal["kernel"], "known_read_bytes": known, "FETCH_SIZE_bytes_raw": cal["FETCH_SIZE_KiB_raw"] * 1024, "ratio_known_over_raw": known / (cal["FETCH_SIZE_KiB_raw"] * 1024)}
    for x in rows:
        if x["FETCH_SIZE_KiB_raw"] is not None and x["WRITE_SIZE_KiB"] is not None:
            x["hbm_bytes_per_launch"] = corr * x["FETCH_SIZE_KiB_raw"] * 1024 + x["WRITE_SIZE_KiB"] * 1024
    summary = {"n": n, "fetch_correction": corr, "fetch_calibration": calib, "kernels": rows}
    json.dump(summary, open(out + ".json", "w"), indent=1)
    with open(out + ".csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "pct", "FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB", "hbm_bytes_per_launch(2*fetch+write)"])
        for x in rows:
            w.writerow([x["kernel"], x["calls"], f"{x['avg_us']:.1f}", f"{x['min_us']:.1f}", f"{x['max_us']:.1f}", x["pct"], x["FETCH_SIZE_KiB_raw"], x["WRITE_SIZE_KiB"], x.get("hbm_bytes_per_launch")])
    print(json.dumps(summary["kernels"][:3], indent=1))


if __name__ == "__main__":
    main()
