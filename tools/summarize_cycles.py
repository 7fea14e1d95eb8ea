#!/usr/bin/env python3
"""Sums the rocprofv3 passes of tools/cyclebench.py between its two marker kernels into per-sample figures.

usage: summarize_cycles.py <dir_with_<key>_{trace,fetch,write}> <out.json> key[:samples] ...
Per workload: kernel time per sample and per kernel name (--kernel-trace), HBM bytes per sample = (2 x FETCH_SIZE +
WRITE_SIZE) x 1024 summed over every dispatch between the markers (MI355X_MICROARCH.md: KiB units, gfx950 FETCH_SIZE
reports half of a streaming read; the counter includes Infinity-Cache hits).  Includes whatever the sample launches that
is not part of the cycle proper EXCEPT the natural <-> layout conversions at both ends of a sample CALL (grid_to_cvec /
grid_from_cvec / permute_in / permute_out: once per call of `its` samples -- bench.py's lines run 20-50 samples per call --
so they are left out here, listed under "excluded")."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))


EXCLUDE = re.compile(r"grid_to_cvec|grid_from_cvec|permute_in_kernel|permute_out_kernel")


def between_markers(rows, key_id):
    rows = sorted(rows, key=key_id)
    idx = [i for i, r in enumerate(rows) if "scan" in r["Kernel_Name"].lower() or "cumsum" in r["Kernel_Name"].lower()]
    if len(idx) < 2:
        raise SystemExit(f"markers not found ({len(idx)})")
    return [r for r in rows[idx[-2] + 1: idx[-1]] if not EXCLUDE.search(r["Kernel_Name"])]


def load(d, pat):
    out = []
    for f in glob.glob(f"{d}/**/*{pat}", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    base, out = sys.argv[1], sys.argv[2]
    res = {"fetch_correction": 2.0, "excluded": EXCLUDE.pattern, "workloads": {}}
    for spec in sys.argv[3:]:
        key, _, s = spec.partition(":")
        S = int(s) if s else 5
        w = {"samples": S}
        tr = load(f"{base}/{key}_trace", "_kernel_trace.csv")
        if tr:
            seg = between_markers(tr, lambda r: int(r["Start_Timestamp"]))
            per = defaultdict(lambda: [0, 0.0])
            for r in seg:
                k = short(r["Kernel_Name"])
                per[k][0] += 1
                per[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            w["kernel_us_per_sample"] = sum(v[1] for v in per.values()) / S
            w["span_us_per_sample"] = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / S
            w["launches_per_sample"] = sum(v[0] for v in per.values()) / S
            w["kernels"] = sorted(({"kernel": k, "launches_per_sample": v[0] / S, "us_per_sample": v[1] / S} for k, v in per.items()), key=lambda x: -x["us_per_sample"])
        tot = 0.0
        bykernel = defaultdict(lambda: [0.0, 0.0])
        for which, ctr, mul in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
            rows = [r for r in load(f"{base}/{key}_{which}", "_counter_collection.csv") if r["Counter_Name"] == ctr]
            if not rows:
                tot = None
                break
            seg = between_markers(rows, lambda r: int(r["Dispatch_Id"]))
            v = sum(float(r["Counter_Value"]) for r in seg) * 1024.0 * mul
            w[f"{which}_bytes_per_sample"] = v / S
            for r in seg:
                bykernel[short(r["Kernel_Name"])][0 if which == "fetch" else 1] += float(r["Counter_Value"]) * 1024.0 * mul / S
            tot += v
        if tot is not None:
            w["hbm_bytes_per_sample"] = tot / S
            for k in w.get("kernels", []):
                if k["kernel"] in bykernel:
                    k["read_bytes_per_sample"], k["write_bytes_per_sample"] = bykernel[k["kernel"]]
        log = glob.glob(f"{base}/{key}_trace.log")
        if log:
            for line in open(log[0]):
                if line.startswith("{"):
                    w.update({k: v for k, v in json.loads(line).items() if k in ("algorithmic_bytes_per_sample", "ms_per_sample_wall")})
        if w.get("hbm_bytes_per_sample") and w.get("algorithmic_bytes_per_sample"):
            w["traffic_over_algorithmic"] = w["hbm_bytes_per_sample"] / w["algorithmic_bytes_per_sample"]
        res["workloads"][key] = w
    json.dump(res, open(out, "w"), indent=1)
    for k, w in res["workloads"].items():
        print(k, {a: w.get(a) for a in ("kernel_us_per_sample", "launches_per_sample", "hbm_bytes_per_sample", "algorithmic_bytes_per_sample", "traffic_over_algorithmic")})


if __name__ == "__main__":
    main()
