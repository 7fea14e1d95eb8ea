#!/usr/bin/env python3
"""The unstructured line of bench.py at larger refinements of lshape.msh, where the sliced-ELL sweep is no longer
launch-bound: python tools/unstructbench.py [refine ...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

for r in [int(a) for a in sys.argv[1:]] or [5, 6, 7]:
    d = bench.unstructured_secondary(refine=r, its=30, larger=False)
    print(json.dumps({"refine": r, "workload": d["workload"][:90], "sweep_ms": d["gibbs_sweep"]["ms_per_sample"], "colors": d["gibbs_sweep"]["colors"], "roofline_frac": d["gibbs_sweep"]["roofline"]["frac"], "mgmc_ms": d["mgmc"]["ms_per_sample"], "host_setup_s": d["host_setup_s"]}), flush=True)
