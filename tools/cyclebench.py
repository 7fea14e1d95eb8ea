#!/usr/bin/env python3
"""ONE secondary workload of bench.py -- set-up, warm-up, then S samples between two marker kernels (torch.cumsum: a
scan kernel nothing else here launches) -- for tools/profile_cycles.sh: rocprofv3 --kernel-trace / --pmc FETCH_SIZE /
--pmc WRITE_SIZE passes, summed between the markers by tools/summarize_cycles.py into HBM bytes and kernel time per sample.

usage: cyclebench.py <key> [samples]     key as bench.py's cycle_roofline: mgmc_257_5 | mgmc_513_6 |
       mgmc_lowrank_257_5_k3 | sell_sweep_377089 | sell_sweep_1505793 | mgmc_aij_377089"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parmgmc_amd import COLORING_ITERATED, MCSOR, MGMC, make_observation_mats  # noqa: E402

key = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mark = torch.arange(7, dtype=torch.float64, device="cuda")


def marker():
    torch.cuda.synchronize()
    torch.cumsum(mark, 0)
    torch.cuda.synchronize()


def lshape(refine):
    from parmgmc_amd.unstructured import assemble_p1, read_gmsh41_triangles, refine_uniform

    xy, tris = read_gmsh41_triangles(os.path.join(ROOT, "tests", "golden", "lshape.msh"))
    for _ in range(refine):
        xy, tris = refine_uniform(xy, tris)
    return assemble_p1(xy, tris, 1.0)


alg = None
if key.startswith("mgmc_lowrank_") or (key.startswith("mgmc_") and not key.startswith("mgmc_aij_")):
    parts = key.split("_")
    n, levels = (int(parts[2]), int(parts[3])) if parts[1] == "lowrank" else (int(parts[1]), int(parts[2]))
    mg = MGMC(n, n, n, 10.0, levels)
    b = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
    if parts[1] == "lowrank":
        k = int(parts[4][1:])
        centres = [(0.25, 0.25, 0.25), (0.75, 0.75, 0.75), (0.25, 0.75, 0.5)] + [(0.5, 0.5, 0.1 + 0.8 * q / max(1, k - 4)) for q in range(max(0, k - 3))]
        radii = ([0.1, 0.15, 0.1] + [0.08] * max(0, k - 3))[:k]
        B, Sd, f = make_observation_mats(n, n, n, np.asarray(centres[:k]).ravel(), radii, np.resize([1.0, -1.0], k), 1e-4)
        mg.set_lowrank(B, Sd)
        b = torch.as_tensor(f, device="cuda")
        del B
    mg.setup()
    y = torch.zeros(n ** 3, dtype=torch.float64, device="cuda")
    alg = mg.algorithmic_bytes()[0]
    run = lambda its, c0: mg.sample(b, y, its, seed=0xCAFE, counter0=c0)
elif key.startswith("sell_sweep_"):
    rows = int(key.split("_")[2])
    A = lshape({377089: 5, 1505793: 6}[rows])
    mc = MCSOR(A.indptr, A.indices, A.data, int(os.environ.get('PMG_BENCH_COLORING', COLORING_ITERATED))).setup()  # as bench.py: first-fit + one round of iterated greedy
    b = torch.ones(rows, dtype=torch.float64, device="cuda")
    y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    alg = 12 * A.nnz + 40 * rows
    run = lambda its, c0: mc.sample(b, y, its, seed=0xCAFE, counter0=c0, scaled=True)
elif key.startswith("mgmc_aij_"):
    from parmgmc_amd.unstructured import build_hierarchy

    A = lshape(5)
    ops, ps = build_hierarchy(A, coarse_max=2000)
    mg = MGMC.from_hierarchy(ops, ps)
    mg.set_coloring(int(os.environ.get('PMG_BENCH_COLORING', COLORING_ITERATED)))
    mg.set_smoother(True, 1.0, 1, 1)
    mg.setup()
    b = torch.ones(A.shape[0], dtype=torch.float64, device="cuda")
    y = torch.zeros(A.shape[0], dtype=torch.float64, device="cuda")
    alg = mg.algorithmic_bytes()[0]
    run = lambda its, c0: mg.sample(b, y, its, seed=0xCAFE, counter0=c0)
else:
    raise SystemExit(f"unknown workload {key}")

c = run(3, 0)
marker()
t = time.perf_counter()
run(S, c)
torch.cuda.synchronize()
dt = time.perf_counter() - t
marker()
print(json.dumps({"key": key, "samples": S, "ms_per_sample_wall": dt / S * 1e3, "algorithmic_bytes_per_sample": alg}), flush=True)
