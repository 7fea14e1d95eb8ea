#!/usr/bin/env python3
"""Does the numbering of the mesh nodes matter?  The P1 matrix of lshape.msh refined r times in the order refine_uniform
leaves (old nodes first, the new edge midpoints appended level after level) against the same matrix renumbered by reverse
Cuthill-McKee: sliced-ELL sweep and MGMC sample (development tool)."""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import COLORING_ITERATED, MCSOR, MGMC
from parmgmc_amd.unstructured import assemble_p1, build_hierarchy, read_gmsh41_triangles, refine_uniform

xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent.parent / "tests" / "golden" / "lshape.msh")
for r in range(1, 7):
    xy, tris = refine_uniform(xy, tris)
    if r < 5:
        continue
    A0 = assemble_p1(xy, tris, 1.0)
    t = time.perf_counter()
    perm = reverse_cuthill_mckee(A0.tocsr(), symmetric_mode=True)
    A1 = A0.tocsr()[perm][:, perm].tocsr()
    A1.sort_indices()
    trcm = time.perf_counter() - t
    for name, A in (("append order", A0), ("rcm", A1)):
        n = A.shape[0]
        mc = MCSOR(A.indptr, A.indices, A.data, COLORING_ITERATED).setup()
        b = torch.ones(n, dtype=torch.float64, device="cuda")
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        c = mc.sample(b, y, 20, seed=1, counter0=0, scaled=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        mc.sample(b, y, 200, seed=1, counter0=c, scaled=True)
        e1.record()
        torch.cuda.synchronize()
        line = f"refine {r}: {n} rows, {name:12s}: {mc.get_num_colors()} colours, sweep {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us"
        if r == 5:
            ops, ps = build_hierarchy(A, coarse_max=2000)
            mg = MGMC.from_hierarchy(ops, ps)
            mg.set_coloring(COLORING_ITERATED)
            mg.set_smoother(True, 1.0, 1, 1)
            mg.setup()
            y.zero_()
            c = mg.sample(b, y, 10, seed=1, counter0=0)
            torch.cuda.synchronize()
            e0.record()
            mg.sample(b, y, 60, seed=1, counter0=c)
            e1.record()
            torch.cuda.synchronize()
            line += f", MGMC sample {e0.elapsed_time(e1) / 60:6.4f} ms on {[len(o[0]) - 1 for o in ops]}"
        print(line + (f"   (rcm {trcm:.2f} s)" if name == "rcm" else ""), flush=True)
