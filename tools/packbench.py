#!/usr/bin/env python3
"""Sweep / V-cycle timings at the multigrid sizes 2^k+1 with the packed lane mapping on (default) and off."""
import os
import subprocess
import sys

code = r'''
import sys, time, torch
sys.path.insert(0, "%s")
from parmgmc_amd import GridMCSOR, MGMC
for n in (257, 513):
    g = GridMCSOR(n, n, n, 10.0)
    b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda")); y = g.new_cvec()
    g.sample_cvec(b, y, 5, 1, 0); torch.cuda.synchronize(); t = time.perf_counter()
    g.sample_cvec(b, y, 50, 1, 5); torch.cuda.synchronize()
    print(f"  {n}^3 sweep {(time.perf_counter()-t)/50*1e6:8.1f} us", end="")
    del g, b, y
mg = MGMC(257, 257, 257, 10.0, 5).setup()
b = torch.ones(257**3, dtype=torch.float64, device="cuda"); y = torch.zeros_like(b)
c = mg.sample(b, y, 3, seed=1); torch.cuda.synchronize(); t = time.perf_counter()
mg.sample(b, y, 20, seed=1, counter0=c); torch.cuda.synchronize()
print(f"  257^3 V-cycle {(time.perf_counter()-t)/20*1e3:6.3f} ms")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ("0", "2"):
    print("PMG_GRID_PACKED=" + mode, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PMG_GRID_PACKED=mode), check=True)
