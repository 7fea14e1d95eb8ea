#!/usr/bin/env python3
"""What costs the 2^k+1 grids their 15-17 % per point against 2^k?  The noisy sweep (ns per 1000 point updates) on boxes that
are odd in one direction at a time; PMG_GRID_SP_PAD=<doubles> (read by pmg_grid_create up to round 3; removed from the production build in round 4) shifts the planes of any grid off
their natural alignment.  Development tool; results in DESIGN.md section 9, item 5."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR


def run(nx, ny, nz, reps=60):
    g = GridMCSOR(nx, ny, nz, 10.0)
    b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
    y = g.new_cvec()
    c = g.sample_cvec(b, y, 150, 0xCAFE, 0, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.sample_cvec(b, y, reps, 0xCAFE, c, True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e9 / (nx * ny * nz)


for rep in range(3):
    for dims in ((512, 512, 512), (513, 513, 513), (257, 257, 257), (256, 256, 256)):
        print(f"rep {rep} {dims}: {run(*dims, reps=60 if dims[0] > 300 else 300):.3f} ns/kpt", flush=True)
