#!/usr/bin/env python3
"""Noisy colour sweep on 512 x 512 x nz grids (the per-GPU share of the headline grid on 8 / 4 / 2 / 1 devices, without the
halo): us per sweep and GB/s -- how much of the one-device efficiency a slab keeps (development tool)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR
from tools.kbench import timeit

for nz in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]:
    g = GridMCSOR(512, 512, nz, 10.0)
    b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
    y = g.new_cvec()
    t = timeit(lambda: g.sample_cvec(b, y, 1, 0xCAFE, 0), 200)
    td = timeit(lambda: g.apply_cvec(b, y), 200)
    byts = 24 * 512 * 512 * nz
    print(f"512x512x{nz}: noisy sweep {t * 1e3:7.1f} us ({byts / t / 1e6:7.1f} GB/s, {t * 1e3 / 2:6.1f} us per launch)   deterministic {td * 1e3:7.1f} us ({byts / td / 1e6:7.1f} GB/s)", flush=True)
    del g, b, y
