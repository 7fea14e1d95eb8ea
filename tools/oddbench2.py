#!/usr/bin/env python3
"""one grid shape, noisy + deterministic sweep (development tool): oddbench2.py nx ny nz"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR
from tools.kbench import timeit

nx, ny, nz = (int(a) for a in sys.argv[1:4])
g = GridMCSOR(nx, ny, nz, 10.0)
b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
y = g.new_cvec()
t = timeit(lambda: g.sample_cvec(b, y, 1, 0xCAFE, 0), 300)
td = timeit(lambda: g.apply_cvec(b, y), 300)
byts = 24 * nx * ny * nz
print(f"{nx}x{ny}x{nz}: noisy {t * 1e3:7.1f} us ({byts / t / 1e6:7.1f} GB/s)   deterministic {td * 1e3:7.1f} us ({byts / td / 1e6:7.1f} GB/s)", flush=True)
