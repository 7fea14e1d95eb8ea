#!/usr/bin/env python3
"""Which direction's 2^k+1 costs the colour sweep its efficiency: nx x ny x nz with each extent 256 or 257 (development tool)."""
import itertools
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR
from tools.kbench import timeit

base = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for dx, dy, dz in itertools.product((0, 1), repeat=3):
    nx, ny, nz = base + dx, base + dy, base + dz
    g = GridMCSOR(nx, ny, nz, 10.0)
    b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
    y = g.new_cvec()
    t = timeit(lambda: g.sample_cvec(b, y, 1, 0xCAFE, 0), 300)
    td = timeit(lambda: g.apply_cvec(b, y), 300)
    byts = 24 * nx * ny * nz
    print(f"{nx}x{ny}x{nz}: noisy {t * 1e3:7.1f} us ({byts / t / 1e6:7.1f} GB/s)   deterministic {td * 1e3:7.1f} us ({byts / td / 1e6:7.1f} GB/s)", flush=True)
    del g, b, y
