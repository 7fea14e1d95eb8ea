#!/usr/bin/env python3
"""BASELINE config 5 on one GPU (development tool): MGMC on a 257^3 grid with k ball observations (low-rank update
on every level), ms per sample next to the plain sampler."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from parmgmc_amd import MGMC


def balls(n, k, seed=0):
    rng = np.random.default_rng(seed)
    xs = np.linspace(0, 1, n)
    B = np.zeros((n ** 3, k))
    for c in range(k):
        ctr, r = rng.uniform(0.2, 0.8, 3), rng.uniform(0.05, 0.12)
        ix = np.nonzero(np.abs(xs - ctr[0]) < r)[0]
        iy = np.nonzero(np.abs(xs - ctr[1]) < r)[0]
        iz = np.nonzero(np.abs(xs - ctr[2]) < r)[0]
        I, J, K = np.meshgrid(ix, iy, iz, indexing="ij")
        inside = (xs[I] - ctr[0]) ** 2 + (xs[J] - ctr[1]) ** 2 + (xs[K] - ctr[2]) ** 2 < r * r
        rows = (I + n * (J + n * K))[inside]
        B[rows, c] = (1.0 / (n - 1) ** 3) / (4 / 3 * np.pi * r ** 3)
    return B, np.full(k, 1e4)


n, levels = int(sys.argv[1]) if len(sys.argv) > 1 else 257, 5
import os
for k in ([int(os.environ["K"])] if "K" in os.environ else (0, 3, 17)):
    mg = MGMC(n, n, n, 10.0, levels)
    if k:
        B, S = balls(n, k)
        t0 = time.perf_counter()
        mg.set_lowrank(B, S)
        del B
    t0 = time.perf_counter()
    mg.setup()
    torch.cuda.synchronize()
    ts = time.perf_counter() - t0
    b = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
    y = torch.zeros(n ** 3, dtype=torch.float64, device="cuda")
    ctr = mg.sample(b, y, 3, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mg.sample(b, y, 10, seed=1, counter0=ctr)
    torch.cuda.synchronize()
    print(f"{n}^3, {levels} levels, k = {k:2d}: setup {ts:6.2f} s, {(time.perf_counter() - t0) * 100:8.3f} ms/sample, finite {bool(torch.isfinite(y).all())}", flush=True)
    mg.destroy()
