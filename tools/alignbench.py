#!/usr/bin/env python3
"""What costs the padded 2^k+1 layouts their 13-15 %?  512^3 sweep (ns per 1000 point updates) with the lines kept
contiguous inside a plane but the PLANES shifted off their natural alignment (PMG_GRID_SP_PAD doubles behind every plane),
against padded lines (PMG_GRID_SX_ALIGN cannot pad 256; the padded-line number is 513^3's).  Development tool."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR


def run(n, reps=60):
    g = GridMCSOR(n, n, n, 10.0)
    b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
    y = g.new_cvec()
    c = g.sample_cvec(b, y, 150, 0xCAFE, 0, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.sample_cvec(b, y, reps, 0xCAFE, c, True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e9 / n ** 3


for rep in range(3):
    for n, pad in ((512, 0), (512, 16), (512, 32), (512, 64), (512, 256), (512, 2048), (513, 0), (513, 48), (513, 240)):
        os.environ["PMG_GRID_SP_PAD"] = str(pad)
        print(f"rep {rep} n={n} sp_pad={pad:5d}: {run(n):.3f} ns/kpt", flush=True)
