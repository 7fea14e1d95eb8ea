#!/usr/bin/env python3
"""Line stride of the colour arrays on the multigrid sizes 2^k+1 (development tool): top-level sweep (ns per 1000 point
updates, noisy omega = 1) and whole V-cycle sample for PMG_GRID_SX_ALIGN = 16 (whole 128-byte lines), 8, 4, 2 (tightest
even stride); three runs each, interleaved so that clock drift hits every variant alike."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import MGMC, GridMCSOR

sizes = [int(a) for a in sys.argv[1:]] or [129, 257, 513]


def ev_time(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(reps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in sizes:
    levels, m = 1, n
    while m > 25 and (m - 1) % 2 == 0:
        m, levels = (m - 1) // 2 + 1, levels + 1
    aligns = [16, 2] if n not in (129, 257, 513) else [16, 8, 4, 2]
    res = {a: {"sweep": [], "cycle": []} for a in aligns}
    for rep in range(3):
        for a in aligns:
            os.environ["PMG_GRID_SX_ALIGN"] = str(a)
            g = GridMCSOR(n, n, n, 10.0)
            b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
            y = g.new_cvec()
            c = [0]

            def sw(r):
                c[0] = g.sample_cvec(b, y, r, 0xCAFE, c[0], True)

            sw(100 if rep else 300)
            res[a]["sweep"].append(ev_time(sw, 60) * 1e9 / n ** 3)  # ms per sweep -> ns per 1000 points
            del g, b, y
            mg = MGMC(n, n, n, 10.0, levels).setup()
            bb = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
            yy = torch.zeros(n ** 3, dtype=torch.float64, device="cuda")
            cc = [mg.sample(bb, yy, 5, seed=1)]

            def cy(r):
                cc[0] = mg.sample(bb, yy, r, seed=1, counter0=cc[0])

            res[a]["cycle"].append(ev_time(cy, 20 if n < 500 else 8))
            mg.destroy()
            del mg, bb, yy
            torch.cuda.empty_cache()
    for a in aligns:
        sx = ((n + 1) // 2 + a - 1) // a * a
        print(f"n={n} align={a:2d} sx={sx:4d}  sweep ns/kpt {' '.join(f'{v:6.3f}' for v in res[a]['sweep'])}   cycle ms {' '.join(f'{v:7.4f}' for v in res[a]['cycle'])}", flush=True)
