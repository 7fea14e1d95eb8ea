#!/usr/bin/env python3
"""Residual + restriction on the grid level: the two kernels against the fused one (PMG_GRID_RR_CHUNK = coarse planes per
chunk, read once per process; tools/rr_sweep.sh walks it), events over 20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parmgmc_amd import MGMC  # noqa: E402

tag = f"chunk {os.environ.get('PMG_GRID_RR_CHUNK', 'auto')}"
for n, levels in ((65, 3), (129, 4), (257, 5), (513, 6)):
    mg = MGMC(n, n, n, 10.0, levels).setup()
    top = levels - 1
    _, ld, _ = mg.level_layout(top)
    _, ldc, offc = mg.level_layout(top - 1)
    b = torch.randn(ld, dtype=torch.float64, device="cuda")
    x = torch.randn(ld, dtype=torch.float64, device="cuda")
    r = torch.zeros(ld, dtype=torch.float64, device="cuda")
    c1 = torch.zeros(ldc, dtype=torch.float64, device="cuda")
    c2 = torch.zeros(ldc, dtype=torch.float64, device="cuda")

    def two():
        mg.level_residual(top, b, x, r)
        mg.level_restrict(top, r, c1)

    def one():
        mg.level_residual_restrict(top, b, x, c2)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    t2, t1 = timed(two), timed(one)
    same = bool(torch.equal(c1, c2))
    print(f"{n}^3 [{tag}]: residual + restrict {t2:7.1f} us, fused {t1:7.1f} us, same bits {same}", flush=True)
    del mg, b, x, r, c1, c2
