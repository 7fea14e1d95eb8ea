#!/usr/bin/env python3
"""V-cycle time per sample at 257^3 (5 levels) and 513^3 (6 levels); run under `rocprofv3 --kernel-trace --stats`
for the per-kernel split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parmgmc_amd import MGMC  # noqa: E402

only = os.environ.get("PMG_VC_ONLY")
for n, levels, reps in ((257, 5, 20), (513, 6, 8)):
    if only and int(only) != n:
        continue
    mg = MGMC(n, n, n, 10.0, levels).setup()
    b = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(b)
    c = mg.sample(b, y, 3, seed=1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    mg.sample(b, y, reps, seed=1, counter0=c)
    torch.cuda.synchronize()
    print(f"{n}^3 {levels}-level V-cycle {(time.perf_counter() - t) / reps * 1e3:7.3f} ms/sample", flush=True)
    del mg, b, y
