#!/usr/bin/env python3
"""Is the V-cycle host-bound?  Host time to ENQUEUE `its` samples (the C call returns when everything is queued) against
the device time of the same samples: a HIP graph can only help where the first exceeds the second."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parmgmc_amd import MGMC  # noqa: E402

for n, levels in ((129, 4), (257, 5), (513, 6)):
    mg = MGMC(n, n, n, 10.0, levels).setup()
    b = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(b)
    c = mg.sample(b, y, 5, seed=1)
    torch.cuda.synchronize()
    its = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    mg.sample(b, y, its, seed=1, counter0=c)
    e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dev = e0.elapsed_time(e1) * 1e-3
    print(f"{n}^3 {levels} levels: host enqueue {host / its * 1e6:7.1f} us/sample, device {dev / its * 1e6:7.1f} us/sample")
    del mg, b, y
