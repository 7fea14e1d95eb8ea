#!/usr/bin/env python3
"""Upper bound of what a HIP graph could buy the V-cycle: ONE sample (fixed noise counters -- a timing probe, not a
sampler) is captured into a graph and replayed; eager samples of the same hierarchy are timed beside it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parmgmc_amd import MGMC  # noqa: E402

for n, levels in ((129, 4), (257, 5), (513, 6)):
    mg = MGMC(n, n, n, 10.0, levels).setup()
    b = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(b)
    mg.sample(b, y, 5, seed=1)
    torch.cuda.synchronize()
    reps = 40

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    eager = timed(lambda: mg.sample(b, y, reps, seed=1, counter0=5))
    s = torch.cuda.Stream()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            mg.sample(b, y, 1, seed=1, counter0=5)  # warm-up on the capture stream
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                mg.sample(b, y, 1, seed=1, counter0=5)
        graph = timed(lambda: [g.replay() for _ in range(reps)])
        print(f"{n}^3 {levels} levels: eager {eager:8.1f} us/sample, graph replay {graph:8.1f} us/sample")
    except Exception as e:  # noqa: BLE001
        print(f"{n}^3 {levels} levels: eager {eager:8.1f} us/sample, capture failed: {type(e).__name__}: {str(e)[:200]}")
    del mg, b, y
