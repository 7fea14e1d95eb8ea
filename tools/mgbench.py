#!/usr/bin/env python3
"""V-cycle (MGMC) sample timing (development tool): python tools/mgbench.py --n 257 --levels 6"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import MGMC


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=257)
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--its", type=int, default=20)
    a = ap.parse_args()
    n = a.n
    t0 = time.time()
    mg = MGMC(n, n, n, 10.0, a.levels).setup()
    print(f"setup {time.time()-t0:.1f} s", flush=True)
    b = torch.ones(n ** 3, dtype=torch.float64, device="cuda")
    y = torch.zeros(n ** 3, dtype=torch.float64, device="cuda")
    mg.sample(b, y, 2, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mg.sample(b, y, a.its, seed=1, counter0=2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.its
    print(f"n={n} levels={a.levels}: {ms:.3f} ms per MGMC sample ({1e3/ms:.1f} samples/s), model 160 B/unknown -> {160*n**3/ms/1e6:.0f} GB/s; finite={bool(torch.isfinite(y).all())}")


if __name__ == "__main__":
    main()
