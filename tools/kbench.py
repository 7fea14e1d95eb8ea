#!/usr/bin/env python3
"""Kernel micro-benchmark for the grid colour sweep (development tool, not the headline bench)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR


def timeit(fn, reps, warm=100):
    for _ in range(warm):  # past the clock transient after idle (the first ~100 launches run up to 40 % slower)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[512])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--omega", type=float, nargs="+", default=[1.0, 1.2])
    ap.add_argument("--mode", choices=["both", "det", "noisy"], default="both")
    ap.add_argument("--no-copy", action="store_true")
    a = ap.parse_args()
    for n in a.n:
        g = GridMCSOR(n, n, n, 10.0)
        b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
        y = g.new_cvec()
        N = n ** 3
        for om in a.omega:
            g.set_omega(om)
            t_det = timeit(lambda: g.apply_cvec(b, y), a.reps) if a.mode != "noisy" else float("nan")
            t_noisy = timeit(lambda: g.sample_cvec(b, y, 1, 0xCAFE, 0), a.reps) if a.mode != "det" else float("nan")
            byts = (24 if om == 1.0 else 32) * N
            print(f"n={n} omega={om}: deterministic sweep {t_det*1e3:8.1f} us ({byts/t_det/1e6:7.1f} GB/s)   noisy sweep {t_noisy*1e3:8.1f} us ({byts/t_noisy/1e6:7.1f} GB/s)", flush=True)
        if a.no_copy:
            continue
        # plain copy ceiling
        x = torch.empty(N, dtype=torch.float64, device="cuda")
        z = torch.empty(N, dtype=torch.float64, device="cuda")
        t = timeit(lambda: z.copy_(x), a.reps)
        print(f"n={n} torch copy {t*1e3:8.1f} us ({16*N/t/1e6:7.1f} GB/s)")
        del g, b, y, x, z


if __name__ == "__main__":
    main()
