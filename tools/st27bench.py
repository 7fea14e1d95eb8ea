#!/usr/bin/env python3
"""Times ONE kernel of the class-stencil level 257^3 (first coarse level of the 513^3 hierarchy) through the level
diagnostics of the C-ABI: residual, sweep (deterministic / noisy; includes one device copy of the vector)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parmgmc_amd import MGMC  # noqa: E402

n, levels = int(os.environ.get("N", "513")), int(os.environ.get("LEVELS", "6"))
mg = MGMC(n, n, n, 10.0, levels).setup()
lv = levels - 2
kind, ld, off = mg.level_layout(lv)
g = torch.Generator(device="cuda").manual_seed(1)
b = torch.randn(ld, dtype=torch.float64, device="cuda", generator=g)
x = torch.randn(ld, dtype=torch.float64, device="cuda", generator=g)
r = torch.zeros_like(x)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


nc = (n - 1) // 2 + 1
print(f"level {nc}^3 ({nc**3 * 8 / 1e6:.0f} MB per vector)")
print(f"residual            {timed(lambda: mg.level_residual(lv, b, x, r)):8.1f} us")
print(f"torch r = x + b     {timed(lambda: torch.add(x, b, out=r)):8.1f} us")
xz, bo = torch.zeros_like(x), torch.ones_like(x)
print(f"residual (x=0,b=1)  {timed(lambda: mg.level_residual(lv, bo, xz, r)):8.1f} us")
print(f"copy (d2d)          {timed(lambda: r.copy_(x)):8.1f} us")
print(f"sweep det + copy    {timed(lambda: mg.level_sweep(lv, b, x)):8.1f} us")
print(f"sweep noisy + copy  {timed(lambda: mg.level_sweep(lv, b, x, noisy=True, seed=3, counter=1)):8.1f} us")
print(f"residual (again)    {timed(lambda: mg.level_residual(lv, b, x, r), 200):8.1f} us")
print(f"torch add (again)   {timed(lambda: torch.add(x, b, out=r), 200):8.1f} us")
print(f"sweep noisy (again) {timed(lambda: mg.level_sweep(lv, b, x, noisy=True, seed=3, counter=1), 100):8.1f} us")
big = torch.randn(3 * ld + 3 * 65536, dtype=torch.float64, device="cuda", generator=g)
for o1, o2 in ((0, 0), (512, 1024), (2048, 4096), (8192 + 16, 16384 + 48), (32768, 65536)):
    xb, bb, rb = big[0:ld], big[ld + o1: 2 * ld + o1], big[2 * ld + o2: 3 * ld + o2]
    print(f"residual, vectors at +0, ld+{o1}, 2ld+{o2} doubles: {timed(lambda: mg.level_residual(lv, bb, xb, rb), 50):8.1f} us")
xz = torch.zeros_like(x)
print(f"sweep noisy, x = 0 each time   {timed(lambda: (xz.zero_(), mg.level_sweep(lv, b, xz, noisy=True, seed=3, counter=1)), 100):8.1f} us (incl. zero fill + copy)")
bs = b * 1e-3
print(f"sweep noisy, small b            {timed(lambda: mg.level_sweep(lv, bs, x, noisy=True, seed=3, counter=1), 100):8.1f} us")
