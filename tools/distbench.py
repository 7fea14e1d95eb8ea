#!/usr/bin/env python3
"""Per-rank cost model of the 8-GPU run on ONE GPU (development tool): a 512x512x64 slab swept (a) by the plain
sampler and (b) by the C/RCCL driver in loopback mode (same kernels split boundary/interior + ncclSend/Recv to self)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR
from parmgmc_amd.dist import IpcSlabDriver, RcclSlabDriver


def timeit(fn, reps=50):
    fn(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(reps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


import os
for nz in ([64] if os.environ.get('DISTBENCH_QUICK') else (64, 128, 256)):
    g = GridMCSOR(512, 512, nz, 10.0)
    b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
    y = g.new_cvec()
    t_plain = timeit(lambda its: g.sample_cvec(b, y, its, 1, 0))
    drv = RcclSlabDriver(g, 0, 1, loopback=True)
    t_dist = timeit(lambda its: drv.sample_cvec(b, y, its, True, 1, 1, 0))
    ipc = IpcSlabDriver(g, 0, 1, loopback=True)
    t_ipc = timeit(lambda its: ipc.sample_cvec(b, y, its, True, 1, 1, 0))
    print(f"slab 512x512x{nz}: plain {t_plain*1e6:7.1f} us/sweep   rccl-loopback {t_dist*1e6:7.1f}   ipc-loopback {t_ipc*1e6:7.1f} us/sweep   (512^3 single GPU ~650 us -> ideal {650*nz/512:5.1f})", flush=True)
