#!/usr/bin/env python3
"""Latency of the generic neighbour exchange (pmg_dist_exchange) on one GPU in loopback mode (development tool)."""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import GridMCSOR
from parmgmc_amd.capi import check, lib
from parmgmc_amd.dist import IpcSlabDriver, RcclSlabDriver
from parmgmc_amd.wrappers import _ptr, _stream

g = GridMCSOR(257, 257, 32, 10.0)
for name, drv in (("ipc", IpcSlabDriver(g, 0, 1, loopback=True)), ("rccl", RcclSlabDriver(g, 0, 1, loopback=True))):
    for n in (16641, 66049, 2 * 37008):
        send = torch.ones((2, n), dtype=torch.float64, device="cuda")
        recv = torch.zeros((2, n), dtype=torch.float64, device="cuda")
        P, I64 = C.c_void_p * 1, C.c_int64 * 1
        nn = I64(n)
        args = (drv._h, 1, P(_ptr(send[0])), nn, P(_ptr(recv[0])), nn, P(_ptr(send[1])), nn, P(_ptr(recv[1])), nn, _stream())
        for _ in range(10):
            check(lib.pmg_dist_exchange(*args))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            check(lib.pmg_dist_exchange(*args))
        torch.cuda.synchronize()
        print(f"{name}: exchange of 2 x {n} doubles: {(time.perf_counter() - t) / 200 * 1e6:7.1f} us", flush=True)
