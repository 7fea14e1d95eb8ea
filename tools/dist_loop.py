#!/usr/bin/env python3
"""Repeat the z-slab V-cycle line of bench.py on N ranks sharing ONE GPU (lifecycle + flag protocol under time slicing):
PMG_BENCH_SHARE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/dist_loop.py [reps] [n] [levels]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 513
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
for it in range(reps):
    t0 = time.perf_counter()
    try:
        r = bench.mgmc_dist_secondary(rank, world, "ipc", True, n, levels, its=10)
        msg = f"ok {r['ms_per_sample']:.2f} ms/sample finite {r['finite']}"
    except Exception as e:  # noqa: BLE001
        msg = f"FAILED {type(e).__name__}: {str(e)[:400]}"
    print(f"[rank {rank}] iteration {it}: {msg} ({time.perf_counter() - t0:.1f} s)", flush=True)
    if msg.startswith("FAILED"):
        break
dist.barrier()
dist.destroy_process_group()
