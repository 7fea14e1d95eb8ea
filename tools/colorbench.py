#!/usr/bin/env python3
"""Sliced-ELL sweep time at the config-4 bench size for colourings with fewer classes than first-fit (user colourings through
the C-ABI): DSATUR, and DSATUR with its small classes dissolved by Kempe-chain swaps.  Development tool."""
import heapq
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from parmgmc_amd import MCSOR
from parmgmc_amd.unstructured import assemble_p1, read_gmsh41_triangles, refine_uniform


def dsatur(ind, idx):
    n = len(ind) - 1
    col = -np.ones(n, np.int64)
    deg = np.diff(ind) - 1
    sat = [set() for _ in range(n)]
    heap = [(0, -int(deg[i]), i) for i in range(n)]
    heapq.heapify(heap)
    done = 0
    while done < n:
        s, d, i = heapq.heappop(heap)
        if col[i] >= 0 or -s != len(sat[i]):
            continue
        c = 0
        while c in sat[i]:
            c += 1
        col[i] = c
        done += 1
        for j in idx[ind[i]:ind[i + 1]]:
            if j != i and col[j] < 0 and c not in sat[j]:
                sat[j].add(c)
                heapq.heappush(heap, (-len(sat[j]), -int(deg[j]), j))
    return col


def kempe_reduce(ind, idx, col, keep):
    """try to recolour every vertex of the classes >= keep with a colour < keep: directly, or after swapping the colours a, b
    on the (a, b)-component of one neighbour (Kempe chain) when that frees a for the vertex"""
    col = col.copy()
    for v in np.flatnonzero(col >= keep):
        nb = [j for j in idx[ind[v]:ind[v + 1]] if j != v]
        used = {int(col[j]) for j in nb}
        free = [c for c in range(keep) if c not in used]
        if free:
            col[v] = free[0]
            continue
        ok = False
        for a in range(keep):
            na = [j for j in nb if col[j] == a]
            if len(na) != 1:
                continue
            for b_ in range(keep):
                if b_ == a:
                    continue
                # component of na[0] in the subgraph of colours a, b
                comp, stack = {na[0]}, [na[0]]
                while stack and len(comp) < 20000:
                    u = stack.pop()
                    for w in idx[ind[u]:ind[u + 1]]:
                        if w not in comp and (col[w] == a or col[w] == b_):
                            comp.add(w)
                            stack.append(w)
                if len(comp) >= 20000 or any(col[j] == b_ and j in comp for j in nb):
                    continue
                for u in comp:
                    col[u] = b_ if col[u] == a else a
                col[v] = a
                ok = True
                break
            if ok:
                break
    return col


def timed(mc, b, y, reps=200):
    mc.sample(b, y, 20, seed=1, counter0=0, scaled=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mc.sample(b, y, reps, seed=1, counter0=20, scaled=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


refine = int(sys.argv[1]) if len(sys.argv) > 1 else 5
xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent.parent / "tests" / "golden" / "lshape.msh")
for _ in range(refine):
    xy, tris = refine_uniform(xy, tris)
A = assemble_p1(xy, tris, 1.0)
n = A.shape[0]
b = torch.ones(n, dtype=torch.float64, device="cuda")
y = torch.zeros(n, dtype=torch.float64, device="cuda")
mc = MCSOR(A.indptr, A.indices, A.data).setup()
print(f"{n} rows; first-fit: classes {np.bincount(mc.get_coloring())}, {timed(mc, b, y):.1f} us per sweep", flush=True)
t0 = time.time()
cd = dsatur(A.indptr, A.indices)
print(f"DSATUR ({time.time() - t0:.0f} s): classes {np.bincount(cd)}", flush=True)
mcd = MCSOR(A.indptr, A.indices, A.data, user_colors=cd.astype(np.int32)).setup()
print(f"  {timed(mcd, b, y):.1f} us per sweep", flush=True)
for keep in (4,):
    t0 = time.time()
    ck = kempe_reduce(A.indptr, A.indices, cd, keep)
    print(f"DSATUR + Kempe to {keep} ({time.time() - t0:.0f} s): classes {np.bincount(ck)}", flush=True)
    if ck.max() < cd.max():
        _, ck = np.unique(ck, return_inverse=True)
        mck = MCSOR(A.indptr, A.indices, A.data, user_colors=ck.astype(np.int32)).setup()
        print(f"  {timed(mck, b, y):.1f} us per sweep", flush=True)
