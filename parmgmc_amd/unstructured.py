"""Host-side set-up for the unstructured (MATAIJ) path -- BASELINE config 4, SURVEY 8 f-2.

The reference gets all of this from PETSc: the mesh from DMPlex (Gmsh reader, `-dm_refine`), the P1 matrix
kappa^2 M + K from PetscFE (src/ms.c:87-105,109-164) and the hierarchy from PCGAMG (-pc_gamgmc_mg_type gamg,
src/pc_gamgmc.c:398).  None of that is ParMGMC code and none of it is pinned by a reference fixture, so these are
own, deterministic algorithms (numpy / scipy, set-up only): a Gmsh 4.1 ASCII reader, vectorised P1 assembly, uniform
("red") refinement, plain aggregation and a hierarchy builder whose output `MGMC.from_hierarchy` takes.  The
sampling itself runs on the device (sliced-ELL multicolour kernels)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def read_gmsh41_triangles(path):
    """nodes (n, 2) and 3-node triangles (m, 3, zero-based) of a Gmsh 4.1 ASCII file"""
    lines = open(path).read().split("\n")
    i = lines.index("$Nodes") + 1
    nblocks, nnodes = (int(x) for x in lines[i].split()[:2])
    i += 1
    xy = np.zeros((nnodes, 2))
    for _ in range(nblocks):
        _, _, _, nb = (int(x) for x in lines[i].split())
        tags = [int(lines[i + 1 + q]) for q in range(nb)]
        for q, t in enumerate(tags):
            xy[t - 1] = [float(v) for v in lines[i + 1 + nb + q].split()[:2]]
        i += 1 + 2 * nb
    i = lines.index("$Elements") + 1
    nblocks = int(lines[i].split()[0])
    i += 1
    tris = []
    for _ in range(nblocks):
        _dim, _, etype, nb = (int(x) for x in lines[i].split())
        if etype == 2:  # 3-node triangle
            for q in range(nb):
                tris.append([int(v) - 1 for v in lines[i + 1 + q].split()[1:4]])
        i += 1 + nb
    return xy, np.array(tris)


def assemble_p1(xy, tris, kappa):
    """kappa^2 M + K for P1 elements on triangles (consistent mass), CSR with sorted columns"""
    n = len(xy)
    p = xy[tris]  # (m, 3, 2)
    e1, e2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    area = 0.5 * np.abs(det)
    # gradients of the barycentric functions: rows of inv([e1 e2])^T applied to the reference gradients
    inv = np.empty((len(tris), 2, 2))
    inv[:, 0, 0], inv[:, 0, 1], inv[:, 1, 0], inv[:, 1, 1] = e2[:, 1] / det, -e2[:, 0] / det, -e1[:, 1] / det, e1[:, 0] / det
    ref = np.array([[-1.0, 1.0, 0.0], [-1.0, 0.0, 1.0]])
    G = np.einsum("mji,jk->mik", inv, ref)  # (m, 2, 3): gradient of basis function k
    K = area[:, None, None] * np.einsum("mia,mib->mab", G, G)
    M = (area / 12.0)[:, None, None] * (np.ones((3, 3)) + np.eye(3))
    E = kappa ** 2 * M + K
    rows = np.repeat(tris, 3, axis=1).ravel()
    cols = np.tile(tris, (1, 3)).ravel()
    A = sp.csr_matrix((E.ravel(), (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A.sort_indices()
    return A


def refine_uniform(xy, tris):
    """one uniform ("red") refinement: every triangle into four through its edge midpoints (what -dm_refine does)"""
    n = len(xy)
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    e.sort(axis=1)
    uniq, inv = np.unique(e, axis=0, return_inverse=True)
    mid = n + inv.reshape(3, -1).T  # (m, 3): midpoint node of edges 01, 12, 20
    xy2 = np.concatenate([xy, 0.5 * (xy[uniq[:, 0]] + xy[uniq[:, 1]])])
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    ab, bc, ca = mid[:, 0], mid[:, 1], mid[:, 2]
    t2 = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
    return xy2, t2


def greedy_aggregation(A):
    """plain (unsmoothed) aggregation: every unaggregated node grabs its unaggregated neighbours"""
    n = A.shape[0]
    agg = -np.ones(n, dtype=np.int64)
    indptr, indices = A.indptr, A.indices
    na = 0
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = indices[indptr[i]:indptr[i + 1]]
        nb = nb[agg[nb] < 0]
        agg[nb] = na
        agg[i] = na
        na += 1
    P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
    P.sort_indices()
    return P


def build_hierarchy(A, coarse_max: int = 2000, max_levels: int = 12):
    """aggregation hierarchy with Galerkin operators: returns (operators, interpolations) for MGMC.from_hierarchy,
    level 0 = coarsest; stops when a level has at most coarse_max rows"""
    ops, ps = [sp.csr_matrix(A)], []
    while ops[-1].shape[0] > coarse_max and len(ops) < max_levels:
        P = greedy_aggregation(ops[-1])
        Ac = (P.T @ ops[-1] @ P).tocsr()
        Ac.sort_indices()
        ps.append(P)
        ops.append(Ac)
    ops, ps = ops[::-1], ps[::-1]
    operators = [(m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data) for m in ops]
    interpolations = [None] + [(p.indptr.astype(np.int32), p.indices.astype(np.int32), p.data) for p in ps]
    return operators, interpolations
