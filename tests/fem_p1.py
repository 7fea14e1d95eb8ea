"""Test-side helpers for BASELINE config 4 (unstructured AIJ path): a Gmsh 4.1 ASCII reader, P1 assembly of
kappa^2 M + K on triangles -- the matrix the reference obtains from PETSc's DMPlex/PetscFE (src/ms.c:109-164;
PETSc-side, parity unpinned) -- and a plain greedy aggregation that stands in for PETSc GAMG's
(third party) to produce interpolation operators.  Test infrastructure only."""
import numpy as np
import scipy.sparse as sp


def read_gmsh41_triangles(path):
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
        dim, _, etype, nb = (int(x) for x in lines[i].split())
        if etype == 2:  # 3-node triangle
            for q in range(nb):
                tris.append([int(v) - 1 for v in lines[i + 1 + q].split()[1:4]])
        i += 1 + nb
    return xy, np.array(tris)


def assemble_p1(xy, tris, kappa):
    n = len(xy)
    rows, cols, vals = [], [], []
    for t in tris:
        p = xy[t]
        B = np.array([p[1] - p[0], p[2] - p[0]]).T
        area = 0.5 * abs(np.linalg.det(B))
        G = np.linalg.inv(B).T @ np.array([[-1.0, 1.0, 0.0], [-1.0, 0.0, 1.0]])
        K = area * (G.T @ G)
        M = area / 12.0 * (np.ones((3, 3)) + np.eye(3))
        E = kappa ** 2 * M + K
        for a in range(3):
            for b in range(3):
                rows.append(t[a]); cols.append(t[b]); vals.append(E[a, b])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A.sort_indices()
    return A


def greedy_aggregation(A):
    """plain (unsmoothed) aggregation: every unaggregated node grabs its unaggregated neighbours."""
    n = A.shape[0]
    agg = -np.ones(n, dtype=np.int64)
    na = 0
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = [j for j in A.indices[A.indptr[i]:A.indptr[i + 1]] if agg[j] < 0]
        for j in [i] + nb:
            agg[j] = na
        na += 1
    P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
    P.sort_indices()
    return P
