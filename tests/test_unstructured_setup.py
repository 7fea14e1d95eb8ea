"""Host-side set-up of the unstructured path (parmgmc_amd/unstructured.py): mesh reader, vectorised P1 assembly,
uniform refinement, aggregation hierarchy.  CPU only."""
from pathlib import Path

import numpy as np
import scipy.sparse as sp

from parmgmc_amd.unstructured import assemble_p1, build_hierarchy, read_gmsh41_triangles, refine_uniform

MSH = Path(__file__).resolve().parent / "golden" / "lshape.msh"


def assemble_p1_loop(xy, tris, kappa):
    """element-by-element reference of the vectorised assembly"""
    n = len(xy)
    A = np.zeros((n, n))
    for t in tris:
        p = xy[t]
        B = np.array([p[1] - p[0], p[2] - p[0]]).T
        area = 0.5 * abs(np.linalg.det(B))
        G = np.linalg.inv(B).T @ np.array([[-1.0, 1.0, 0.0], [-1.0, 0.0, 1.0]])
        E = kappa ** 2 * area / 12.0 * (np.ones((3, 3)) + np.eye(3)) + area * (G.T @ G)
        A[np.ix_(t, t)] += E
    return A


def test_reader_and_assembly():
    xy, tris = read_gmsh41_triangles(MSH)
    assert xy.shape == (408, 2) and tris.shape == (734, 3)
    A = assemble_p1(xy, tris, 1.3)
    ref = assemble_p1_loop(xy, tris, 1.3)
    assert np.abs(A.toarray() - ref).max() < 1e-13 * np.abs(ref).max()
    assert abs(A - A.T).max() < 1e-14 and np.linalg.eigvalsh(ref).min() > 0
    # 1^T M 1 = area of the mesh, K 1 = 0
    M = assemble_p1(xy, tris, 1.0) - assemble_p1(xy, tris, 0.0)
    p = xy[tris]
    e1, e2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    area = 0.5 * np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]).sum()
    assert abs(M.sum() - area) < 1e-12 * area
    assert np.abs(assemble_p1(xy, tris, 0.0) @ np.ones(408)).max() < 1e-12


def test_uniform_refinement_conserves_area_and_conformity():
    xy, tris = read_gmsh41_triangles(MSH)
    area0 = assemble_p1(xy, tris, 1.0).sum() - assemble_p1(xy, tris, 0.0).sum()
    xy2, t2 = refine_uniform(xy, tris)
    assert len(t2) == 4 * len(tris)
    # Euler: V - E + F is unchanged; new vertices = old edges
    e = np.sort(np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]]), axis=1)
    assert len(xy2) == len(xy) + len(np.unique(e, axis=0))
    area1 = assemble_p1(xy2, t2, 1.0).sum() - assemble_p1(xy2, t2, 0.0).sum()
    assert abs(area1 - area0) < 1e-12 * area0
    # conforming: every interior edge of the fine mesh is shared by exactly two triangles
    e2 = np.sort(np.concatenate([t2[:, [0, 1]], t2[:, [1, 2]], t2[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e2, axis=0, return_counts=True)
    assert set(cnt.tolist()) <= {1, 2}
    A = assemble_p1(xy2, t2, 1.0)
    assert np.linalg.eigvalsh(A.toarray()).min() > 0


def test_aggregation_hierarchy_shapes():
    xy, tris = read_gmsh41_triangles(MSH)
    for _ in range(2):
        xy, tris = refine_uniform(xy, tris)
    A = assemble_p1(xy, tris, 1.0)
    ops, ps = build_hierarchy(A, coarse_max=200)
    assert len(ops) >= 3 and ps[0] is None
    sizes = [len(o[0]) - 1 for o in ops]
    assert sizes == sorted(sizes) and sizes[0] <= 200 and sizes[-1] == A.shape[0]
    for l in range(1, len(ops)):
        P = sp.csr_matrix((ps[l][2], ps[l][1], ps[l][0]), shape=(sizes[l], sizes[l - 1]))
        assert np.array_equal(np.asarray(P.sum(1)).ravel(), np.ones(sizes[l]))  # every node in exactly one aggregate
        Af = sp.csr_matrix((ops[l][2], ops[l][1], ops[l][0]), shape=(sizes[l], sizes[l]))
        Ac = sp.csr_matrix((ops[l - 1][2], ops[l - 1][1], ops[l - 1][0]), shape=(sizes[l - 1], sizes[l - 1]))
        assert abs(P.T @ Af @ P - Ac).max() < 1e-12 * abs(Ac).max()
