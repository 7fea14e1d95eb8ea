"""Pins the CPU oracle against the known-answer tests the reference itself holds for the hot path
(SURVEY.md section 8c).  CPU only."""
import numpy as np
import pytest

import oracle as O


def test_philox_random123_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, out in kat:
        assert list(O.philox4x32_10(ctr, key)) == out


def test_problems_c_3x3_by_hand():
    """reference src/problems.c:14-75 on 3x3, kappa=10: h2 = 1/((3-1)*(3-1)) = 0.25; corner diag
    100+0.25+0.25, edge 100+3*0.25, centre 100+4*0.25; columns sorted."""
    A = O.shifted_laplace(3, 3, 1, 10.0)
    D = A.dense()
    assert np.array_equal(np.diag(D), [100.5, 100.75, 100.5, 100.75, 101.0, 100.75, 100.5, 100.75, 100.5])
    assert D[0, 1] == -0.25 and D[0, 3] == -0.25 and D[0, 4] == 0 and D[4, 1] == D[4, 3] == D[4, 5] == D[4, 7] == -0.25
    assert np.array_equal(D, D.T)
    assert list(A.rowptr) == [0, 3, 7, 10, 14, 19, 23, 26, 30, 33]
    assert list(A.colidx[:7]) == [0, 1, 3, 0, 1, 2, 4]
    # the integer-product quirk: mx is used for both directions (problems.c:24)
    B = O.shifted_laplace(5, 3, 1, 1.0)
    assert B.dense()[0, 1] == -1.0 / 16 and B.dense()[0, 5] == -1.0 / 16


def test_problems_c_repeated_addition_not_multiplication():
    # diag = ((((k2 + h)+h)+h)+h): with h = 1/81 repeated addition differs from k2 + 4h in the last bit
    A = O.shifted_laplace(10, 10, 1, 1.0)
    h = 1.0 / 81
    d = 1.0
    for _ in range(4):
        d += h
    assert A.dense()[11, 11] == d


def test_ex6_matrix():
    A = O.ex6_matrix(10, 1e-4).dense()
    assert A[0, 0] == 2 + 1e-4 and A[1, 1] == 3 + 1e-4 and A[11, 11] == 4 + 1e-4 and A[0, 1] == -1 and A[0, 10] == -1


@pytest.mark.parametrize("colors", ["single", "redblack"])
def test_ex5_symmetric_is_forward_then_backward(colors):
    """reference examples/ex5.c:53-70: 9x9, kappa = 1, random b, x: ||fwd;bwd - sym||_2 < 1e-15."""
    A = O.shifted_laplace(9, 9, 1, 1.0)
    rng = np.random.default_rng(5)
    b, x = rng.random(81), rng.random(81)
    col = O.coloring_single(81) if colors == "single" else O.coloring_redblack(9, 9)
    fb = O.mcsor_apply(A, col, b, O.mcsor_apply(A, col, b, x, 1.0, O.SOR_FORWARD), 1.0, O.SOR_BACKWARD)
    s = O.mcsor_apply(A, col, b, x, 1.0, O.SOR_SYMMETRIC)
    assert np.linalg.norm(fb - s) < 1e-15


def test_ex5_on_four_ranks():
    """same check through the MPIAIJ kernel (make check-par runs ex5 with -np 4)."""
    A = O.shifted_laplace(9, 9, 1, 1.0)
    rng = np.random.default_rng(6)
    b, x = rng.random(81), rng.random(81)
    col = O.coloring_redblack(9, 9)
    own = [0, 20, 41, 61, 81]
    f = O.mcsor_sweep_domains(A, own, col, b, x, 1.0, O.SOR_FORWARD)
    fb = O.mcsor_sweep_domains(A, own, col, b, f, 1.0, O.SOR_BACKWARD)
    s = O.mcsor_sweep_domains(A, own, col, b, x, 1.0, O.SOR_SYMMETRIC)
    assert np.linalg.norm(fb - s) < 1e-15
    # true parallel Gauss-Seidel (src/mc_sor.c:30-33): same as the one-rank multicolour sweep up to the
    # different summation order of the MPI kernel (sum starts at 0, b added last, :328-334)
    ref = O.mcsor_apply(A, col, b, x, 1.0, O.SOR_SYMMETRIC)
    assert np.allclose(s, ref, rtol=0, atol=1e-15)


def test_box_muller_pairing_and_tail():
    """reference src/parmgmc.c:99-110: entries (i, i+1) share (u1,u2); odd n keeps only the cosine."""
    u = np.array([0.3, 0.7, 0.5, 0.25, 0.9, 0.1])
    z = O.box_muller_vec(5, u)
    r = np.sqrt(-2 * np.log(u[0::2]))
    th = 2 * np.pi * u[1::2]
    assert np.allclose(z, [r[0] * np.cos(th[0]), r[0] * np.sin(th[0]), r[1] * np.cos(th[1]), r[1] * np.sin(th[1]), r[2] * np.cos(th[2])], rtol=1e-15)


@pytest.mark.parametrize("sweep,colors", [(O.SOR_FORWARD, "redblack"), (O.SOR_BACKWARD, "redblack"), (O.SOR_SYMMETRIC, "redblack"), (O.SOR_FORWARD, "single")])
def test_ex1_sample_mean_converges(sweep, colors):
    """reference examples/ex1.c:83-135: 9x9, kappa = 10, b = 1, x0 = 0; relative L2 error of the running
    mean < 0.02 after 1e6 samples / 1e4 burn-in.  Here A ~ 100 I, so a component has mean 0.01 and standard
    deviation 0.1 and the relative error of the running mean is ~10/sqrt(n): 0.01 at the reference's 1e6
    samples (bound 0.02).  Budget scaled to 4e4 samples / 400 burn-in, bound scaled as 1/sqrt(n):
    0.02*sqrt(1e6/4e4) = 0.1 (expected error 0.05)."""
    A = O.shifted_laplace(9, 9, 1, 10.0)
    n = 81
    b = np.ones(n)
    col = O.coloring_redblack(9, 9) if colors == "redblack" else O.coloring_single(n)
    mean = np.zeros(n)

    def cb(it, y):  # SampleCallback of ex1.c:57-64
        mean[:] = mean * (it / (it + 1.0)) + y / (it + 1)

    y = O.gibbs_samples(A, col, b, np.zeros(n), 400, lambda d: O.noise_rows(n, 0xCAFE, d), 1.0, sweep)
    O.gibbs_samples(A, col, b, y, 40000, lambda d: O.noise_rows(n, 0xCAFE, 10_000 + d), 1.0, sweep, callback=cb)
    ex = np.linalg.solve(A.dense(), b)
    assert np.linalg.norm(mean - ex) / np.linalg.norm(ex) < 0.02 * np.sqrt(1e6 / 4e4)


def test_stats_covariance_metric_on_exact_samples():
    """reference src/stats.c:63-117: unbiased sample covariance across chains vs dense inverse, relative
    Frobenius norm.  Exact Cholesky samples of the ex6 matrix (10x10, kappa 1e-4 -> use 1 for conditioning)
    must converge like 1/sqrt(chains)."""
    A = O.ex6_matrix(6, 1.0)
    n = A.n
    L = O.potrf_lower(A.dense())
    errs = []
    for chains in (200, 3200):
        S = np.stack([O.chol_sample(L, np.zeros(n), O.noise_rows(n, 1, c)) for c in range(chains)])
        errs.append(O.covariance_error(A, S))
    assert errs[1] < errs[0] / 2.5 and errs[1] < 0.15


def test_chol_sampler_algebra():
    """reference src/pc_chols.c:284-288: y = L^-T (L^-1 x + xi), so E y = A^-1 x and cov = A^-1."""
    A = O.shifted_laplace(4, 4, 1, 3.0)
    n = A.n
    L = O.potrf_lower(A.dense())
    assert np.allclose(L @ L.T, A.dense(), rtol=1e-14)
    x = np.arange(1.0, n + 1)
    y0 = O.chol_sample(L, x, np.zeros(n))
    assert np.allclose(y0, np.linalg.solve(A.dense(), x), rtol=1e-13)
    e = np.eye(n)
    Nmap = np.stack([O.chol_sample(L, np.zeros(n), e[i]) for i in range(n)], 1)
    assert np.allclose(Nmap @ Nmap.T, np.linalg.inv(A.dense()), rtol=1e-12, atol=1e-15)
