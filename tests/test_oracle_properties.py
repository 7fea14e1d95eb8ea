"""Deterministic properties of the oracle that make the north-star "covariance within 1e-6" testable
without Monte-Carlo noise, the build's own colouring rules, and the golden files."""
from pathlib import Path

import numpy as np
import pytest

import oracle as O

GOLD = Path(__file__).resolve().parent / "golden"


def _maps(A, col, om, sweep):
    n = A.n
    z, e = np.zeros(n), np.eye(n)
    sd = O.sqrtdiag(A, om, True)
    if sweep != O.SOR_SYMMETRIC:
        G = np.stack([O.mcsor_apply(A, col, z, e[i], om, sweep) for i in range(n)], 1)
        N = np.stack([O.mcsor_apply(A, col, e[i] * sd, z, om, sweep) for i in range(n)], 1)
        return G, N

    def sym(y, x1, x2):  # two sweeps, two fresh draws: reference src/pc_mcgibbs.c:172-181
        y = O.mcsor_apply(A, col, x1 * sd, y, om, O.SOR_FORWARD)
        return O.mcsor_apply(A, col, x2 * sd, y, om, O.SOR_BACKWARD)

    G = np.stack([sym(e[i], z, z) for i in range(n)], 1)
    N = np.concatenate([np.stack([sym(z, e[i], z) for i in range(n)], 1), np.stack([sym(z, z, e[i]) for i in range(n)], 1)], 1)
    return G, N


@pytest.mark.parametrize("grid", [(9, 9, 1), (5, 5, 5)])
@pytest.mark.parametrize("om", [1.0, 1.3])
@pytest.mark.parametrize("sweep", [O.SOR_FORWARD, O.SOR_BACKWARD, O.SOR_SYMMETRIC])
@pytest.mark.parametrize("colors", ["redblack", "single"])
def test_stationary_covariance_is_inverse_precision(grid, om, sweep, colors):
    """The chain y <- G y + N xi + c has stationary covariance S = G S G^T + N N^T; for the SOR-Gibbs
    sampler with noise sqrt((2-omega)/omega) D^{1/2} xi this must equal A^-1 (metric of src/stats.c)."""
    if grid == (5, 5, 5) and (om != 1.0 or colors == "single"):
        pytest.skip("3-D case kept to the headline configuration")
    A = O.shifted_laplace(*grid, 10.0)
    col = O.coloring_redblack(*grid) if colors == "redblack" else O.coloring_single(A.n)
    G, N = _maps(A, col, om, sweep)
    S = O.stationary_covariance(G, N)
    Q = np.linalg.inv(A.dense())
    assert np.linalg.norm(S - Q) / np.linalg.norm(Q) < 1e-10
    # and the stationary mean is A^-1 b
    b = np.linspace(1, 2, A.n)
    if sweep != O.SOR_SYMMETRIC:
        c = O.mcsor_apply(A, col, b, np.zeros(A.n), om, sweep)
    else:
        c = O.mcsor_apply(A, col, b, np.zeros(A.n), om, O.SOR_SYMMETRIC)
    mean = np.linalg.solve(np.eye(A.n) - G, c)
    assert np.allclose(mean, np.linalg.solve(A.dense(), b), rtol=1e-11)


def test_colorings_are_valid_and_deterministic():
    for g in [(9, 9, 1), (6, 5, 4), (2, 2, 2), (3, 1, 1)]:
        A = O.shifted_laplace(*g, 1.0)
        for col in (O.coloring_redblack(*g), O.coloring_greedy(A), O.coloring_lexlevels(A), O.coloring_iterated(A)):
            assert O.coloring_is_valid(A, col)
        assert O.coloring_redblack(*g).max() <= 1
        assert np.array_equal(O.coloring_greedy(A), O.coloring_redblack(*g))  # first-fit on a star stencil is red-black
        assert O.coloring_lexlevels(A).max() + 1 == sum(g) - 2  # hyperplanes i+j+k
    P = O.q1_interp(3, 3, 3)
    Ac = O.CSR.from_scipy(O.galerkin(O.shifted_laplace(5, 5, 5, 1.0).scipy(), P))
    assert O.coloring_is_valid(Ac, O.coloring_parity8(3, 3, 3)) and not O.coloring_is_valid(Ac, O.coloring_redblack(3, 3, 3))


def test_lexlevels_reproduces_serial_sweep_bitwise():
    """sweeping dependency levels == the reference's serial one-colour sweep (src/mc_sor.c:397-410)."""
    rng = np.random.default_rng(3)
    for g in [(9, 9, 1), (6, 5, 4)]:
        A = O.shifted_laplace(*g, 2.0)
        b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
        for om in (1.0, 1.2):
            for t in (O.SOR_FORWARD, O.SOR_BACKWARD, O.SOR_SYMMETRIC):
                assert np.array_equal(O.mcsor_apply(A, O.coloring_lexlevels(A), b, y, om, t), O.mcsor_apply(A, O.coloring_single(A.n), b, y, om, t))


def test_multidomain_equals_single_domain():
    rng = np.random.default_rng(4)
    A = O.shifted_laplace(6, 5, 4, 2.0)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    col = O.coloring_redblack(6, 5, 4)
    for t in (O.SOR_FORWARD, O.SOR_BACKWARD):
        one = O.mcsor_apply(A, col, b, y, 1.2, t)
        for own in ([0, 60, 120], [0, 30, 60, 90, 120], [0, 17, 120]):
            assert np.allclose(O.mcsor_sweep_domains(A, own, col, b, y, 1.2, t), one, rtol=0, atol=4e-16)


def test_q1_interpolation_and_galerkin():
    P = O.q1_interp(3, 3, 1).toarray()
    assert P.shape == (25, 9)
    assert np.array_equal(np.unique(P), [0, 0.25, 0.5, 1.0])
    assert np.allclose(P.sum(1), 1)  # partition of unity
    assert np.array_equal(P[12], [0.25 if c in (0, 1, 3, 4) else 0 for c in range(9)]) is False  # centre is a coarse node
    assert P[12, 4] == 1.0
    assert np.array_equal(np.nonzero(P[6])[0], [0, 1, 3, 4]) and np.all(P[6, [0, 1, 3, 4]] == 0.25)
    P3 = O.q1_interp(2, 2, 2).toarray()
    assert P3.shape == (27, 8) and np.all(P3[13] == 0.125)
    A = O.shifted_laplace(5, 5, 1, 10.0).scipy()
    Ac = O.galerkin(A, O.q1_interp(3, 3, 1)).toarray()
    assert np.allclose(Ac, Ac.T) and np.all(np.linalg.eigvalsh(Ac) > 0)
    assert np.count_nonzero(Ac[4]) == 9  # 9-point coarse stencil


def test_vcycle_sampler_has_exact_stationary_covariance():
    """MGMC with Galerkin coarse operators and an exact coarse sampler: the V-cycle chain in correction form
    (reference src/pc_gamgmc.c:242-259) is again a Gibbs-type chain with stationary covariance A^-1."""
    nxc = 3
    Pm = O.q1_interp(nxc, nxc, 1)
    Af = O.shifted_laplace(5, 5, 1, 2.0)
    Ac = O.CSR.from_scipy(O.galerkin(Af.scipy(), Pm))
    levels = [dict(A=Ac.scipy(), P=None), dict(A=Af.scipy(), P=Pm)]
    n, ncs = Af.n, Ac.n
    Lc = O.potrf_lower(Ac.dense())
    rb = O.coloring_redblack(5, 5, 1)
    sd = O.sqrtdiag(Af, 1.0, False)
    nnoise = 2 * n + ncs

    def chain(b, y, xi):  # one outer Richardson iteration with all noises injected
        def smooth(l, rhs, x, leg):
            w = O.prepare_rhs(xi[leg * n:(leg + 1) * n], sd, rhs)
            return O.mcsor_apply(Af, rb, w, x, 1.0, O.SOR_FORWARD)

        def coarse(rhs):
            return O.chol_sample(Lc, rhs, xi[2 * n:])

        return O.gamgmc_richardson(levels, b, y, 1, False, smooth, coarse)

    z = np.zeros(n)
    zx = np.zeros(nnoise)
    G = np.stack([chain(z, e, zx) for e in np.eye(n)], 1)
    N = np.stack([chain(z, z, e) for e in np.eye(nnoise)], 1)
    S = O.stationary_covariance(G, N)
    Q = np.linalg.inv(Af.dense())
    assert np.linalg.norm(S - Q) / np.linalg.norm(Q) < 1e-10
    assert np.max(np.abs(np.linalg.eigvals(G))) < 0.2  # and it mixes much faster than plain Gibbs


def test_golden_files_match_oracle():
    ops = np.load(GOLD / "operators.npz")
    A = O.shifted_laplace(9, 9, 1, 10.0)
    assert np.array_equal(ops["lap_9x9_k10_rowptr"], A.rowptr) and np.array_equal(ops["lap_9x9_k10_colidx"], A.colidx) and np.array_equal(ops["lap_9x9_k10_vals"], A.vals)
    assert np.array_equal(ops["lap_9x9_k10_redblack"], O.coloring_redblack(9, 9))
    sw = np.load(GOLD / "sweeps.npz")
    A = O.shifted_laplace(6, 5, 4, 2.0)
    got = O.mcsor_apply(A, O.coloring_redblack(6, 5, 4), sw["6x5x4_b"], sw["6x5x4_y"], 1.2, O.SOR_SYMMETRIC)
    assert np.array_equal(got, sw["6x5x4_redblack_om1.2_sym"])
    nz = np.load(GOLD / "noise.npz")
    assert np.array_equal(nz["grid_6x5x4_seed51966_sweep3"], O.noise_grid(6, 5, 4, 0xCAFE, 3))
    assert np.array_equal(nz["rows_seed51966_sweep7_n33"], O.noise_rows(33, 0xCAFE, 7))
    # normals are standard: mean ~ 0, var ~ 1 over a long row stream
    xi = O.noise_rows(200000, 12345, 0)
    assert abs(xi.mean()) < 0.01 and abs(xi.var() - 1) < 0.01
    from scipy import stats

    assert stats.kstest(xi, "norm").pvalue > 1e-3


def test_parallel_coloured_sample_equals_the_serial_restatement():
    """the all-cores CPU baseline of bench.py (one OpenMP loop per colour) is the same chain as the scalar oracle"""
    A = O.shifted_laplace(11, 7, 5, 2.0)
    n = A.n
    cols = O.coloring_redblack(11, 7, 5)
    nc, cptr, crows = O.color_lists(cols)
    dp, idg, sd = O.diag_pointers(A), O.idiag(A, 1.0), O.sqrtdiag(A, 1.0, True)
    rng = np.random.default_rng(0)
    b, y = rng.standard_normal(n), rng.standard_normal(n)
    want = O.gibbs_samples(A, cols, b, y, 1, lambda d: O.noise_rows(n, 9, 4), 1.0, O.SOR_FORWARD, True)
    for native in (False, True):
        got, w = y.copy(), np.zeros(n)
        O.lib(native).orc_gibbs_sample_colored_parallel(n, nc, cptr, crows, A.rowptr, A.colidx, A.vals, dp, idg, sd, 1.0, b, got, w, 9, 4)
        assert np.array_equal(got, want)
