"""Chain diagnostics of the reference (src/iact.c, src/stats.c): the library's host routines against
the oracle's numpy restatement.  No device work, so these run without a GPU."""
import numpy as np
import pytest

import oracle as O


def ar1(n, rho, seed):
    rng = np.random.default_rng(seed)
    x = np.empty(n)
    x[0] = rng.standard_normal()
    e = rng.standard_normal(n) * np.sqrt(1 - rho * rho)
    for i in range(1, n):
        x[i] = rho * x[i - 1] + e[i]
    return x


@pytest.mark.parametrize("n", [2, 7, 1000, 4096, 5001])
def test_autocorrelation_matches_oracle(n):
    from parmgmc_amd import autocorrelation

    x = ar1(n, 0.7, n)
    got, want = autocorrelation(x), O.autocorrelation(x)
    assert got[0] == 1.0
    assert np.abs(got - want).max() < 1e-12


@pytest.mark.parametrize("rho", [0.0, 0.5, 0.9])
def test_iact_matches_oracle_and_theory(rho):
    """AR(1) has IACT (1 + rho) / (1 - rho); examples/ex2.c:107-112 prints the estimate for a QOI series."""
    from parmgmc_amd import iact

    x = ar1(200000, rho, 3)
    tau, valid = iact(x)
    wtau, wvalid = O.iact(x)
    assert abs(tau - wtau) < 1e-9 * max(1, abs(wtau)) and valid == wvalid and valid
    assert abs(tau - (1 + rho) / (1 - rho)) < 0.1 * (1 + rho) / (1 - rho)


def test_iact_rejects_short_series():
    from parmgmc_amd import PMGError, iact

    with pytest.raises(PMGError) as e:
        iact(np.ones(1))
    assert e.value.code == 63 and "Too few data points" in str(e.value)  # src/iact.c:79


def test_covariance_errors_match_oracle_and_decay():
    """examples/ex6.c:184-199: independent chains, error of the sample covariance vs A^-1 per sample index."""
    from parmgmc_amd import estimate_covariance_errors

    A = O.ex6_matrix(4, 1.0)
    n, chains, spc = A.n, 400, 3
    Lc = np.linalg.cholesky(np.linalg.inv(A.dense()))
    rng = np.random.default_rng(0)
    S = (Lc @ rng.standard_normal((n, chains * spc))).T.copy()
    got = estimate_covariance_errors(A.rowptr, A.colidx, A.vals, S, chains)
    want = O.covariance_errors(A, S, chains)
    assert np.abs(got - want).max() < 1e-12
    assert np.all(got < 0.25)  # ~ sqrt(n / chains)
    # the metric of the oracle's single-chain helper (tests of the samplers use it) is the same formula
    assert abs(O.covariance_error(A, S[:chains]) - want[0]) < 1e-12


def test_make_observation_mats_on_the_grid():
    """MakeObservationMats (reference src/obs.c:135-180) restated for the DMDA: ball indicators x lumped mass / ball
    volume, S = 1/sigma2, f = B (S o y); slab rows are a slice of the full matrix."""
    from parmgmc_amd import make_observation_mats

    nx, ny, nz = 17, 13, 9
    coords = [0.25, 0.25, 0.5, 0.75, 0.6, 0.3]
    radii, vals, s2 = [0.2, 0.25], [1.0, -2.0], 1e-3
    B, S, f = make_observation_mats(nx, ny, nz, coords, radii, vals, s2)
    X, Y, Z = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), np.linspace(0, 1, nz), indexing="ij")
    pts = np.stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")], 1)
    h3 = 1.0 / ((nx - 1) * (ny - 1) * (nz - 1))
    for c in range(2):
        inside = ((pts - np.asarray(coords[3 * c:3 * c + 3])) ** 2).sum(1) < radii[c] ** 2
        want = np.where(inside, h3 / (4 * np.pi / 3 * radii[c] ** 3), 0.0)
        assert np.allclose(B[:, c], want, rtol=1e-14, atol=0)
        assert 0.5 < B[:, c].sum() < 1.3  # a ball average: the weights add up to about 1
    assert np.array_equal(S, np.full(2, 1.0 / s2))
    assert np.allclose(f, B @ (S * np.asarray(vals)), rtol=1e-14)
    Bs, _, fs = make_observation_mats(nx, ny, nz, coords, radii, vals, s2, kz0=3, nz_owned=4)
    assert np.array_equal(Bs, B[3 * nx * ny:7 * nx * ny]) and np.array_equal(fs, f[3 * nx * ny:7 * nx * ny])
    # 2-D: area of the disc
    B2, _, _ = make_observation_mats(33, 33, 1, [0.5, 0.5], [0.3], [1.0], 1.0)
    assert abs(B2.sum() - 1.0) < 0.05
