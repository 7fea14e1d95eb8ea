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
