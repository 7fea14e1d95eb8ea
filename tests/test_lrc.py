"""Low-rank (MATLRC) updates, SURVEY 8 rows a11 / f-3: the oracle's restatement has the right stationary law
(CPU, deterministic), and the HIP path matches it (GPU)."""
import numpy as np
import pytest

import oracle as O

ETA_TAG = 0x632BE59BD9B4E019
M64 = (1 << 64) - 1


def observation_matrix(n, k, seed):
    """k localised "ball average" observation vectors (the role of MakeObservationMats, reference src/obs.c:135-180)."""
    rng = np.random.default_rng(seed)
    B = np.zeros((n, k))
    for c in range(k):
        idx = rng.choice(n, size=max(3, n // 12), replace=False)
        B[idx, c] = rng.random(len(idx))
        B[:, c] /= B[:, c].sum()
    return B


@pytest.mark.parametrize("sweep", [O.SOR_FORWARD, O.SOR_BACKWARD, O.SOR_SYMMETRIC])
@pytest.mark.parametrize("omega", [1.0, 1.3])
def test_oracle_lrc_chain_samples_the_posterior(sweep, omega):
    """Lyapunov check: the chain of src/pc_mcgibbs.c:130-188 + src/mc_sor.c:101-112 on A + B S B^T has stationary
    covariance (A + B S B^T)^-1 and mean (A + B S B^T)^-1 b -- exactly (metric of src/stats.c)."""
    A = O.shifted_laplace(6, 6, 1, 3.0)
    n, k = A.n, 3
    B = observation_matrix(n, k, 1)
    S = np.array([50.0, 20.0, 80.0])
    col = O.coloring_redblack(6, 6)
    ndraw = 2 if sweep == O.SOR_SYMMETRIC else 1

    def chain(b, y, xi, eta):
        return O.lrc_gibbs_samples(A, col, B, S, b, y, 1, lambda d: xi[d * n:(d + 1) * n], lambda d: eta[d * k:(d + 1) * k], omega, sweep, True)

    z, zx, ze = np.zeros(n), np.zeros(ndraw * n), np.zeros(ndraw * k)
    G = np.stack([chain(z, e, zx, ze) for e in np.eye(n)], 1)
    N = np.concatenate([np.stack([chain(z, z, e, ze) for e in np.eye(ndraw * n)], 1), np.stack([chain(z, z, zx, e) for e in np.eye(ndraw * k)], 1)], 1)
    Sig = O.stationary_covariance(G, N)
    Apost = A.dense() + B @ np.diag(S) @ B.T
    Q = np.linalg.inv(Apost)
    assert np.linalg.norm(Sig - Q) / np.linalg.norm(Q) < 1e-10
    b = np.linspace(1, 2, n)
    mean = np.linalg.solve(np.eye(n) - G, chain(b, z, zx, ze))
    assert np.allclose(mean, np.linalg.solve(Apost, b), rtol=1e-10)


def dev(a):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a, np.float64), device="cuda")


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["grid", "csr"])
def test_device_lrc_matches_oracle(path):
    from parmgmc_amd import MCSOR, GridMCSOR

    nx, ny, nz, kappa, k = 10, 7, 4, 2.0, 5
    A = O.shifted_laplace(nx, ny, nz, kappa)
    n = A.n
    B = observation_matrix(n, k, 2)
    S = np.array([40.0, 10.0, 25.0, 60.0, 5.0])
    rng = np.random.default_rng(3)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    for omega, sweep, scaled in [(1.0, O.SOR_FORWARD, True), (1.2, O.SOR_SYMMETRIC, True), (1.0, O.SOR_FORWARD, False)]:
        if path == "grid":
            s = GridMCSOR(nx, ny, nz, kappa)
            col = O.coloring_redblack(nx, ny, nz)
            noise = lambda d: O.noise_grid(nx, ny, nz, 9, 4 + d)
        else:
            s = MCSOR(A.rowptr, A.colidx, A.vals).setup()
            col = s.get_coloring()
            noise = lambda d: O.noise_rows(n, 9, 4 + d)
        s.set_omega(omega)
        s.set_sweep_type(sweep)
        s.set_lowrank(B, S)
        # deterministic MCSORApply with the Woodbury repair (src/mc_sor.c:216-239 + :101-112)
        yd = dev(y0)
        s.apply(dev(b), yd)
        Bb_f = O.lrc_build_correction(A, col, B, S, omega, O.SOR_FORWARD)
        Bb_b = O.lrc_build_correction(A, col, B, S, omega, O.SOR_BACKWARD)
        want = O.lrc_mcsor_apply(A, col, B, Bb_f, Bb_b, b, y0, omega, sweep)
        assert np.abs(host(yd) - want).max() / np.abs(want).max() < 1e-12
        # noisy chain
        yd = dev(y0)
        s.sample(dev(b), yd, 3, seed=9, counter0=4, scaled=scaled)
        eta = lambda d: O.noise_rows(k, (9 + ETA_TAG) & M64, 4 + d)
        want = O.lrc_gibbs_samples(A, col, B, S, b, y0, 3, noise, eta, omega, sweep, scaled)
        assert np.abs(host(yd) - want).max() / np.abs(want).max() < 1e-11
        # removing the update restores the plain sampler
        s.set_lowrank(np.zeros((n, 0)), np.zeros(0))
        yd = dev(y0)
        s.sample(dev(b), yd, 2, seed=9, counter0=4, scaled=scaled)
        want = O.gibbs_samples(A, col, b, y0, 2, noise, omega, sweep, scaled)
        assert np.abs(host(yd) - want).max() / np.abs(want).max() < 1e-13


@pytest.mark.gpu
def test_device_lrc_posterior_mean():
    """reference examples/ex4.c: posterior mean with a low-rank update, rel. error <= tol after burn-in (here the
    stand-alone mcgibbs line, 1e4 samples; the LRC chain mixes like plain Gibbs, bound 0.1)."""
    import torch

    from parmgmc_amd import GridMCSOR

    nx, ny, kappa, k = 9, 9, 10.0, 3
    A = O.shifted_laplace(nx, ny, 1, kappa)
    B = observation_matrix(81, k, 5)
    S = np.array([2e3, 1e3, 3e3])
    g = GridMCSOR(nx, ny, 1, kappa)
    g.set_lowrank(B, S)
    b = np.ones(81) + B @ (S * np.array([0.5, -0.2, 0.1]))  # f = prior rhs + B S y_obs (src/obs.c:176-178)
    bd, y = dev(b), dev(np.zeros(81))
    ctr = g.sample(bd, y, 500, seed=77)
    mean = torch.zeros_like(y)
    for it in range(20000):
        ctr = g.sample(bd, y, 1, seed=77, counter0=ctr)
        mean.mul_(it / (it + 1.0)).add_(y, alpha=1.0 / (it + 1))
    ex = np.linalg.solve(A.dense() + B @ np.diag(S) @ B.T, b)
    assert np.linalg.norm(host(mean) - ex) / np.linalg.norm(ex) < 0.1
