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


# ------------------------------------------------------------------------------------------------------------
# MGMC on a MATLRC operator: every level gets A_l + B_l S B_l^T with B_{l-1} = P_l^T B_l (src/pc_gamgmc.c:157-196)
# ------------------------------------------------------------------------------------------------------------
GOLD = 0x9E3779B97F4A7C15


def level_seed(seed, l):
    return (seed + GOLD * (l + 1)) & M64


def mg_hierarchy(grid, kappa, levels):
    dims = [grid]
    for _ in range(levels - 1):
        dims.append(tuple((d - 1) // 2 + 1 if d > 1 else 1 for d in dims[-1]))
    dims = dims[::-1]
    lv = [None] * levels
    lv[levels - 1] = dict(A=O.shifted_laplace(*grid, kappa).scipy(), P=None, dims=dims[-1])
    for l in range(levels - 1, 0, -1):
        lv[l]["P"] = O.q1_interp(*dims[l - 1])
        lv[l - 1] = dict(A=O.galerkin(lv[l]["A"], lv[l]["P"]), P=None, dims=dims[l - 1])
    return lv


class LrcMgmcOracle:
    """The oracle's restatement of PCGAMGMC on a MATLRC operator: the hierarchy of the base matrix, per-level
    factors B_l, level samplers = LRC Gibbs sweeps (src/mc_sor.c:101-112, src/pc_mcgibbs.c:130-140), level residuals
    with the LRC operator (src/pc_gamgmc.c:186-194), coarse = Cholesky of the explicit sum (src/pc_chols.c:119-153)."""

    def __init__(self, lv, colors, B, S, nu=1, omega=1.0, sweep=O.SOR_FORWARD, scaled=True, coarse="cholsampler", coarse_its=1):
        self.base, self.colors, self.S = lv, colors, np.asarray(S, float)
        self.nu, self.omega, self.sweep, self.scaled, self.coarse, self.coarse_its = nu, omega, sweep, scaled, coarse, coarse_its
        self.Bl = O.lrc_level_factors(lv, B)
        self.csr = [O.CSR.from_scipy(x["A"]) for x in lv]
        self.lv = [dict(A=O.LRCOperator(x["A"], self.Bl[l], self.S), P=x["P"]) for l, x in enumerate(lv)]
        dirs = [O.SOR_FORWARD, O.SOR_BACKWARD]
        self.Bb = [{d: O.lrc_build_correction(self.csr[l], colors[l], self.Bl[l], self.S, omega, d) for d in dirs} for l in range(len(lv))]
        self.sd = [O.sqrtdiag(self.csr[l], omega, scaled) for l in range(len(lv))]
        self.Lc = O.potrf_lower(self.lv[0]["A"].dense()) if coarse == "cholsampler" else None
        self.ndir = 2 if sweep == O.SOR_SYMMETRIC else 1

    def draws_per_smooth(self, l):
        return (self.coarse_its if (l == 0 and self.coarse == "gibbs") else self.nu) * self.ndir

    def sweeps(self, l, rhs, x, its, xi_fn, eta_fn):
        """its samples of the level sampler; xi_fn(d) / eta_fn(d) = the d-th directional sweep's draws"""
        d = 0
        sq = np.sqrt(np.abs(self.S))
        for _ in range(its):
            for direction in ([O.SOR_FORWARD, O.SOR_BACKWARD] if self.sweep == O.SOR_SYMMETRIC else [self.sweep]):
                w = O.prepare_rhs(xi_fn(d), self.sd[l], rhs) + self.Bl[l] @ (sq * eta_fn(d))
                d += 1
                x = O.lrc_mcsor_apply(self.csr[l], self.colors[l], self.Bl[l], self.Bb[l][O.SOR_FORWARD], self.Bb[l][O.SOR_BACKWARD], w, x, self.omega, direction)
        return x

    def chain(self, b, y, its, guesszero, xi, eta, chol_xi):
        """xi(l, c) / eta(l, c): the c-th draw of level l in this chain call (c counts from 0 per SAMPLE via the
        caller's closures); chol_xi(c) the coarse Cholesky draw."""
        top = len(self.lv) - 1
        for it in range(its):
            ctr = {l: 0 for l in range(top + 1)}

            def smooth(l, rhs, x, leg, n=None):
                n = self.nu if n is None else n
                c0 = ctr[l]
                ctr[l] += n * self.ndir
                return self.sweeps(l, rhs, x, n, lambda d: xi(it, l, c0 + d), lambda d: eta(it, l, c0 + d))

            def coarse_fn(rhs):
                if self.coarse == "cholsampler":
                    return O.chol_sample(self.Lc, rhs, chol_xi(it))
                return smooth(0, rhs, np.zeros(len(rhs)), 0, self.coarse_its)

            y = O.gamgmc_richardson(self.lv, b, y, 1, guesszero and it == 0, smooth, coarse_fn)
        return y


@pytest.mark.parametrize("coarse", ["cholsampler", "gibbs"])
def test_oracle_mgmc_lrc_samples_the_posterior(coarse):
    """Lyapunov check of the V-cycle restatement: with per-level operators A_l + B_l S B_l^T (src/pc_gamgmc.c:157-196)
    the MGMC chain's stationary law is N(A_post^-1 b, A_post^-1) exactly."""
    grid, kappa, levels, k = (5, 5, 1), 3.0, 2, 2
    lv = mg_hierarchy(grid, kappa, levels)
    n = 25
    B = observation_matrix(n, k, 4)
    S = np.array([30.0, 70.0])
    colors = [O.coloring_parity8(*lv[0]["dims"]), O.coloring_redblack(*grid)]
    orc = LrcMgmcOracle(lv, colors, B, S, nu=1, omega=1.0, sweep=O.SOR_SYMMETRIC, scaled=True, coarse=coarse, coarse_its=2)
    sizes = [x["A"].shape[0] for x in lv]
    # noise layout per sample: level l gets D_l draws of (n_l + k), the coarse Cholesky n_0
    D = [orc.draws_per_smooth(0) if coarse == "gibbs" else 0, 2 * orc.draws_per_smooth(1)]
    off, total = {}, 0
    for l in range(levels):
        for c in range(D[l]):
            off[("xi", l, c)] = total
            total += sizes[l]
            off[("eta", l, c)] = total
            total += k
    off["chol"] = total
    total += sizes[0] if coarse == "cholsampler" else 0

    def run(b, y, z):
        return orc.chain(b, y, 1, False, lambda it, l, c: z[off[("xi", l, c)]:off[("xi", l, c)] + sizes[l]], lambda it, l, c: z[off[("eta", l, c)]:off[("eta", l, c)] + k],
                         lambda it: z[off["chol"]:off["chol"] + sizes[0]])

    zb, zz = np.zeros(n), np.zeros(total)
    G = np.stack([run(zb, e, zz) for e in np.eye(n)], 1)
    N = np.stack([run(zb, zb, e) for e in np.eye(total)], 1)
    Sig = O.stationary_covariance(G, N)
    Apost = lv[-1]["A"].toarray() + B @ np.diag(S) @ B.T
    Q = np.linalg.inv(Apost)
    assert np.linalg.norm(Sig - Q) / np.linalg.norm(Q) < 1e-10
    b = np.linspace(1, 2, n)
    mean = np.linalg.solve(np.eye(n) - G, run(b, zb, zz))
    assert np.allclose(mean, np.linalg.solve(Apost, b), rtol=1e-10)


def _device_mgmc_lrc_case(grid, kappa, levels, k, nu, omega, sweep, scaled, coarse, coarse_its, literal, colors_fn=None):
    from parmgmc_amd import MGMC

    lv = mg_hierarchy(grid, kappa, levels)
    n = int(np.prod(grid))
    B = observation_matrix(n, k, 6)
    S = np.linspace(20.0, 90.0, k)
    rng = np.random.default_rng(12)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    mg = MGMC(*grid, kappa, levels)
    mg.set_smoother(scaled, omega, sweep, nu)
    mg.set_coarse(coarse, coarse_its)
    mg.set_correction_form(literal)
    mg.set_lowrank(B, S)
    mg.setup()
    yd = dev(y0)
    seed, c0, its = 0xBEEF, 3, 2
    mg.sample(dev(b), yd, its, seed=seed, counter0=c0, guesszero=False)
    top = levels - 1
    colors = [O.coloring_parity8(*x["dims"]) for x in lv]
    colors[top] = O.coloring_redblack(*grid)
    orc = LrcMgmcOracle(lv, colors, B, S, nu, omega, sweep, scaled, coarse, coarse_its)
    sizes = [x["A"].shape[0] for x in lv]

    def xi(it, l, c):
        ctr = 64 * (c0 + it) + c
        return O.noise_grid(*grid, level_seed(seed, l), ctr) if l == top else O.noise_rows(sizes[l], level_seed(seed, l), ctr)

    eta = lambda it, l, c: O.noise_rows(k, (level_seed(seed, l) + ETA_TAG) & M64, 64 * (c0 + it) + c)
    chol_xi = lambda it: O.noise_rows(sizes[0], level_seed(seed, 0), 64 * (c0 + it))
    want = orc.chain(b, y0, its, False, xi, eta, chol_xi)
    return np.abs(host(yd) - want).max() / np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("literal", [False, True], ids=["in_place", "correction_form"])
@pytest.mark.parametrize("grid,levels", [((9, 9, 1), 3), ((9, 5, 5), 2), ((17, 9, 9), 3)])
def test_device_mgmc_lrc_matches_oracle(grid, levels, literal):
    """BASELINE config 5 in small: the whole MGMC chain on A + B S B^T (grid level, class-stencil levels, coarse
    Cholesky of the explicit sum) against the oracle with the same noise streams; tolerance as the plain chain."""
    assert _device_mgmc_lrc_case(grid, 2.0, levels, 4, 1, 1.0, O.SOR_FORWARD, False, "cholsampler", 1, literal) < 1e-10


@pytest.mark.gpu
def test_device_mgmc_lrc_mcgibbs_levels_gibbs_coarse():
    assert _device_mgmc_lrc_case((9, 9, 1), 10.0, 3, 3, 2, 1.2, O.SOR_SYMMETRIC, True, "gibbs", 2, False) < 1e-10


@pytest.mark.gpu
def test_device_mgmc_lrc_sliced_ell_levels(monkeypatch):
    """the same chain with the class-stencil levels switched off: coarse levels run the sliced-ELL sampler"""
    monkeypatch.setenv("PMG_MG_NO_STENCIL", "1")
    assert _device_mgmc_lrc_case((9, 9, 5), 2.0, 3, 3, 1, 1.0, O.SOR_SYMMETRIC, True, "gibbs", 1, False) < 1e-10


@pytest.mark.gpu
def test_device_chol_sampler_lowrank():
    from parmgmc_amd import CholSampler

    A = O.shifted_laplace(7, 6, 1, 2.0)
    B = observation_matrix(A.n, 3, 9)
    S = np.array([10.0, 30.0, 20.0])
    ch = CholSampler(A.rowptr, A.colidx, A.vals, B, S)
    Apost = A.dense() + B @ np.diag(S) @ B.T
    L = O.potrf_lower(Apost)
    assert np.allclose(ch.factor(), L, rtol=1e-13, atol=1e-15)
    rng = np.random.default_rng(2)
    b = rng.standard_normal(A.n)
    y = dev(np.zeros(A.n))
    ch.sample(dev(b), y, seed=5, counter=1)
    want = O.chol_sample(L, b, O.noise_rows(A.n, 5, 1))
    assert np.abs(host(y) - want).max() / np.abs(want).max() < 1e-12


@pytest.mark.gpu
def test_ex4_mgmc_posterior_mean_on_device():
    """reference examples/ex4.c (MGMC with a low-rank update, posterior mean vs a direct solve): 17x17 grid, 3 levels,
    5 observations, 2e4 samples; bound scaled as the ex1 test (0.02*sqrt(1e6/2e4))."""
    import torch

    from parmgmc_amd import MGMC

    grid, kappa, k = (17, 17, 1), 10.0, 5
    n = 289
    A = O.shifted_laplace(*grid, kappa)
    B = observation_matrix(n, k, 11)
    S = np.full(k, 1e3)
    mg = MGMC(*grid, kappa, 3)
    mg.set_smoother(True, 1.0, O.SOR_SYMMETRIC, 1)
    mg.set_lowrank(B, S)
    mg.setup()
    b = np.ones(n) + B @ (S * np.linspace(-0.5, 0.5, k))
    bd, y = dev(b), dev(np.zeros(n))
    ctr = mg.sample(bd, y, 100, seed=5)
    mean = torch.zeros_like(y)
    mg.sample(bd, y, 20000, seed=5, counter0=ctr, callback=lambda it, yy: mean.mul_(it / (it + 1.0)).add_(yy, alpha=1.0 / (it + 1)))
    ex = np.linalg.solve(A.dense() + B @ np.diag(S) @ B.T, b)
    assert np.linalg.norm(host(mean) - ex) / np.linalg.norm(ex) < 0.14


def ball_matrix(grid, centres, radii):
    """ball-indicator observation vectors on the unit cube grid (reference src/obs.c:39-50): supported on << N rows"""
    nx, ny, nz = grid
    X, Y, Z = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), np.linspace(0, 1, nz) if nz > 1 else np.zeros(1), indexing="ij")
    pts = np.stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")], 1)  # natural order, x fastest
    B = np.zeros((nx * ny * nz, len(radii)))
    for c, (ctr, r) in enumerate(zip(centres, radii)):
        inside = ((pts - np.asarray(ctr)) ** 2).sum(1) < r * r
        B[inside, c] = 1.0 / max(1, inside.sum())
    return B


@pytest.mark.gpu
def test_device_mgmc_lrc_row_compact_form(monkeypatch):
    """ball observations are stored for their support rows only; the chain must agree with the dense
    storage to rounding, sample by sample, and with the oracle"""
    import torch

    from parmgmc_amd import MGMC

    grid, kappa, levels = (33, 33, 17), 2.0, 3
    n = int(np.prod(grid))
    B = ball_matrix(grid, [(0.3, 0.3, 0.4), (0.7, 0.6, 0.5), (0.5, 0.2, 0.8)], [0.12, 0.15, 0.1])
    assert 0 < (np.abs(B).sum(1) > 0).mean() < 0.05
    S = np.array([50.0, 80.0, 30.0])
    rng = np.random.default_rng(2)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    out = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("PMG_LRC_DENSE", "1")
        mg = MGMC(*grid, kappa, levels)
        mg.set_smoother(True, 1.1, O.SOR_SYMMETRIC, 1)
        mg.set_lowrank(B, S)
        mg.setup()
        bd, yd = dev(b), dev(y0)
        mg.sample(bd, yd, 3, seed=9, counter0=0)
        assert np.array_equal(host(bd), b)  # the right-hand side is restored bit for bit
        out.append(host(yd).copy())
    assert np.abs(out[0] - out[1]).max() / np.abs(out[1]).max() < 1e-12
