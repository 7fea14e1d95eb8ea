"""Parity of the V-cycle pieces with the CPU oracle: Galerkin hierarchy, exact coarse
sampler, and the whole Multigrid Monte Carlo chain (PCApplyRichardson_GAMGMC) with identical noise."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O

pytestmark = pytest.mark.gpu
GOLD = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def dev(a):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a, np.float64), device="cuda")


def host(t):
    return t.detach().cpu().numpy()


def level_seed(seed, l):
    return (seed + GOLD * (l + 1)) & M64


def oracle_hierarchy(nx, ny, nz, kappa, levels):
    dims = [(nx, ny, nz)]
    for _ in range(levels - 1):
        dims.append(tuple((d - 1) // 2 + 1 if d > 1 else 1 for d in dims[-1]))
    dims = dims[::-1]  # dims[0] coarsest
    A = O.shifted_laplace(nx, ny, nz, kappa).scipy()
    lv = [None] * levels
    lv[levels - 1] = dict(A=A, P=None, dims=dims[-1])
    for l in range(levels - 1, 0, -1):
        P = O.q1_interp(*dims[l - 1])
        lv[l]["P"] = P
        lv[l - 1] = dict(A=O.galerkin(lv[l]["A"], P), P=None, dims=dims[l - 1])
    return lv


@pytest.mark.parametrize("grid,levels", [((9, 9, 1), 3), ((9, 5, 5), 2), ((17, 9, 9), 3)])
def test_hierarchy_matches_oracle(grid, levels):
    from parmgmc_amd import MGMC

    mg = MGMC(*grid, 2.0, levels, keep_host=True).setup()
    lv = oracle_hierarchy(*grid, 2.0, levels)
    for l in range(levels):
        assert mg.level_dims(l) == lv[l]["dims"]
    for l in range(1, levels):
        rp, ci, v = mg.level_matrix(l, "P")
        P = sp.csr_matrix(lv[l]["P"])
        P.sort_indices()
        assert np.array_equal(rp, P.indptr) and np.array_equal(ci, P.indices) and np.array_equal(v, P.data)  # index maps + weights bit-exact
    for l in range(levels - 1):
        rp, ci, v = mg.level_matrix(l, "A")
        Ac = sp.csr_matrix(lv[l]["A"])
        Ac.sort_indices()
        got = sp.csr_matrix((v, ci, rp), shape=Ac.shape)
        assert abs(got - Ac).max() < 1e-13 * abs(Ac).max()  # P^T A P: same entries, summation order differs (unpinned)
        assert got.nnz >= Ac.nnz


def test_chol_sampler_matches_oracle():
    from parmgmc_amd import CholSampler, PMGError

    A = O.CSR.from_scipy(O.galerkin(O.shifted_laplace(9, 9, 9, 1.0).scipy(), O.q1_interp(5, 5, 5)))
    ch = CholSampler(A.rowptr, A.colidx, A.vals)
    L = O.potrf_lower(A.dense())
    assert np.allclose(ch.factor(), L, rtol=1e-13, atol=1e-15)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(A.n)
    y = dev(np.zeros(A.n))
    ch.sample(dev(b), y, seed=5, counter=9)
    want = O.chol_sample(L, b, O.noise_rows(A.n, 5, 9))
    assert np.abs(host(y) - want).max() / np.abs(want).max() < 1e-12
    ch.sample(dev(b), y, noisy=False)
    assert np.allclose(host(y), np.linalg.solve(A.dense(), b), rtol=1e-11)
    # not SPD -> PETSC_ERR_MAT_CH_ZRPVT naming the failing minor (src/pc_chols.c:190)
    bad = O.CSR.from_scipy(sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 1.0]])))
    with pytest.raises(PMGError) as e:
        CholSampler(bad.rowptr, bad.colidx, bad.vals)
    assert e.value.code == 81 and "leading minor of order 2" in str(e.value)


def oracle_chain(grid, kappa, levels, b, y0, its, seed, counter0, guesszero, nu=1, scaled=False, omega=1.0, sweep=O.SOR_FORWARD, coarse="cholsampler", coarse_its=1):
    lv = oracle_hierarchy(*grid, kappa, levels)
    top = levels - 1
    csr = [O.CSR.from_scipy(x["A"]) for x in lv]
    cols = [O.coloring_parity8(*x["dims"]) for x in lv]
    cols[top] = O.coloring_redblack(*grid)
    Lc = O.potrf_lower(csr[0].dense()) if coarse == "cholsampler" else None
    y = np.array(y0, copy=True)
    out = []
    for it in range(its):
        s = counter0 + it
        ctr = {l: 64 * s for l in range(levels)}

        def noise(l):
            c = ctr[l]
            ctr[l] += 1
            if l == top:
                return O.noise_grid(*grid, level_seed(seed, l), c)
            return O.noise_rows(csr[l].n, level_seed(seed, l), c)

        def smooth(l, rhs, x, leg, its_=None):
            return O.gibbs_samples(csr[l], cols[l], rhs, x, nu if its_ is None else its_, lambda d: noise(l), omega, sweep, scaled)

        def coarse_fn(rhs):
            if coarse == "cholsampler":
                return O.chol_sample(Lc, rhs, noise(0))
            return smooth(0, rhs, np.zeros(csr[0].n), 0, coarse_its)

        y = O.gamgmc_richardson(lv, b, y, 1, guesszero and it == 0, smooth, coarse_fn)
        out.append(y.copy())
    return out


@pytest.mark.parametrize("literal", [False, True], ids=["in_place", "correction_form"])
@pytest.mark.parametrize("grid,levels", [((9, 9, 1), 3), ((9, 5, 5), 2), ((17, 9, 9), 3)])
def test_mgmc_chain_matches_oracle(grid, levels, literal):
    """The whole sampler against the oracle's restatement of src/pc_gamgmc.c:227-264 + PCMG, same noise streams:
    tolerance 1e-11 relative (Galerkin entries / residual sums are computed in a different order; noise 1e-13)."""
    from parmgmc_amd import MGMC

    kappa = 2.0
    rng = np.random.default_rng(3)
    n = int(np.prod(grid))
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    mg = MGMC(*grid, kappa, levels)
    mg.set_correction_form(literal)  # both forms are the chain of src/pc_gamgmc.c:242-259 (identical up to rounding)
    mg.setup()
    seen = []
    yd = dev(y0)
    nxt = mg.sample(dev(b), yd, 3, seed=0xCAFE, counter0=2, guesszero=False, callback=lambda it, y: seen.append(host(y).copy()))
    assert nxt == 5
    want = oracle_chain(grid, kappa, levels, b, y0, 3, 0xCAFE, 2, False)
    for got, w in zip(seen + [host(yd)], want + [want[-1]]):
        assert np.abs(got - w).max() / np.abs(w).max() < 1e-11
    # zero initial guess: first sample is MG(b)
    yd = dev(np.zeros(n))
    mg.sample(dev(b), yd, 2, seed=7, counter0=0, guesszero=True)
    w = oracle_chain(grid, kappa, levels, b, np.zeros(n), 2, 7, 0, True)[-1]
    assert np.abs(host(yd) - w).max() / np.abs(w).max() < 1e-11


def test_mgmc_options_mcgibbs_levels_and_gibbs_coarse():
    """reference examples/ex1.c:41: mcgibbs on the levels (2 sweeps) and on the coarse grid (2 sweeps)."""
    from parmgmc_amd import MGMC

    grid, kappa, levels = (9, 9, 1), 10.0, 3
    rng = np.random.default_rng(4)
    b, y0 = rng.standard_normal(81), rng.standard_normal(81)
    mg = MGMC(*grid, kappa, levels)
    mg.set_smoother(True, 1.2, O.SOR_SYMMETRIC, 2)
    mg.set_coarse("gibbs", 2)
    mg.setup()
    yd = dev(y0)
    mg.sample(dev(b), yd, 2, seed=11, counter0=0)
    w = oracle_chain(grid, kappa, levels, b, y0, 2, 11, 0, False, nu=2, scaled=True, omega=1.2, sweep=O.SOR_SYMMETRIC, coarse="gibbs", coarse_its=2)[-1]
    assert np.abs(host(yd) - w).max() / np.abs(w).max() < 1e-11


def test_mgmc_rejects_uncoarsenable_grid():
    from parmgmc_amd import MGMC, PMGError

    with pytest.raises(PMGError) as e:
        MGMC(8, 8, 8, 1.0, 2)
    assert e.value.code == 60


def test_ex1_geometric_mgmc_mean_on_device():
    """reference examples/ex1.c:44 (geometric MGMC, 3x3 refined twice = 9x9, 3 levels, coarse cholsampler, kappa 10,
    b = 1): sample mean -> A^-1 b.  MGMC samples are nearly independent, so 2e4 samples / 200 burn-in with the
    bound scaled as 1/sqrt(n): 0.02*sqrt(1e6/2e4) = 0.14."""
    import torch

    from parmgmc_amd import MGMC

    mg = MGMC(9, 9, 1, 10.0, 3)
    mg.set_smoother(True, 1.0, O.SOR_FORWARD, 2)
    mg.setup()
    b, y = dev(np.ones(81)), dev(np.zeros(81))
    ctr = mg.sample(b, y, 200, seed=0xCAFE)
    mean = torch.zeros_like(y)

    def cb(it, yy):
        mean.mul_(it / (it + 1.0)).add_(yy, alpha=1.0 / (it + 1))

    mg.sample(b, y, 20000, seed=0xCAFE, counter0=ctr, callback=cb)
    ex = np.linalg.solve(O.shifted_laplace(9, 9, 1, 10.0).dense(), np.ones(81))
    assert np.linalg.norm(host(mean) - ex) / np.linalg.norm(ex) < 0.14


@pytest.mark.parametrize("n3", [4, 9, 13])
def test_device_cholesky_mfma_matches_oracle(n3):
    """The device potrf (f64 MFMA trailing updates) + blocked inverse against the oracle's scalar potrf, sizes that
    are / are not multiples of the 32-wide tile and need several panel steps (13^3 = 2197 rows = 69 tiles)."""
    from parmgmc_amd import CholSampler

    A = O.shifted_laplace(n3, n3, n3, 1.0)
    ch = CholSampler(A.rowptr, A.colidx, A.vals)
    L = np.linalg.cholesky(A.dense())
    got = ch.factor()
    assert np.abs(got - L).max() / np.abs(L).max() < 1e-12
    rng = np.random.default_rng(n3)
    b = rng.standard_normal(A.n)
    y = dev(np.zeros(A.n))
    ch.sample(dev(b), y, noisy=False)
    x = np.linalg.solve(A.dense(), b)
    assert np.abs(host(y) - x).max() / np.abs(x).max() < 1e-11
    ch.sample(dev(b), y, seed=3, counter=4)
    xi = O.noise_rows(A.n, 3, 4)
    want = np.linalg.solve(L.T, np.linalg.solve(L, b) + xi)
    assert np.abs(host(y) - want).max() / np.abs(want).max() < 1e-11


def test_unstructured_hierarchy_lshape():
    """BASELINE config 4: MGMC on the P1 matrix of the reference's data/lshape.msh with an algebraic (aggregation)
    hierarchy handed over level by level, multicolour Gibbs on every level (AIJ path), exact coarse sampler --
    against the oracle's restatement of src/pc_gamgmc.c:227-264 with the same noise streams."""
    from pathlib import Path

    from fem_p1 import assemble_p1, greedy_aggregation, read_gmsh41_triangles
    from parmgmc_amd import MGMC

    xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent / "golden" / "lshape.msh")
    assert xy.shape == (408, 2) and len(tris) == 734
    A2 = assemble_p1(xy, tris, kappa=1.0)
    P2 = greedy_aggregation(A2)
    A1 = O.galerkin(A2, P2)
    P1 = greedy_aggregation(A1)
    A0 = O.galerkin(A1, P1)
    ops = [O.CSR.from_scipy(m) for m in (A0, A1, A2)]
    assert ops[0].n < ops[1].n < 408
    mg = MGMC.from_hierarchy([(m.rowptr, m.colidx, m.vals) for m in ops], [None, (P1.indptr, P1.indices, P1.data), (P2.indptr, P2.indices, P2.data)])
    mg.set_smoother(True, 1.0, O.SOR_SYMMETRIC, 1)
    mg.setup()
    rng = np.random.default_rng(8)
    b, y0 = rng.standard_normal(408), rng.standard_normal(408)
    yd = dev(y0)
    mg.sample(dev(b), yd, 3, seed=21, counter0=0)
    # oracle chain with the library's stated rules: greedy colouring on every level, row-stream noise, 64 draws/sample
    lv = [dict(A=A0, P=None), dict(A=A1, P=P1), dict(A=A2, P=P2)]
    cols = [O.coloring_greedy(m) for m in ops]
    Lc = O.potrf_lower(ops[0].dense())
    y = y0.copy()
    for s in range(3):
        ctr = {l: 64 * s for l in range(3)}

        def noise(l):
            c = ctr[l]
            ctr[l] += 1
            return O.noise_rows(ops[l].n, level_seed(21, l), c)

        smooth = lambda l, rhs, x, leg: O.gibbs_samples(ops[l], cols[l], rhs, x, 1, lambda d: noise(l), 1.0, O.SOR_SYMMETRIC, True)
        y = O.gamgmc_richardson(lv, b, y, 1, False, smooth, lambda rhs: O.chol_sample(Lc, rhs, noise(0)))
    assert np.abs(host(yd) - y).max() / np.abs(y).max() < 1e-11
    # and the sampler targets N(A^-1 b, A^-1): sample mean of 4000 (nearly independent) MGMC samples
    import torch

    mean = torch.zeros_like(yd)
    mg.sample(dev(b), yd, 4000, seed=21, counter0=3, callback=lambda it, yy: mean.mul_(it / (it + 1.0)).add_(yy, alpha=1.0 / (it + 1)))
    ex = np.linalg.solve(A2.toarray(), b)
    Cdiag = np.diag(np.linalg.inv(A2.toarray()))
    z = (host(mean) - ex) / np.sqrt(Cdiag / 4000)
    # standardised errors: mean square = integrated autocorrelation time of the chain (about 4 with this plain,
    # unsmoothed aggregation; a single-level Gibbs chain on this matrix has IACT in the hundreds)
    assert np.abs(z).max() < 6.0 * np.sqrt(np.mean(z ** 2)) and 0.3 < np.mean(z ** 2) < 10.0


def test_unstructured_hierarchy_with_the_iterated_colouring():
    """round 4: PMG_COLORING_ITERATED on the AIJ levels (pmg_mgmc_set_coloring) -- first-fit, then once more with the classes
    visited last class first.  On the P1 matrix of lshape.msh refined twice it saves a class (and a launch per sweep) against
    first-fit; the colourings are bit-equal to the oracle's twin, the chain agrees with the oracle chain on those colourings."""
    from pathlib import Path

    from fem_p1 import assemble_p1, greedy_aggregation, read_gmsh41_triangles
    from parmgmc_amd import COLORING_ITERATED, MCSOR, MGMC
    from parmgmc_amd.unstructured import refine_uniform

    xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent / "golden" / "lshape.msh")
    for _ in range(2):
        xy, tris = refine_uniform(xy, tris)
    A2 = assemble_p1(xy, tris, kappa=1.0)
    P2 = greedy_aggregation(A2)
    A1 = O.galerkin(A2, P2)
    P1 = greedy_aggregation(A1)
    A0 = O.galerkin(A1, P1)
    ops = [O.CSR.from_scipy(m) for m in (A0, A1, A2)]
    n = ops[2].n
    cols = [O.coloring_iterated(m) for m in ops]
    assert all(O.coloring_is_valid(m, c) for m, c in zip(ops, cols))
    assert cols[2].max() < O.coloring_greedy(ops[2]).max()  # one class fewer on the fine P1 matrix
    mc = MCSOR(ops[2].rowptr, ops[2].colidx, ops[2].vals, COLORING_ITERATED).setup()
    assert np.array_equal(mc.get_coloring(), cols[2]) and mc.get_num_colors() == cols[2].max() + 1
    mg = MGMC.from_hierarchy([(m.rowptr, m.colidx, m.vals) for m in ops], [None, (P1.indptr, P1.indices, P1.data), (P2.indptr, P2.indices, P2.data)])
    mg.set_coloring(COLORING_ITERATED)
    mg.set_smoother(True, 1.0, O.SOR_FORWARD, 1)
    mg.setup()
    rng = np.random.default_rng(9)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    yd = dev(y0)
    mg.sample(dev(b), yd, 3, seed=22, counter0=0)
    lv = [dict(A=A0, P=None), dict(A=A1, P=P1), dict(A=A2, P=P2)]
    Lc = O.potrf_lower(ops[0].dense())
    y = y0.copy()
    for s in range(3):
        ctr = {l: 64 * s for l in range(3)}

        def noise(l):
            c = ctr[l]
            ctr[l] += 1
            return O.noise_rows(ops[l].n, level_seed(22, l), c)

        smooth = lambda l, rhs, x, leg: O.gibbs_samples(ops[l], cols[l], rhs, x, 1, lambda d: noise(l), 1.0, O.SOR_FORWARD, True)
        y = O.gamgmc_richardson(lv, b, y, 1, False, smooth, lambda rhs: O.chol_sample(Lc, rhs, noise(0)))
    assert np.abs(host(yd) - y).max() / np.abs(y).max() < 1e-11


@pytest.mark.parametrize("grid,levels,coarse", [((33, 17, 17), 3, "cholsampler"), ((65, 33, 1), 4, "gibbs"), ((17, 17, 17), 4, "cholsampler")])
def test_proxy_stencil_setup_is_bit_identical_to_full_galerkin(grid, levels, coarse, monkeypatch):
    """The default set-up takes the coarse class-stencil tables from a small proxy hierarchy (2^levels + 1 points per
    direction) instead of forming P^T A P of the full-size matrices; the chain must not change by a bit."""
    from parmgmc_amd import MGMC

    rng = np.random.default_rng(5)
    n = int(np.prod(grid))
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    out = []
    for full in (False, True):
        if full:
            monkeypatch.setenv("PMG_MG_FULL_GALERKIN", "1")
        mg = MGMC(*grid, 1.5, levels)
        mg.set_smoother(True, 1.1, O.SOR_SYMMETRIC, 1)
        mg.set_coarse(coarse, 2)
        mg.setup()
        yd = dev(y0)
        mg.sample(dev(b), yd, 3, seed=17, counter0=1)
        out.append(host(yd).copy())
    assert np.array_equal(out[0], out[1])
