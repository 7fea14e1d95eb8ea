"""Parity of the colour-partitioned sliced-ELL multicolour HIP sweep (through the C-ABI) with the CPU oracle."""
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def dev(a):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a, np.float64), device="cuda")


def host(t):
    return t.detach().cpu().numpy()


def random_spd(n, density, seed):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="csr")
    M = M + M.T
    M = M + sp.diags(np.abs(M).sum(axis=1).A1 + 1.0)
    return O.CSR.from_scipy(M)


def cases():
    yield "lap9x9", O.shifted_laplace(9, 9, 1, 10.0)
    yield "lap6x5x4", O.shifted_laplace(6, 5, 4, 2.0)
    yield "lap40x30", O.shifted_laplace(40, 30, 1, 1e-2)
    yield "galerkin27", O.CSR.from_scipy(O.galerkin(O.shifted_laplace(9, 9, 9, 1.0).scipy(), O.q1_interp(5, 5, 5)))
    yield "random200", random_spd(200, 0.03, 1)
    yield "diagonal", O.CSR.from_scipy(sp.diags(np.arange(1.0, 8.0)))
    yield "one_row", O.CSR.from_scipy(sp.csr_matrix(np.array([[2.0]])))


@pytest.mark.parametrize("name,A", list(cases()), ids=[c[0] for c in cases()])
@pytest.mark.parametrize("rule", ["greedy", "lexlevels", "iterated"])
def test_deterministic_sweep_is_bit_exact(name, A, rule):
    from parmgmc_amd import COLORING_GREEDY, COLORING_ITERATED, COLORING_LEXLEVELS, MCSOR

    rng = np.random.default_rng(2)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    mc = MCSOR(A.rowptr, A.colidx, A.vals, {"greedy": COLORING_GREEDY, "lexlevels": COLORING_LEXLEVELS, "iterated": COLORING_ITERATED}[rule]).setup()
    col = mc.get_coloring()
    # colouring / index maps bit-exact against the build's stated rules
    assert np.array_equal(col, {"greedy": O.coloring_greedy, "lexlevels": O.coloring_lexlevels, "iterated": O.coloring_iterated}[rule](A))
    if rule == "iterated":
        assert col.max() <= O.coloring_greedy(A).max()  # never more classes than first-fit
    assert mc.get_num_colors() == col.max() + 1 and O.coloring_is_valid(A, col)
    for om in (1.0, 1.2):
        mc.set_omega(om)
        for t in (O.SOR_FORWARD, O.SOR_BACKWARD, O.SOR_SYMMETRIC):
            mc.set_sweep_type(t)
            yd = dev(y)
            mc.apply(dev(b), yd)
            assert np.array_equal(host(yd), O.mcsor_apply(A, col, b, y, om, t)), (name, rule, om, t)
            if rule == "lexlevels":
                # == the reference's serial path: one colour, plain lexicographic Gauss-Seidel (src/mc_sor.c:397-410)
                assert np.array_equal(host(yd), O.mcsor_apply(A, O.coloring_single(A.n), b, y, om, t))


def test_user_coloring_and_rejection_of_invalid_one():
    from parmgmc_amd import MCSOR, PMGError

    A = O.shifted_laplace(6, 5, 4, 2.0)
    rb = O.coloring_redblack(6, 5, 4)
    mc = MCSOR(A.rowptr, A.colidx, A.vals, user_colors=rb).setup()
    assert np.array_equal(mc.get_coloring(), rb) and mc.get_num_colors() == 2
    rng = np.random.default_rng(3)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    yd = dev(y)
    mc.apply(dev(b), yd)
    assert np.array_equal(host(yd), O.mcsor_apply(A, rb, b, y))
    # the reference's serial "all rows colour 0" is not a valid GPU colouring: refused, not silently raced
    bad = MCSOR(A.rowptr, A.colidx, A.vals, user_colors=np.zeros(A.n, np.int32))
    with pytest.raises(PMGError) as e:
        bad.setup()
    assert e.value.code == 62
    with pytest.raises(PMGError) as e:
        mc.set_sweep_type(4)
    assert e.value.code == 56


def test_golden_sweeps_csr():
    from parmgmc_amd import MCSOR

    sw = np.load(GOLD / "sweeps.npz")
    ops = np.load(GOLD / "operators.npz")
    for name, key in {"9x9": "lap_9x9_k10", "5x5x5": "lap_5x5x5_k10", "6x5x4": "lap_6x5x4_k2"}.items():
        mc = MCSOR(ops[key + "_rowptr"], ops[key + "_colidx"], ops[key + "_vals"]).setup()
        assert np.array_equal(mc.get_coloring(), ops[key + "_greedy"])
        for om in (1.0, 1.2):
            mc.set_omega(om)
            for tname, t in {"fwd": 1, "bwd": 2, "sym": 3}.items():
                mc.set_sweep_type(t)
                yd = dev(sw[f"{name}_y"])
                mc.apply(dev(sw[f"{name}_b"]), yd)
                assert np.array_equal(host(yd), sw[f"{name}_greedy_om{om}_{tname}"])


@pytest.mark.parametrize("name,A", [c for c in cases() if c[0] in ("lap9x9", "lap6x5x4", "galerkin27", "random200")], ids=["lap9x9", "lap6x5x4", "galerkin27", "random200"])
def test_noisy_chain_matches_oracle(name, A):
    from parmgmc_amd import MCSOR

    rng = np.random.default_rng(4)
    b, y0 = rng.standard_normal(A.n), rng.standard_normal(A.n)
    for om, scaled, t in [(1.0, True, O.SOR_FORWARD), (1.3, True, O.SOR_SYMMETRIC), (1.0, False, O.SOR_BACKWARD)]:
        mc = MCSOR(A.rowptr, A.colidx, A.vals).setup()
        mc.set_omega(om)
        mc.set_sweep_type(t)
        yd = dev(y0)
        nxt = mc.sample(dev(b), yd, 3, seed=0xCAFE, counter0=5, scaled=scaled)
        assert nxt == 5 + (6 if t == O.SOR_SYMMETRIC else 3)
        want = O.gibbs_samples(A, mc.get_coloring(), b, y0, 3, lambda d: O.noise_rows(A.n, 0xCAFE, 5 + d), om, t, scaled)
        assert np.abs(host(yd) - want).max() / np.abs(want).max() < 1e-13


def test_vec_set_random_standard_normal():
    import torch

    from parmgmc_amd import vec_set_random_standard_normal

    nz = np.load(GOLD / "noise.npz")
    x = torch.zeros(33, dtype=torch.float64, device="cuda")
    vec_set_random_standard_normal(x, 0xCAFE, 7)
    assert np.abs(host(x) - nz["rows_seed51966_sweep7_n33"]).max() < 1e-14
    x = torch.zeros(1_000_001, dtype=torch.float64, device="cuda")
    vec_set_random_standard_normal(x, 12345, 0)
    h = host(x)
    assert np.abs(h[:200000] - O.noise_rows(200000, 12345, 0)).max() < 1e-13
    from scipy import stats

    assert abs(h.mean()) < 5e-3 and abs(h.var() - 1) < 5e-3 and stats.kstest(h[::7], "norm").pvalue > 1e-3


def test_residual():
    from parmgmc_amd import MCSOR

    A = O.shifted_laplace(6, 5, 4, 2.0)
    rng = np.random.default_rng(5)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    mc = MCSOR(A.rowptr, A.colidx, A.vals).setup()
    r = dev(np.zeros(A.n))
    mc.residual(dev(b), dev(y), r)
    assert np.allclose(host(r), b - A.scipy() @ y, rtol=0, atol=1e-13)


def test_csr_and_grid_paths_agree_bitwise():
    from parmgmc_amd import MCSOR, GridMCSOR

    nx, ny, nz, kappa = 12, 7, 5, 1.5
    A = O.shifted_laplace(nx, ny, nz, kappa)
    rng = np.random.default_rng(6)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    mc = MCSOR(A.rowptr, A.colidx, A.vals).setup()
    g = GridMCSOR(nx, ny, nz, kappa)
    for om in (1.0, 0.9):
        mc.set_omega(om)
        g.set_omega(om)
        y1, y2 = dev(y), dev(y)
        mc.apply(dev(b), y1)
        g.apply(dev(b), y2)
        assert np.array_equal(host(y1), host(y2))
