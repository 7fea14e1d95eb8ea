"""64-bit PetscInt across the C-ABI (reference include/parmgmc/parmgmc.h:18-24: the library builds against either
index width; the arrays come from MatSeqAIJGetCSRAndMemType, src/mc_sor.c:250).  The _idx entry points with
idx_width = 64 must give exactly what the 32-bit ones give, and that is the oracle's result."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a, np.float64), device="cuda")


@pytest.mark.parametrize("dims", [(9, 9, 1), (6, 5, 4)])
def test_mcsor_with_64bit_indices_matches_the_oracle(dims):
    from parmgmc_amd import MCSOR

    A = O.shifted_laplace(*dims, 3.0)
    rng = np.random.default_rng(1)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    mc = MCSOR(A.rowptr, A.colidx, A.vals, idx_width=64)
    # the 64-bit arrays were narrowed inside the call: the caller may drop them before set-up
    mc._rowptr = mc._colidx = None
    mc.setup()
    col = mc.get_coloring()
    assert np.array_equal(col, O.coloring_greedy(A))
    for om, t in ((1.0, O.SOR_FORWARD), (1.3, O.SOR_SYMMETRIC)):
        mc.set_omega(om)
        mc.set_sweep_type(t)
        yd = dev(y)
        mc.apply(dev(b), yd)
        assert np.array_equal(yd.cpu().numpy(), O.mcsor_apply(A, col, b, y, om, t))
    mc.set_omega(1.0)
    mc.set_sweep_type(O.SOR_FORWARD)
    yd = dev(y)
    mc.sample(dev(b), yd, 2, seed=5, counter0=0, scaled=True)
    want = O.gibbs_samples(A, col, b, y, 2, lambda d: O.noise_rows(A.n, 5, d), 1.0, O.SOR_FORWARD, True)
    assert np.abs(yd.cpu().numpy() - want).max() <= 1e-13 * np.abs(want).max()


def test_hierarchy_and_cholesky_with_64bit_indices_equal_the_32bit_path():
    import torch

    from parmgmc_amd import MGMC, CholSampler

    nf, nc = 9, 5
    A1 = O.shifted_laplace(nf, nf, nf, 2.0)
    P = O.q1_interp(nc, nc, nc)
    A0 = O.CSR.from_scipy(O.galerkin(A1.scipy(), P))
    Pc = O.CSR.from_scipy(P)
    ops = [(A0.rowptr, A0.colidx, A0.vals), (A1.rowptr, A1.colidx, A1.vals)]
    ps = [None, (Pc.rowptr, Pc.colidx, Pc.vals)]
    b = torch.ones(A1.n, dtype=torch.float64, device="cuda")
    out = []
    for w in (32, 64):
        mg = MGMC.from_hierarchy(ops, ps, idx_width=w)
        mg.setup()
        y = torch.zeros(A1.n, dtype=torch.float64, device="cuda")
        mg.sample(b, y, 3, seed=11)
        out.append(y.cpu().numpy())
        mg.destroy()
    assert np.array_equal(out[0], out[1]) and np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
    L32 = CholSampler(A0.rowptr, A0.colidx, A0.vals).factor()
    L64 = CholSampler(A0.rowptr, A0.colidx, A0.vals, idx_width=64).factor()
    assert np.array_equal(L32, L64)
