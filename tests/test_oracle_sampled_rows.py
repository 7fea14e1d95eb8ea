"""The sampled-row oracle functions (oracle/pmg_oracle.c: orc_grid7_rows_sweep, orc_grid7_rows_residual,
orc_st27_rows, orc_q1_rows) are what the full-size GPU parity tests (tests/test_gpu_fullsize_oracle.py) compare the
kernels with at 256^3 ... 513^3.  Here they are pinned, bit for bit and for EVERY row, to the whole-vector oracle
functions that the reference's known-answer tests pin (tests/test_oracle_reference_kat.py): the multicolour sweep
of reference src/mc_sor.c:256-289 with the right-hand side of src/pc_mcgibbs.c:119-128, the residual, and the
scipy restatement of PCMG's Q1 transfers."""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("dims", [(7, 5, 4), (9, 9, 9), (6, 1, 1), (5, 4, 1)])
@pytest.mark.parametrize("omega,backward", [(1.0, False), (1.3, False), (1.0, True), (0.8, True)])
def test_grid7_rows_sweep_equals_whole_sweep(dims, omega, backward):
    nx, ny, nz = dims
    A = O.shifted_laplace(nx, ny, nz, 2.5)
    rng = np.random.default_rng(nx * 100 + ny)
    b, y0 = rng.standard_normal(A.n), rng.standard_normal(A.n)
    cols = O.coloring_redblack(nx, ny, nz)
    direction = O.SOR_BACKWARD if backward else O.SOR_FORWARD
    y1 = O.mcsor_apply(A, cols, b, y0, omega, direction)
    rows = np.arange(A.n)
    got = O.grid7_rows_sweep(nx, ny, nz, 2.5, rows, b, y0, y1, omega=omega, backward=backward)
    assert np.array_equal(got, y1)
    # noisy: one sample of the mcgibbs / sorgibbs loop
    for scaled in (True, False):
        if not scaled and omega != 1.0:
            continue
        y1n = O.gibbs_samples(A, cols, b, y0, 1, lambda d: O.noise_grid(nx, ny, nz, 77, 5), omega, direction, scaled)
        gotn = O.grid7_rows_sweep(nx, ny, nz, 2.5, rows, b, y0, y1n, omega=omega, backward=backward, noisy=True, scaled=scaled, seed=77, sweep=5)
        assert np.array_equal(gotn, y1n)


def test_grid7_rows_residual_equals_spmv():
    nx, ny, nz = 6, 7, 5
    A = O.shifted_laplace(nx, ny, nz, 0.7)
    rng = np.random.default_rng(3)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    want = b - A.scipy() @ y
    rows = rng.permutation(A.n)
    assert np.array_equal(O.grid7_rows_residual(nx, ny, nz, 0.7, rows, b, y), want[rows])


@pytest.mark.parametrize("backward", [False, True])
def test_st27_rows_equal_whole_sweep_of_the_galerkin_operator(backward):
    nf, nc = 9, 5
    A = O.shifted_laplace(nf, nf, nf, 3.0).scipy()
    P = O.q1_interp(nc, nc, nc)
    Ac = O.CSR.from_scipy(O.galerkin(A, P))
    coef, have, _exact = O.st27_table_from_csr(nc, nc, nc, Ac)
    assert have.all()
    # use the class table as the operator (the library's class-stencil levels do exactly that)
    rp, ci, v = [0], [], []
    for k in range(nc):
        for j in range(nc):
            for i in range(nc):
                cls = (0 if i == 0 else 2 if i == nc - 1 else 1) + 3 * (0 if j == 0 else 2 if j == nc - 1 else 1) + 9 * (0 if k == 0 else 2 if k == nc - 1 else 1)
                for dz in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            if 0 <= i + dx < nc and 0 <= j + dy < nc and 0 <= k + dz < nc:
                                ci.append(i + dx + nc * (j + dy + nc * (k + dz)))
                                v.append(coef[cls, 9 * (dz + 1) + 3 * (dy + 1) + dx + 1])
                rp.append(len(ci))
    At = O.CSR(rp, ci, v)
    n = nc ** 3
    rng = np.random.default_rng(11)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    cols = O.coloring_parity8(nc, nc, nc)
    direction = O.SOR_BACKWARD if backward else O.SOR_FORWARD
    sd = O.sqrtdiag(At, 1.2, True)
    sqrtd_cls = np.zeros(27)
    for k in range(nc):
        for j in range(nc):
            for i in range(nc):
                cls = (0 if i == 0 else 2 if i == nc - 1 else 1) + 3 * (0 if j == 0 else 2 if j == nc - 1 else 1) + 9 * (0 if k == 0 else 2 if k == nc - 1 else 1)
                sqrtd_cls[cls] = sd[i + nc * (j + nc * k)]
    rows = np.arange(n)
    y1 = O.mcsor_apply(At, cols, b, y0, 1.2, direction)
    assert np.array_equal(O.st27_rows_sweep(nc, nc, nc, coef, sqrtd_cls, rows, b, y0, y1, omega=1.2, backward=backward), y1)
    y1n = O.gibbs_samples(At, cols, b, y0, 1, lambda d: O.noise_rows(n, 9, 4), 1.2, direction, True)
    assert np.array_equal(O.st27_rows_sweep(nc, nc, nc, coef, sqrtd_cls, rows, b, y0, y1n, omega=1.2, backward=backward, noisy=True, seed=9, sweep=4), y1n)
    # residual with the diagonal last (the order the library documents for its stored-matrix residual)
    want = np.zeros(n)
    dp = O.diag_pointers(At)
    for r in range(n):
        s = 0.0
        for q in range(At.rowptr[r], At.rowptr[r + 1]):
            if q != dp[r]:
                s += At.vals[q] * y0[At.colidx[q]]
        s += At.vals[dp[r]] * y0[r]
        want[r] = b[r] - s
    assert np.array_equal(O.st27_rows_residual(nc, nc, nc, coef, rows, b, y0), want)


@pytest.mark.parametrize("nc", [(5, 5, 5), (5, 3, 1), (9, 5, 3)])
def test_q1_rows_equal_the_scipy_interpolation(nc):
    nf = tuple(2 * (c - 1) + 1 if c > 1 else 1 for c in nc)
    P = O.q1_interp(*nc)
    rng = np.random.default_rng(5)
    r = rng.standard_normal(P.shape[0])
    e = rng.standard_normal(P.shape[1])
    x = rng.standard_normal(P.shape[0])
    R = P.T.tocsr()
    R.sort_indices()
    assert np.array_equal(O.q1_rows_restrict(nf, nc, np.arange(P.shape[1]), r), R @ r)
    assert np.array_equal(O.q1_rows_prolong_add(nf, nc, np.arange(P.shape[0]), x, e), x + P @ e)
