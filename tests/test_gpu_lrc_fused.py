"""Round 4: the low-rank chain of a sweep in fewer launches (kernels_lrc.hip: noise draw + scale + B eta in one kernel, the
update that also puts the right-hand side back, one workgroup for small supports; PMG_LRC_FUSED=1 when the sampler is
built -- measured 2 % SLOWER than the chain of small kernels at 257^3, so not the default, pmg_lrc.c).  The fused kernels
keep every sum in the order of the kernels they replace, so whole chains must be BIT-IDENTICAL to the chain of small
kernels -- the form tests/test_lrc.py and tests/test_gpu_benchsize_lowrank.py pin against the oracle (reference src/mc_sor.c:101-112,
src/pc_mcgibbs.c:130-140, src/pc_sorgibbs.c:86-101).

Also the three folds that ARE the default since round 4 (the restore of b in the B^T y pass, the partial sums added by the update
kernel, the noise terms of a whole cycle drawn by one launch: PMG_LRC_RESTORE=0, PMG_LRC_REDUCE=0, PMG_LRC_BATCH=0 switch them off) and a third that is not (PMG_LRC_BTY=1: the repair in front of a
residual leaves the partial sums of B^T y_new for it; no gain measured), whole V-cycle chains bit for bit."""
import numpy as np
import pytest

import oracle as O
from test_lrc import ball_matrix, dev, host

pytestmark = pytest.mark.gpu


def _chain(monkeypatch, unfused, grid, levels, B, S, b, y0, sweep, omega, its):
    from parmgmc_amd import MGMC

    if unfused:
        monkeypatch.delenv("PMG_LRC_FUSED", raising=False)
    else:
        monkeypatch.setenv("PMG_LRC_FUSED", "1")
    mg = MGMC(*grid, 2.0, levels)
    mg.set_smoother(True, omega, sweep, 1)
    mg.set_lowrank(B, S)
    mg.setup()
    bd, yd = dev(b), dev(y0)
    mg.sample(bd, yd, its, seed=11, counter0=0)
    assert np.array_equal(host(bd), b), "the right-hand side must come back bit for bit"
    return host(yd).copy()


@pytest.mark.parametrize("sweep,omega", [(O.SOR_FORWARD, 1.0), (O.SOR_SYMMETRIC, 1.2), (O.SOR_BACKWARD, 1.0)])
def test_fused_chain_equals_the_chain_of_small_kernels_bit_for_bit(monkeypatch, sweep, omega):
    """small supports: every level takes the one-workgroup path (B^T y and its update in one launch)"""
    grid, levels = (33, 33, 17), 3
    n = int(np.prod(grid))
    B = ball_matrix(grid, [(0.3, 0.3, 0.4), (0.7, 0.6, 0.5), (0.5, 0.2, 0.8)], [0.12, 0.15, 0.1])
    S = np.array([50.0, 80.0, 30.0])
    rng = np.random.default_rng(5)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    got = _chain(monkeypatch, False, grid, levels, B, S, b, y0, sweep, omega, 3)
    want = _chain(monkeypatch, True, grid, levels, B, S, b, y0, sweep, omega, 3)
    assert np.isfinite(got).all() and np.array_equal(got, want)


@pytest.mark.parametrize("k", [3, 17])
def test_fused_chain_with_many_blocks_of_support_rows(monkeypatch, k):
    """129^3 with wide balls: tens of thousands of support rows on the two finest levels -> several blocks of 4096 rows, the
    last one to finish adds the partial sums (fixed order); odd k = a half-used Box-Muller pair in the noise kernel"""
    grid, levels = (129, 129, 129), 4
    n = int(np.prod(grid))
    rng = np.random.default_rng(k)
    ctr = [tuple(rng.uniform(0.25, 0.75, 3)) for _ in range(k)]
    B = ball_matrix(grid, ctr, list(rng.uniform(0.12, 0.16, k) if k == 3 else rng.uniform(0.06, 0.08, k)))
    ns = int((np.abs(B).sum(1) > 0).sum())
    assert ns > 3 * 4096, ns
    S = rng.uniform(20.0, 90.0, k)
    b, y0 = rng.standard_normal(n), np.zeros(n)
    got = _chain(monkeypatch, False, grid, levels, B, S, b, y0, O.SOR_FORWARD, 1.0, 2)
    want = _chain(monkeypatch, True, grid, levels, B, S, b, y0, O.SOR_FORWARD, 1.0, 2)
    assert np.isfinite(got).all() and np.array_equal(got, want)


def test_fused_chain_on_an_aij_operator(monkeypatch):
    """the standalone sampler on an AIJ operator (pmg_mcsor + MATLRC): fused and unfused chains agree bit for bit"""
    import torch

    from parmgmc_amd import MCSOR

    nx, ny, nz = 17, 17, 9
    A = O.shifted_laplace(nx, ny, nz, 3.0)
    n = A.n
    B = ball_matrix((nx, ny, nz), [(0.4, 0.4, 0.5), (0.6, 0.7, 0.3)], [0.2, 0.25])
    S = np.array([40.0, 60.0])
    rng = np.random.default_rng(1)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    outs = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.delenv("PMG_LRC_FUSED", raising=False)
        else:
            monkeypatch.setenv("PMG_LRC_FUSED", "1")
        mc = MCSOR(A.rowptr, A.colidx, A.vals)
        mc.set_omega(1.1)
        mc.set_sweep_type(O.SOR_SYMMETRIC)
        mc.setup()
        mc.set_lowrank(B, S)
        bd, yd = dev(b), dev(y0)
        mc.sample(bd, yd, 3, seed=4, counter0=0)
        torch.cuda.synchronize()
        assert np.array_equal(host(bd), b)
        outs.append(host(yd).copy())
    assert np.array_equal(outs[0], outs[1])


def _vcycle_chain(monkeypatch, env, grid, levels, B, S, b, y0, sweep, its):
    from parmgmc_amd import MGMC

    for key in ("PMG_LRC_FUSED", "PMG_LRC_RESTORE", "PMG_LRC_REDUCE", "PMG_LRC_BTY", "PMG_LRC_BATCH"):
        monkeypatch.delenv(key, raising=False)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    mg = MGMC(*grid, 2.0, levels)
    mg.set_smoother(True, 1.0, sweep, 1)
    mg.set_lowrank(B, S)
    mg.setup()
    bd, yd = dev(b), dev(y0)
    mg.sample(bd, yd, its, seed=23, counter0=0)
    assert np.array_equal(host(bd), b), "the right-hand side must come back bit for bit"
    return host(yd).copy()


@pytest.mark.parametrize("k", [3, 8, 11])
@pytest.mark.parametrize("sweep", [O.SOR_FORWARD, O.SOR_SYMMETRIC])
def test_default_folds_equal_their_switched_off_forms_bit_for_bit(monkeypatch, k, sweep):
    """65^3 x 33, 4 levels: several blocks of support rows on the two finest levels; k = 8 is the last rank whose partial sums
    the update kernel adds itself, k = 11 takes lrc_reduce_kernel either way"""
    grid, levels = (65, 65, 33), 4
    n = int(np.prod(grid))
    rng = np.random.default_rng(100 + k)
    ctr = [tuple(rng.uniform(0.25, 0.75, 3)) for _ in range(k)]
    lo, hi = {3: (0.18, 0.22), 8: (0.12, 0.15), 11: (0.10, 0.13)}[k]
    B = ball_matrix(grid, ctr, list(rng.uniform(lo, hi, k)))
    assert 4096 < int((np.abs(B).sum(1) > 0).sum()) < n // 6  # row-compact form, several blocks
    S = rng.uniform(20.0, 90.0, k)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    want = _vcycle_chain(monkeypatch, {"PMG_LRC_RESTORE": "0", "PMG_LRC_REDUCE": "0", "PMG_LRC_BATCH": "0"}, grid, levels, B, S, b, y0, sweep, 3)
    assert np.isfinite(want).all()
    for env in ({}, {"PMG_LRC_BATCH": "0"}, {"PMG_LRC_BTY": "1"}, {"PMG_LRC_REDUCE": "0"}, {"PMG_LRC_RESTORE": "0"}, {"PMG_LRC_BTY": "1", "PMG_LRC_RESTORE": "0"}):
        got = _vcycle_chain(monkeypatch, env, grid, levels, B, S, b, y0, sweep, 3)
        assert np.array_equal(got, want), env
