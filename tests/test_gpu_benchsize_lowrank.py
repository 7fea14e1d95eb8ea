"""Oracle / numpy parity of the low-rank (MATLRC) pieces at the size bench.py times for BASELINE config 5.

bench.py's `secondary_mgmc_lowrank` line runs the 5-level MGMC chain on A + B S B^T at 257^3 with k ball observations
(reference src/obs.c:135-180, examples/ex4.c:150-168).  tests/test_lrc.py compares that chain with the oracle on grids
up to 17 x 9 x 9; here every low-rank step of the cycle is run ONCE at 257^3 (k = 3 as in the bench line, and k = 17)
and checked on its support rows:

  B_l on the grid level = the caller's B (bit-exact); B_{l-1} = P^T B_l (src/pc_gamgmc.c:177-178) vs a numpy Q1
      restriction (1e-13: the sums run in a different order)
  Bb = C (S^-1 + B^T C)^-1, C = one deterministic sweep on B from zero (MCSORBuildLRCCorrection, src/mc_sor.c:480-544):
      C is recovered from the device's B, Bb and every support row of it is recomputed by the oracle's row sweep
  y -= Bb (B^T y) (MCSORPostSOR_LRC, src/mc_sor.c:101-112) and r -= B S B^T x (the MATLRC residual, src/pc_gamgmc.c:194)
      vs numpy on the support rows, untouched rows bit-equal
  the RESTRICTED residual term b_{l-1} -= B_{l-1} (S B_l^T x) the cycle uses behind its fused residual + restriction,
      vs numpy; and two whole samples with that form against two samples in the reference's operation order
      (pmg_mgmc_set_fused_transfers(0): residual, low-rank term, then restriction) at 1e-12 -- the stated tolerance of
      this deliberate operation-order difference.
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

N, LEVELS, KAPPA = 257, 5, 10.0


def bench_observations(k):
    """the observation set of bench.py's mgmc_lowrank_secondary"""
    centres = [(0.25, 0.25, 0.25), (0.75, 0.75, 0.75), (0.25, 0.75, 0.5)] + [(0.5, 0.5, 0.1 + 0.8 * q / max(1, k - 4)) for q in range(max(0, k - 3))]
    radii = ([0.1, 0.15, 0.1] + [0.08] * max(0, k - 3))[:k]
    return np.asarray(centres[:k]).ravel(), radii, np.resize([1.0, -1.0], k)


def restrict_q1(col, nf, nc):
    """P^T col for the Q1 interpolation nc^3 -> nf^3 (natural order, x fastest), one direction after the other"""
    P1 = O.q1_interp_1d(nc)  # nf x nc
    a = col.reshape(nf, nf, nf)  # [z, y, x]
    a = (P1.T @ a.reshape(nf, -1)).reshape(nc, nf, nf)  # z
    a = np.stack([P1.T @ a[q] for q in range(nc)])  # y
    a = (a.reshape(-1, nf) @ P1).reshape(nc, nc, nc)  # x
    return a.ravel()


def rel(got, want):
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-300)


@pytest.mark.parametrize("k", [3, 17])
def test_lowrank_steps_at_257(k):
    import torch

    from parmgmc_amd import MGMC, GridMCSOR, make_observation_mats

    n, top = N, LEVELS - 1
    coords, radii, vals = bench_observations(k)
    B, S, f = make_observation_mats(n, n, n, coords, radii, vals, 1e-4)
    mg = MGMC(n, n, n, KAPPA, LEVELS)
    mg.set_lowrank(B, S)
    mg.setup()
    g = GridMCSOR(n, n, n, KAPPA)
    kind, ld, _ = mg.level_layout(top)
    iota = torch.arange(1, g.n + 1, dtype=torch.float64, device="cuda")
    nat_of_pos = g.to_cvec(iota).cpu().numpy().astype(np.int64) - 1  # -1: pad slot
    assert kind == 0 and ld == len(nat_of_pos)
    del iota

    # ---- the grid level's factors ----
    rows, Bl, Bf, Bb = mg.level_lowrank_factors(top)
    nat = nat_of_pos[rows]
    assert (nat >= 0).all() and np.all(np.diff(rows) > 0)
    assert np.array_equal(Bl, B[nat, :])  # the caller's B on the support rows, bit for bit
    support = np.flatnonzero(np.abs(B).sum(1) > 0)
    assert np.isin(support, nat).all() and len(rows) < g.n // 4  # row-compact form; nothing of B outside it

    # Bb against its definition: C = Bb (S^-1 + B^T C), (I - B^T Bb) (S^-1 + B^T C) = S^-1
    kappa2 = KAPPA
    cols = range(k) if k <= 3 else (0, 2, k - 1)
    for backward, Bx in ((False, Bf), (True, Bb)):
        T = np.linalg.solve(np.eye(k) - Bl.T @ Bx, np.diag(1.0 / S))
        Cc = Bx @ T
        for c in cols:
            y1 = np.zeros(g.n)
            y1[nat] = Cc[:, c]
            want = O.grid7_rows_sweep(n, n, n, kappa2, nat, B[:, c].copy(), np.zeros(g.n), y1, omega=1.0, backward=backward)
            assert rel(Cc[:, c], want) < 1e-9, (k, backward, c)
        # negative control: the check tells the two sweep directions apart (they differ by ~1e-6 on this operator)
        wrong = O.grid7_rows_sweep(n, n, n, kappa2, nat, B[:, c].copy(), np.zeros(g.n), y1, omega=1.0, backward=not backward)
        assert rel(Cc[:, c], wrong) > 1e-8

    # ---- y -= Bb (B^T y) and r -= B (S o B^T x) on cvecs ----
    gen = torch.Generator(device="cuda").manual_seed(5 + k)
    y = torch.randn(ld, dtype=torch.float64, device="cuda", generator=gen)
    yh = y.cpu().numpy()
    for backward, Bx in ((False, Bf), (True, Bb)):
        y2 = y.clone()
        mg.level_lowrank_post(top, y2, backward=backward)
        want = yh.copy()
        want[rows] -= Bx @ (Bl.T @ yh[rows])
        got = y2.cpu().numpy()
        mask = np.ones(ld, bool)
        mask[rows] = False
        assert np.array_equal(got[mask], yh[mask])
        assert rel(got[rows], want[rows]) < 1e-13
        # the repair itself is ~1e-10 of a random y (it would drown in the rounding of y - corr): a y that is zero except
        # at one row inside every ball shows it without cancellation
        ys = np.zeros(ld)
        hot = [int(np.flatnonzero(Bl[:, c])[len(np.flatnonzero(Bl[:, c])) // 2]) for c in range(k)]
        ys[rows[hot]] = 1.0 + np.arange(k)
        y3 = torch.as_tensor(ys, device="cuda")
        mg.level_lowrank_post(top, y3, backward=backward)
        corr = Bx @ (Bl.T @ ys[rows])
        cold = np.ones(len(rows), bool)
        cold[hot] = False
        assert np.abs(corr[cold]).max() > 0 and rel(-y3.cpu().numpy()[rows][cold], corr[cold]) < 1e-13
    r = torch.randn(ld, dtype=torch.float64, device="cuda", generator=gen)
    rh = r.cpu().numpy()
    mg.level_lowrank_residual_sub(top, y, r, restricted=False)
    wk = S * (Bl.T @ yh[rows])
    want = rh.copy()
    want[rows] -= Bl @ wk
    got = r.cpu().numpy()
    assert np.array_equal(got[mask], rh[mask]) and rel(got[rows], want[rows]) < 1e-13
    rz = torch.zeros(ld, dtype=torch.float64, device="cuda")  # the term alone: no cancellation against r
    mg.level_lowrank_residual_sub(top, y, rz, restricted=False)
    assert rel(-rz.cpu().numpy()[rows], Bl @ wk) < 1e-12

    # ---- the next level: B_{l-1} = P^T B_l, and the restricted residual term ----
    kind1, ld1, off1 = mg.level_layout(top - 1)
    nc = (n - 1) // 2 + 1
    assert kind1 == 1 and mg.level_dims(top - 1) == (nc, nc, nc)
    rows1, B1, B1f, _ = mg.level_lowrank_factors(top - 1)
    nat1 = rows1 - off1
    assert nat1.min() >= 0 and nat1.max() < nc ** 3
    for c in cols:
        want = restrict_q1(B[:, c], n, nc)
        assert rel(B1[:, c], want[nat1]) < 1e-13
        out = np.ones(nc ** 3, bool)
        out[nat1] = False
        assert not want[out].any()  # nothing of P^T B outside the compact rows
    bc = torch.randn(ld1, dtype=torch.float64, device="cuda", generator=gen)
    bch = bc.cpu().numpy()
    mg.level_lowrank_residual_sub(top, y, bc, restricted=True)
    want = bch.copy()
    want[rows1] -= B1 @ wk
    got = bc.cpu().numpy()
    mask1 = np.ones(ld1, bool)
    mask1[rows1] = False
    assert np.array_equal(got[mask1], bch[mask1]) and rel(got[rows1], want[rows1]) < 1e-13
    bz = torch.zeros(ld1, dtype=torch.float64, device="cuda")  # the term alone
    mg.level_lowrank_residual_sub(top, y, bz, restricted=True)
    term = -bz.cpu().numpy()
    assert not term[mask1].any() and rel(term[rows1], B1 @ wk) < 1e-12
    # ... which is the restriction of the unrestricted term: P^T (B wk) = (P^T B) wk
    full = np.zeros(g.n)
    full[nat] = Bl @ wk
    assert rel(term[rows1], restrict_q1(full, n, nc)[nat1]) < 1e-12
    # the class-stencil level's own repair
    y1v = torch.randn(ld1, dtype=torch.float64, device="cuda", generator=gen)
    y1h = y1v.cpu().numpy()
    mg.level_lowrank_post(top - 1, y1v, backward=False)
    want = y1h.copy()
    want[rows1] -= B1f @ (B1.T @ y1h[rows1])
    assert rel(y1v.cpu().numpy()[rows1], want[rows1]) < 1e-13
    del y, r, bc, y1v, y2

    # ---- two whole samples: restricted form (default) against the reference's operation order ----
    b = torch.as_tensor(f, device="cuda")
    out = []
    for fused in (True, False):
        mg.set_fused_transfers(fused)
        yy = torch.zeros(g.n, dtype=torch.float64, device="cuda")
        mg.sample(b, yy, 2, seed=0xCAFE, counter0=0)
        out.append(yy.cpu().numpy())
    assert not np.array_equal(out[0], out[1])  # the switch does select another operation order
    assert rel(out[0], out[1]) < 1e-12
    mg.destroy()
    del g
    torch.cuda.empty_cache()
