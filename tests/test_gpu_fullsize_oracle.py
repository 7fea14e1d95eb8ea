"""Oracle parity AT BASELINE.json's full sizes (256^3, 512^3 sweeps; 257^3, 513^3 hierarchies).

A whole CPU sweep of 1.3e8 rows is out of reach for a test, but the reference's row update (src/mc_sor.c:256-271)
is local: given the vector before and after ONE directional sweep, the value the reference loop gives any row follows
from the colour rule (neighbours of an earlier colour are read new, of a later colour old).  So every kernel is run
once at full size and ~5e5 rows -- 1e5 random ones plus the first and last rows of every plane, of the lines at the
plane edges, at the wavefront seams (256 grid columns per wavefront) and at the XCD band edges -- are recomputed by
the oracle (oracle/pmg_oracle.c: orc_grid7_rows_sweep / _residual, orc_st27_rows, orc_q1_rows, themselves pinned row
for row to the whole-vector oracle in tests/test_oracle_sampled_rows.py).  Bit equality for deterministic kernels,
1e-13 (relative to the largest entry) for sweeps with in-kernel noise: the device's log / sincos differ from glibc's
in the last bits.  This catches what the property tests of test_gpu_fullsize.py cannot: an indexing fault that the
sweep and the residual kernel share (row offsets beyond 2^31 bytes, band / tail / packed mappings)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def sample_rows(nx, ny, nz, nrand=100_000, seed=0):
    rng = np.random.default_rng(seed)

    def edges(n, extra=()):
        s = set(range(0, min(n, 5))) | set(range(max(0, n - 5), n))
        for e in extra:
            for d in (-2, -1, 0, 1, 2):
                if 0 <= e + d < n:
                    s.add(e + d)
        return np.array(sorted(s), np.int64)

    band = (ny + 7) // 8
    ie = edges(nx, [256 * q for q in range(1, nx // 256 + 1)] + [nx // 2])
    je = edges(ny, [band * q for q in range(1, 8)] + [4 * (ny // 8), ny // 2])
    ke = edges(nz, [nz // 2])
    ia, ja, ka = np.arange(nx, dtype=np.int64), np.arange(ny, dtype=np.int64), np.arange(nz, dtype=np.int64)

    def box(i, j, k):
        return (i[None, None, :] + nx * (j[None, :, None] + ny * k[:, None, None])).ravel()

    parts = [box(ie, je, ka), box(ie, ja, ke), box(ia, je, ke), rng.integers(0, nx * ny * nz, nrand, dtype=np.int64)]
    return np.unique(np.concatenate(parts))


def _rand(n, seed):
    import torch

    gen = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)


def _close(got, want, tol=1e-13):
    return float(np.abs(got - want).max()) <= tol * float(np.abs(want).max())


@pytest.mark.parametrize("n,kappa", [(256, 10.0), (512, 10.0), (257, 3.0), (513, 10.0)])
def test_grid_sweep_and_residual_rows_match_the_oracle(n, kappa):
    """grid_color_sweep_kernel (all instantiations the sizes select: banded, TAIL, packed) and grid_residual_kernel"""
    import torch

    from parmgmc_amd import GridMCSOR

    g = GridMCSOR(n, n, n, kappa)
    b, y0 = _rand(g.n, n), _rand(g.n, n + 1)
    bc = g.to_cvec(b)
    bh, y0h = b.cpu().numpy(), y0.cpu().numpy()
    rows = sample_rows(n, n, n)
    cases = [(1.0, False), (1.0, True), (1.25, False)] if n != 513 else [(1.0, False)]
    for omega, backward in cases:
        g.set_omega(omega)
        g.set_sweep_type(2 if backward else 1)
        yc = g.to_cvec(y0)
        g.apply_cvec(bc, yc)  # MCSORApply: deterministic
        y1h = g.from_cvec(yc).cpu().numpy()
        want = O.grid7_rows_sweep(n, n, n, kappa, rows, bh, y0h, y1h, omega=omega, backward=backward)
        assert np.array_equal(y1h[rows], want), f"deterministic sweep, omega {omega}, backward {backward}: {int((y1h[rows] != want).sum())} of {len(rows)} rows differ"
        if n == 256:  # negative control: the check notices a wrong sweep order and one wrong neighbour value
            assert not np.array_equal(y1h[rows], O.grid7_rows_sweep(n, n, n, kappa, rows, bh, y0h, y1h, omega=omega, backward=not backward))
            q = int(rows[len(rows) // 2])
            q = q if q % n < n - 1 else q - 1  # a row with an east neighbour
            y0x, y1x = y0h.copy(), y1h.copy()
            y0x[q + 1] += 1.0
            y1x[q + 1] += 1.0
            bad = O.grid7_rows_sweep(n, n, n, kappa, np.array([q]), bh, y0x, y1x, omega=omega, backward=backward)
            assert bad[0] != y1h[q]
        yc = g.to_cvec(y0)
        g.sample_cvec(bc, yc, 1, seed=0xCAFE, counter0=7, scaled=True)  # one mcgibbs sample = one noisy sweep, draw 7
        y1h = g.from_cvec(yc).cpu().numpy()
        want = O.grid7_rows_sweep(n, n, n, kappa, rows, bh, y0h, y1h, omega=omega, backward=backward, noisy=True, scaled=True, seed=0xCAFE, sweep=7)
        assert _close(y1h[rows], want), f"noisy sweep, omega {omega}, backward {backward}"
    g.set_omega(1.0)
    g.set_sweep_type(1)
    # residual r = b - A y
    r = g.new_cvec()
    g.residual_cvec(bc, g.to_cvec(y0), r)
    rh = g.from_cvec(r).cpu().numpy()
    assert np.array_equal(rh[rows], O.grid7_rows_residual(n, n, n, kappa, rows, bh, y0h))
    del g, bc, r
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,levels", [(257, 5), (513, 6)])
def test_vcycle_kernels_rows_match_the_oracle(n, levels):
    """Q1 restriction / prolongation from the grid level, and on the first class-stencil level (129^3 resp. 257^3): one
    directional sweep (deterministic forward + backward, noisy), the residual, and the transfers to the next level"""
    import torch

    from parmgmc_amd import MGMC, GridMCSOR

    kappa = 10.0
    mg = MGMC(n, n, n, kappa, levels).setup()
    top = levels - 1
    g = GridMCSOR(n, n, n, kappa)  # same cvec layout as the hierarchy's fine level
    kind, ld, off = mg.level_layout(top)
    assert kind == 0 and ld == g.cvec_len
    nc = (n - 1) // 2 + 1
    kc, ldc, offc = mg.level_layout(top - 1)
    assert kc == 1 and offc == nc * nc and ldc == nc * nc * (nc + 2)
    # --- grid level: restriction and prolongation -----------------------------------------------------------------
    r = _rand(n ** 3, 1)
    bcoarse = torch.zeros(ldc, dtype=torch.float64, device="cuda")
    mg.level_restrict(top, g.to_cvec(r), bcoarse)
    crow = sample_rows(nc, nc, nc, 50_000)
    got = bcoarse[offc:offc + nc ** 3].cpu().numpy()
    assert np.array_equal(got[crow], O.q1_rows_restrict((n, n, n), (nc, nc, nc), crow, r.cpu().numpy()))
    assert float(bcoarse[:offc].abs().max()) == 0.0 and float(bcoarse[offc + nc ** 3:].abs().max()) == 0.0  # ghost planes untouched
    e = torch.zeros(ldc, dtype=torch.float64, device="cuda")
    e[offc:offc + nc ** 3] = _rand(nc ** 3, 2)
    x0 = _rand(n ** 3, 3)
    xc = g.to_cvec(x0)
    mg.level_prolong_add(top, e, xc)
    frow = sample_rows(n, n, n)
    got = g.from_cvec(xc).cpu().numpy()
    assert np.array_equal(got[frow], O.q1_rows_prolong_add((n, n, n), (nc, nc, nc), frow, x0.cpu().numpy(), e[offc:offc + nc ** 3].cpu().numpy()))
    del xc, x0, r, g
    torch.cuda.empty_cache()
    # --- first class-stencil level ----------------------------------------------------------------------------------
    coef, sqrtd = mg.level_stencil(top - 1)
    N = nc ** 3

    def padded(v):
        out = torch.zeros(ldc, dtype=torch.float64, device="cuda")
        out[offc:offc + N] = v
        return out

    b, y0 = _rand(N, 4), _rand(N, 5)
    bp = padded(b)
    bh, y0h = b.cpu().numpy(), y0.cpu().numpy()
    for backward in (False, True):
        yp = padded(y0)
        mg.level_sweep(top - 1, bp, yp, backward=backward)
        y1h = yp[offc:offc + N].cpu().numpy()
        want = O.st27_rows_sweep(nc, nc, nc, coef, sqrtd, crow, bh, y0h, y1h, omega=1.0, backward=backward)
        assert np.array_equal(y1h[crow], want), f"class-stencil sweep, backward {backward}: {int((y1h[crow] != want).sum())} rows differ"
        assert float(yp[:offc].abs().max()) == 0.0 and float(yp[offc + N:].abs().max()) == 0.0
    yp = padded(y0)
    mg.level_sweep(top - 1, bp, yp, noisy=True, seed=0xBEEF, counter=3)
    y1h = yp[offc:offc + N].cpu().numpy()
    want = O.st27_rows_sweep(nc, nc, nc, coef, sqrtd, crow, bh, y0h, y1h, omega=1.0, noisy=True, seed=0xBEEF, sweep=3)
    assert _close(y1h[crow], want)
    rp = torch.zeros(ldc, dtype=torch.float64, device="cuda")
    mg.level_residual(top - 1, bp, padded(y0), rp)
    assert np.array_equal(rp[offc:offc + N].cpu().numpy()[crow], O.st27_rows_residual(nc, nc, nc, coef, crow, bh, y0h))
    # transfers between the class-stencil level and the next coarser one
    n2 = (nc - 1) // 2 + 1
    k2, ld2, off2 = mg.level_layout(top - 2)
    assert off2 == n2 * n2
    b2 = torch.zeros(ld2, dtype=torch.float64, device="cuda")
    mg.level_restrict(top - 1, padded(b), b2)
    rows2 = sample_rows(n2, n2, n2, 20_000)
    assert np.array_equal(b2[off2:off2 + n2 ** 3].cpu().numpy()[rows2], O.q1_rows_restrict((nc, nc, nc), (n2, n2, n2), rows2, bh))
    e2 = torch.zeros(ld2, dtype=torch.float64, device="cuda")
    e2[off2:off2 + n2 ** 3] = _rand(n2 ** 3, 6)
    xp = padded(y0)
    mg.level_prolong_add(top - 1, e2, xp)
    assert np.array_equal(xp[offc:offc + N].cpu().numpy()[crow], O.q1_rows_prolong_add((nc, nc, nc), (n2, n2, n2), crow, y0h, e2[off2:off2 + n2 ** 3].cpu().numpy()))
    mg.destroy()
