"""Parity of the matrix-free red-black HIP sweep (through the C-ABI) with the CPU oracle."""
from pathlib import Path

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
SWEEPS = {"fwd": O.SOR_FORWARD, "bwd": O.SOR_BACKWARD, "sym": O.SOR_SYMMETRIC}


def dev(a):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a, np.float64), device="cuda")


def host(t):
    return t.detach().cpu().numpy()


GRIDS = [(9, 9, 1, 10.0), (5, 5, 5, 10.0), (6, 5, 4, 2.0), (2, 2, 2, 1.0), (3, 1, 1, 1.0), (33, 7, 3, 0.5), (64, 6, 5, 10.0), (70, 3, 2, 1.0), (128, 128, 1, 1e-4)]
# lines of 64 m + a few threads (the multigrid sizes 2^k+1): full wavefronts + collected tail threads, kernels_grid.hip
TAIL_GRIDS = [(257, 5, 3, 10.0), (261, 3, 2, 1.0), (287, 2, 1, 0.5), (513, 2, 2, 2.0), (257, 70, 2, 1.0)]


@pytest.mark.parametrize("grid", GRIDS + TAIL_GRIDS)
@pytest.mark.parametrize("om", [1.0, 1.2])
def test_deterministic_sweep_is_bit_exact(grid, om):
    """MCSORApply: fwd / bwd / sym on the HIP path == reference loop (src/mc_sor.c:256-289), bit for bit."""
    from parmgmc_amd import GridMCSOR

    nx, ny, nz, kappa = grid
    A = O.shifted_laplace(nx, ny, nz, kappa)
    rng = np.random.default_rng(nx * 131 + ny * 17 + nz)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    g = GridMCSOR(nx, ny, nz, kappa)
    assert np.array_equal(g.get_coloring(), O.coloring_redblack(nx, ny, nz))
    g.set_omega(om)
    for name, t in SWEEPS.items():
        g.set_sweep_type(t)
        yd = dev(y)
        g.apply(dev(b), yd)
        want = O.mcsor_apply(A, O.coloring_redblack(nx, ny, nz), b, y, om, t)
        assert np.array_equal(host(yd), want), (grid, om, name, np.abs(host(yd) - want).max())


def test_golden_sweeps():
    from parmgmc_amd import GridMCSOR

    sw = np.load(GOLD / "sweeps.npz")
    for name, (nx, ny, nz, kappa) in {"9x9": (9, 9, 1, 10.0), "5x5x5": (5, 5, 5, 10.0), "6x5x4": (6, 5, 4, 2.0)}.items():
        g = GridMCSOR(nx, ny, nz, kappa)
        for om in (1.0, 1.2):
            g.set_omega(om)
            for tname, t in SWEEPS.items():
                g.set_sweep_type(t)
                yd = dev(sw[f"{name}_y"])
                g.apply(dev(sw[f"{name}_b"]), yd)
                assert np.array_equal(host(yd), sw[f"{name}_redblack_om{om}_{tname}"])


def test_ex5_symmetric_equals_forward_then_backward_on_device():
    """reference examples/ex5.c:53-70 on the HIP path (||.||_2 < 1e-15; here exactly 0)."""
    from parmgmc_amd import GridMCSOR

    rng = np.random.default_rng(0)
    b, x = rng.random(81), rng.random(81)
    g = GridMCSOR(9, 9, 1, 1.0)
    xd = dev(x)
    g.set_sweep_type(O.SOR_FORWARD)
    g.apply(dev(b), xd)
    g.set_sweep_type(O.SOR_BACKWARD)
    g.apply(dev(b), xd)
    yd = dev(x)
    g.set_sweep_type(O.SOR_SYMMETRIC)
    g.apply(dev(b), yd)
    assert np.linalg.norm(host(xd) - host(yd)) < 1e-15


def test_cvec_roundtrip_and_padding():
    from parmgmc_amd import GridMCSOR

    for (nx, ny, nz) in [(9, 9, 1), (33, 7, 3), (64, 6, 5)]:
        g = GridMCSOR(nx, ny, nz, 1.0)
        x = np.random.default_rng(1).standard_normal(nx * ny * nz)
        cv = g.to_cvec(dev(x))
        assert np.array_equal(host(g.from_cvec(cv)), x)
        # pad slots and ghost planes stay zero; every value appears exactly once
        h = host(cv)
        assert np.count_nonzero(h) == x.size and np.isclose(np.sort(h[h != 0]), np.sort(x)).all()


@pytest.mark.parametrize("grid", [(9, 9, 1, 10.0), (6, 5, 4, 2.0), (33, 7, 3, 0.5), (70, 3, 2, 1.0), (128, 128, 1, 10.0)] + TAIL_GRIDS[:4])  # (128, 128, 1) is BASELINE config 0 (ex1.c at 128x128)
def test_noisy_chain_matches_oracle(grid):
    """PCApplyRichardson_MulticolorGibbs / _SORGibbs sample loop with in-kernel Philox + Box-Muller noise vs
    the oracle (libm log/cos/sin): tolerance 1e-13 relative to max|y| (device log/sincospi differ from glibc in
    the last ulp; everything else is bit-identical)."""
    from parmgmc_amd import GridMCSOR

    nx, ny, nz, kappa = grid
    A = O.shifted_laplace(nx, ny, nz, kappa)
    rng = np.random.default_rng(7)
    b, y0 = rng.standard_normal(A.n), rng.standard_normal(A.n)
    rb = O.coloring_redblack(nx, ny, nz)
    for om, scaled, t in [(1.0, True, O.SOR_FORWARD), (1.3, True, O.SOR_SYMMETRIC), (1.0, False, O.SOR_BACKWARD), (0.8, True, O.SOR_BACKWARD)]:
        g = GridMCSOR(nx, ny, nz, kappa)
        g.set_omega(om)
        g.set_sweep_type(t)
        yd = dev(y0)
        nxt = g.sample(dev(b), yd, 3, seed=0xCAFE, counter0=5, scaled=scaled)
        assert nxt == 5 + (6 if t == O.SOR_SYMMETRIC else 3)
        want = O.gibbs_samples(A, rb, b, y0, 3, lambda d: O.noise_grid(nx, ny, nz, 0xCAFE, 5 + d), om, t, scaled)
        err = np.abs(host(yd) - want).max() / np.abs(want).max()
        assert err < 1e-13, (grid, om, scaled, t, err)


def test_golden_chains():
    from parmgmc_amd import GridMCSOR

    ch = np.load(GOLD / "chains.npz")
    for name, (nx, ny, nz, kappa) in {"9x9": (9, 9, 1, 10.0), "6x5x4": (6, 5, 4, 2.0)}.items():
        for om, scaled, tname, t in [(1.0, True, "fwd", O.SOR_FORWARD), (1.3, True, "sym", O.SOR_SYMMETRIC), (1.0, False, "bwd", O.SOR_BACKWARD)]:
            g = GridMCSOR(nx, ny, nz, kappa)
            g.set_omega(om)
            g.set_sweep_type(t)
            yd = dev(ch[f"{name}_y0"])
            g.sample(dev(ch[f"{name}_b"]), yd, 3, seed=0xCAFE, counter0=5, scaled=scaled)
            want = ch[f"{name}_grid_om{om}_{'mc' if scaled else 'sor'}_{tname}"]
            assert np.abs(host(yd) - want).max() / np.abs(want).max() < 1e-13


def test_unscaled_noise_requires_omega_one():
    from parmgmc_amd import GridMCSOR, PMGError

    g = GridMCSOR(9, 9, 1, 1.0)
    g.set_omega(1.2)
    with pytest.raises(PMGError) as e:
        g.sample(dev(np.zeros(81)), dev(np.zeros(81)), 1, seed=1, scaled=False)
    assert e.value.code == 56


@pytest.mark.parametrize("om,sweep", [(1.0, O.SOR_FORWARD), (1.3, O.SOR_SYMMETRIC)])
def test_device_chain_has_exact_stationary_covariance(om, sweep):
    """North-star "covariance error within 1e-6": build the chain's linear maps column by column FROM THE HIP
    KERNEL (G from unit initial states, N from unit right-hand sides scaled by the noise factor), solve the
    Lyapunov equation and compare with the dense A^-1 in the metric of src/stats.c: < 1e-10."""
    from parmgmc_amd import GridMCSOR

    nx, ny, nz, kappa = 5, 5, 5, 10.0
    A = O.shifted_laplace(nx, ny, nz, kappa)
    n = A.n
    g = GridMCSOR(nx, ny, nz, kappa)
    g.set_omega(om)
    sd = O.sqrtdiag(A, om, True)
    e, z = np.eye(n), np.zeros(n)

    def run(bv, yv, t):
        g.set_sweep_type(t)
        yd = dev(yv)
        g.apply(dev(bv), yd)
        return host(yd)

    if sweep == O.SOR_FORWARD:
        G = np.stack([run(z, e[i], sweep) for i in range(n)], 1)
        N = np.stack([run(e[i] * sd, z, sweep) for i in range(n)], 1)
    else:
        sym = lambda y, x1, x2: run(x2 * sd, run(x1 * sd, y, O.SOR_FORWARD), O.SOR_BACKWARD)
        G = np.stack([sym(e[i], z, z) for i in range(n)], 1)
        N = np.concatenate([np.stack([sym(z, e[i], z) for i in range(n)], 1), np.stack([sym(z, z, e[i]) for i in range(n)], 1)], 1)
    S = O.stationary_covariance(G, N)
    Q = np.linalg.inv(A.dense())
    assert np.linalg.norm(S - Q) / np.linalg.norm(Q) < 1e-10


def test_ex1_sample_mean_on_device():
    """reference examples/ex1.c:20,83-135 (mcgibbs, 9x9, kappa 10, b = 1) with the HIP sampler: 4e4 samples
    after 400 burn-in, bound 0.02*sqrt(1e6/4e4) = 0.1 (see tests/test_oracle_reference_kat.py)."""
    import torch

    from parmgmc_amd import GridMCSOR

    g = GridMCSOR(9, 9, 1, 10.0)
    b = dev(np.ones(81))
    y = dev(np.zeros(81))
    ctr = g.sample(b, y, 400, seed=0xCAFE)
    mean = torch.zeros_like(y)
    for it in range(40000):
        ctr = g.sample(b, y, 1, seed=0xCAFE, counter0=ctr)
        mean.mul_(it / (it + 1.0)).add_(y, alpha=1.0 / (it + 1))  # SampleCallback of ex1.c:57-64
    ex = np.linalg.solve(O.shifted_laplace(9, 9, 1, 10.0).dense(), np.ones(81))
    assert np.linalg.norm(host(mean) - ex) / np.linalg.norm(ex) < 0.1


def test_slab_decomposition_is_bitwise_identical():
    """Domain decomposition in z with ghost planes (the multi-GPU layout, one process here): sweeping the slabs
    colour by colour with a ghost-plane copy before each colour (MCSORApply_MPIAIJ's per-colour scatter,
    src/mc_sor.c:317-340) gives bit-identical samples to the single-domain sweep, noise included."""
    import torch

    from parmgmc_amd import GridMCSOR
    from parmgmc_amd.slab import SlabSet

    nx, ny, nz, kappa = 10, 6, 9, 3.0
    rng = np.random.default_rng(11)
    b, y0 = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    one = GridMCSOR(nx, ny, nz, kappa)
    one.set_omega(1.1)
    yd = dev(y0)
    one.sample(dev(b), yd, 4, seed=99, counter0=0)
    for cuts in ([0, 4, 9], [0, 2, 5, 9], [0, 1, 2, 9]):
        ss = SlabSet(nx, ny, nz, kappa, cuts)
        ss.set_omega(1.1)
        got = ss.sample_natural(b, y0, 4, seed=99, counter0=0)
        assert np.array_equal(got, host(yd)), cuts


def test_plane_range_sweeps_compose_to_the_full_sweep():
    """boundary planes first, interior afterwards (the multi-GPU overlap order) == one full colour pass, bitwise."""
    from parmgmc_amd import GridMCSOR

    nx, ny, nz, kappa = 20, 18, 9, 1.5
    rng = np.random.default_rng(21)
    b, y0 = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    g = GridMCSOR(nx, ny, nz, kappa)
    g.set_omega(1.1)
    bc = g.to_cvec(dev(b))
    y1, y2 = g.to_cvec(dev(y0)), g.to_cvec(dev(y0))
    for c in (0, 1):
        g.sweep_color_cvec(c, bc, y1, True, True, 5, 3)
        for k0, nk in ((0, 1), (nz - 1, 1), (1, nz - 2)):
            g.sweep_color_planes_cvec(c, k0, nk, bc, y2, True, True, 5, 3)
    assert np.array_equal(host(y1), host(y2))
    from parmgmc_amd import PMGError

    with pytest.raises(PMGError) as e:
        g.sweep_color_planes_cvec(0, 5, 9, bc, y2)
    assert e.value.code == 63


@pytest.mark.parametrize("ny", list(range(61, 73)) + [129, 257])
def test_xcd_bands_of_any_remainder_sweep_every_line_once(ny):
    """round 4: with >= 61 lines the single-device sweep deals the lines to eight XCD bands of floor(ny/8) (+1 for the last
    ny mod 8 bands) and lets the wavefronts of a band walk its (plane, line) pairs without gaps (kernels_grid.hip, `zmain`).
    Every remainder 0..7, an x tail (nx = 2^k+1 at ny = 129) and a plane range: deterministic sweeps bit for bit against the
    reference loop, the noisy chain against the oracle, and boundary planes + interior == the full pass."""
    from parmgmc_amd import GridMCSOR

    nx, nz, kappa = (257 if ny == 129 else 12), 5, 2.0
    A = O.shifted_laplace(nx, ny, nz, kappa)
    col = O.coloring_redblack(nx, ny, nz)
    rng = np.random.default_rng(ny)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    g = GridMCSOR(nx, ny, nz, kappa)
    for om in (1.0, 1.3):
        g.set_omega(om)
        for t in SWEEPS.values():
            g.set_sweep_type(t)
            yd = dev(y)
            g.apply(dev(b), yd)
            assert np.array_equal(host(yd), O.mcsor_apply(A, col, b, y, om, t)), (ny, om, t)
    g.set_omega(1.0)
    g.set_sweep_type(O.SOR_FORWARD)
    yd = dev(y)
    g.sample(dev(b), yd, 2, seed=5, counter0=0)
    want = O.gibbs_samples(A, col, b, y, 2, lambda d: O.noise_grid(nx, ny, nz, 5, d), 1.0, O.SOR_FORWARD, True)
    assert np.abs(host(yd) - want).max() < 1e-13 * np.abs(want).max(), ny
    bc = g.to_cvec(dev(b))
    y1, y2 = g.to_cvec(dev(y)), g.to_cvec(dev(y))
    for c in (0, 1):
        g.sweep_color_cvec(c, bc, y1, True, True, 7, 1)
        for k0, nk in ((0, 1), (nz - 1, 1), (1, nz - 2)):
            g.sweep_color_planes_cvec(c, k0, nk, bc, y2, True, True, 7, 1)
    assert np.array_equal(host(y1), host(y2))


@pytest.mark.parametrize("grid", [(6, 5, 4, 2.0), (70, 3, 2, 1.0), (64, 6, 5, 10.0)] + TAIL_GRIDS)
def test_residual_of_both_colours_matches_csr_product(grid):
    """r = b - A y (PCMG's residual, src/pc_gamgmc.c:253-254) from the one-launch kernel that handles both colours:
    row sums in CSR storage order, so bit-identical to the sequential CSR product."""
    from parmgmc_amd import GridMCSOR

    nx, ny, nz, kappa = grid
    A = O.shifted_laplace(nx, ny, nz, kappa)
    rng = np.random.default_rng(nx + ny)
    b, y = rng.standard_normal(A.n), rng.standard_normal(A.n)
    g = GridMCSOR(nx, ny, nz, kappa)
    r = g.new_cvec()
    g.residual_cvec(g.to_cvec(dev(b)), g.to_cvec(dev(y)), r)
    want = np.empty(A.n)
    for i in range(A.n):  # sequential row sums, storage order
        s = 0.0
        for q in range(A.rowptr[i], A.rowptr[i + 1]):
            s = s + A.vals[q] * y[A.colidx[q]]
        want[i] = b[i] - s
    assert np.array_equal(host(g.from_cvec(r)), want), np.abs(host(g.from_cvec(r)) - want).max()
