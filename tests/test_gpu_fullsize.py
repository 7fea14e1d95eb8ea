"""Parity at BASELINE.json's full sizes (256^3 and 512^3 DMDA) through size-independent properties: the oracle
cannot run there in seconds, so the HIP path is checked against identities the reference's own tests and the
algebra of the sweep provide."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[256, 512])
def setup(request):
    import torch

    from parmgmc_amd import GridMCSOR

    n = request.param
    g = GridMCSOR(n, n, n, 10.0)
    gen = torch.Generator(device="cuda").manual_seed(n)
    b = torch.randn(g.n, dtype=torch.float64, device="cuda", generator=gen)
    y = torch.randn(g.n, dtype=torch.float64, device="cuda", generator=gen)
    return n, g, g.to_cvec(b), g.to_cvec(y), b, y


def test_cvec_roundtrip(setup):
    import torch

    n, g, bc, yc, b, y = setup
    assert torch.equal(g.from_cvec(yc), y)
    assert int(torch.count_nonzero(yc)) == g.n  # pads and ghost planes are zero, every value stored once


def test_ex5_symmetric_equals_forward_then_backward(setup):
    """reference examples/ex5.c:53-70 at full size, bitwise."""
    import torch

    n, g, bc, yc, b, y = setup
    g.set_omega(1.0)
    x1, x2 = yc.clone(), yc.clone()
    g.set_sweep_type(1)
    g.apply_cvec(bc, x1)
    g.set_sweep_type(2)
    g.apply_cvec(bc, x1)
    g.set_sweep_type(3)
    g.apply_cvec(bc, x2)
    g.set_sweep_type(1)
    assert torch.equal(x1, x2)


def test_exact_solution_is_a_fixed_point(setup):
    """If b = A y*, a deterministic sweep leaves y* unchanged (to rounding): ties the sweep kernel to the residual
    kernel (two independent implementations of the operator) at full size, for omega = 1 and omega != 1."""
    import torch

    n, g, bc, yc, b, y = setup
    zero = g.new_cvec()
    r = g.new_cvec()
    g.residual_cvec(zero, yc, r)  # r = 0 - A y*
    bstar = -r
    for om in (1.0, 1.3):
        g.set_omega(om)
        x = yc.clone()
        g.apply_cvec(bstar, x)
        assert float((x - yc).abs().max()) < 1e-12 * float(yc.abs().max())
    g.set_omega(1.0)


def test_sweep_is_affine(setup):
    """sweep(b1 + b2, y1 + y2) - sweep(b1, y1) - sweep(b2, y2) + sweep(0, 0) = 0 (the deterministic sweep is linear)."""
    import torch

    n, g, bc, yc, b, y = setup
    s12 = (yc * 0.5 + 0.25 * bc)
    g.apply_cvec(bc * 0.5 + yc * 0.125, s12)
    s1 = yc * 0.5
    g.apply_cvec(bc * 0.5, s1)
    s2 = 0.25 * bc
    g.apply_cvec(yc * 0.125, s2)
    assert float((s12 - s1 - s2).abs().max()) < 1e-13 * float(s12.abs().max())


def test_noise_is_standard_normal_and_scaled_by_sqrt_diag(setup):
    """One noisy sweep from y = 0 with b = 0 and the other colour still zero gives, on the FIRST colour,
    y = idiag * sqrtdiag * xi = xi / sqrt(d): check mean, variance, 4th moment over all interior points and that a
    second call with the same counter reproduces the sample bit for bit while another counter does not."""
    import torch

    n, g, bc, yc, b, y = setup
    zero = g.new_cvec()
    y1 = g.new_cvec()
    g.sweep_color_cvec(0, zero, y1, True, True, 1234, 77)
    nat = g.from_cvec(y1).view(n, n, n)[1:-1, 1:-1, 1:-1]
    idx = torch.arange(1, n - 1, device="cuda")
    red = ((idx[:, None, None] + idx[None, :, None] + idx[None, None, :]) & 1) == 0
    d = 100.0 + 6.0 / ((n - 1) * (n - 1))
    xi = nat[red] * np.sqrt(d)
    m = xi.numel()
    assert abs(float(xi.mean())) < 5 / np.sqrt(m)
    assert abs(float(xi.var()) - 1) < 5 * np.sqrt(2.0 / m)
    assert abs(float((xi ** 4).mean()) - 3) < 5 * np.sqrt(96.0 / m)
    assert float(nat[~red].abs().max()) == 0.0
    y2 = g.new_cvec()
    g.sweep_color_cvec(0, zero, y2, True, True, 1234, 77)
    assert torch.equal(y1, y2)
    g.sweep_color_cvec(0, zero, y2, True, True, 1234, 78)
    assert not torch.equal(y1, y2)


def test_slab_decomposition_at_full_size(setup):
    """Two slabs with ghost-plane copies == one domain, bitwise, including noise (global-index keyed)."""
    import torch

    from parmgmc_amd import GridMCSOR

    n, g, bc, yc, b, y = setup
    if n > 256:
        pytest.skip("256^3 is enough for the decomposition identity")
    one = yc.clone()
    g.sample_cvec(bc, one, 2, seed=5, counter0=0)
    cut = n // 2 + 3
    plane = n * n
    parts = []
    slabs = [GridMCSOR(n, n, n, 10.0, kz0=0, nz_owned=cut), GridMCSOR(n, n, n, 10.0, kz0=cut, nz_owned=n - cut)]
    bs = [s.to_cvec(b[lo * plane:hi * plane].contiguous()) for s, (lo, hi) in zip(slabs, [(0, cut), (cut, n)])]
    ys = [s.to_cvec(y[lo * plane:hi * plane].contiguous()) for s, (lo, hi) in zip(slabs, [(0, cut), (cut, n)])]
    for it in range(2):
        for c in (0, 1):
            oh, gh, cnt = slabs[0].halo_plane(1 - c, 1)
            ol, gl, _ = slabs[1].halo_plane(1 - c, 0)
            ys[0][gh:gh + cnt].copy_(ys[1][ol:ol + cnt])
            ys[1][gl:gl + cnt].copy_(ys[0][oh:oh + cnt])
            for s, bb, yy in zip(slabs, bs, ys):
                s.sweep_color_cvec(c, bb, yy, True, True, 5, it)
    got = torch.cat([s.from_cvec(yy) for s, yy in zip(slabs, ys)])
    assert torch.equal(got, g.from_cvec(one))


@pytest.mark.parametrize("n,levels,kappa", [(257, 5, 10.0), (257, 5, 0.5), (513, 6, 10.0)])
def test_vcycle_properties_at_full_size(n, levels, kappa):
    """BASELINE configs 2 / 3 (257^3 and 513^3 hierarchies; the oracle cannot run there): with the noise held fixed
    (same seed and counters) a sample is an affine map of (b, y0), so
      (i)  y(b1 + b2, y0) - y(b1, y0) = y(b2, 0) - y(0, 0)         (linearity in the right-hand side), and
      (ii) y_k(b, y0) - y_k(b, y0') = E^k (y0 - y0') with E the error propagation of the deterministic multigrid
           V-cycle, which must contract (by more than 3x per cycle even at kappa = 0.5, where plain Gibbs stalls);
      (iii) the packed / unpacked lane mappings and the proxy-table set-up were compared bit for bit on small grids
           -- here the result must at least be finite and deterministic (two runs agree bit for bit)."""
    import torch

    from parmgmc_amd import MGMC

    mg = MGMC(n, n, n, kappa, levels).setup()
    N = n ** 3
    gen = torch.Generator(device="cuda").manual_seed(7)
    b1 = torch.randn(N, dtype=torch.float64, device="cuda", generator=gen)
    b2 = torch.randn(N, dtype=torch.float64, device="cuda", generator=gen)
    y0 = torch.randn(N, dtype=torch.float64, device="cuda", generator=gen)
    zero = torch.zeros(N, dtype=torch.float64, device="cuda")

    def run(b, start, its=1):
        y = start.clone()
        mg.sample(b, y, its, seed=11, counter0=5)
        return y

    a, c, d, e = run(b1 + b2, y0), run(b1, y0), run(b2, zero), run(zero, zero)
    lhs, rhs = a - c, d - e
    assert bool(torch.isfinite(a).all())
    assert float((lhs - rhs).abs().max()) < 1e-11 * float(rhs.abs().max())
    assert torch.equal(run(b1, y0), c)  # deterministic
    # contraction of the difference of two chains driven by the same noise
    prev = float((y0 - zero).norm())
    for k in (1, 2, 3):
        dk = float((run(b1, y0, k) - run(b1, zero, k)).norm())
        assert dk < prev / 3.0 or dk == 0.0, (k, dk, prev)  # kappa = 10: the two chains coincide bit for bit after two cycles
        prev = dk
