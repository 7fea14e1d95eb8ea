"""PCPARSOR's multi-rank sweep (reference src/pc_parsor.c:703-878): the product derives a data-flow graph from the
schedule and level-schedules it on one device (parmgmc_amd/csrc/pmg_parsor.c); the oracle (oracle/parsor.py) emulates
the ranks literally -- ghost vectors, scatters at the points of the schedule, MID messages.  Parity unpinned against
the reference binary (no PETSc here, no fixture for this path in the reference): the emulation is pinned by the
properties below, the product by the emulation, bit for bit."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
from oracle import parsor as PS


def cases():
    rng = np.random.default_rng(42)
    out = []
    for (nx, ny, nz, kappa) in [(7, 6, 1, 1.0), (9, 9, 1, 10.0), (5, 4, 3, 2.0), (12, 5, 2, 0.5)]:
        A = O.shifted_laplace(nx, ny, nz, kappa)
        for nparts in (2, 3, 5, 8):
            cuts = np.sort(rng.choice(np.arange(1, A.n), size=nparts - 1, replace=False))
            out.append((f"{nx}x{ny}x{nz}-{nparts}", A, [0] + [int(c) for c in cuts] + [A.n]))
    # 9-point pattern (a Galerkin coarse operator): more MID rows and MID-MID edges between ranks
    A9 = O.CSR.from_scipy(O.galerkin(O.shifted_laplace(13, 11, 1, 1.0).scipy(), O.q1_interp(7, 6, 1)))
    for nparts in (3, 6):
        cuts = np.sort(rng.choice(np.arange(1, A9.n), size=nparts - 1, replace=False))
        out.append((f"galerkin7x6-{nparts}", A9, [0] + [int(c) for c in cuts] + [A9.n]))
    # random symmetric patterns: all combinations, also a BOT row next to a higher rank's MID row whose message does
    # (another MID row of the same rank references it) or does not reach the BOT row's ghost slot
    for n, dens, seed, nparts in [(60, 0.05, 2, 6), (50, 0.06, 3, 4), (80, 0.04, 4, 8)]:
        r2 = np.random.default_rng(seed)
        M = sp.random(n, n, density=dens, random_state=r2, format="csr")
        M = (M + M.T).tolil()
        M.setdiag(0)
        M = M.tocsr()
        M.eliminate_zeros()
        M.data[:] = -np.abs(M.data)
        A = O.CSR.from_scipy(M + sp.diags(np.asarray(-M.sum(axis=1)).ravel() + 1.0))
        cuts = np.sort(r2.choice(np.arange(1, n), size=nparts - 1, replace=False))
        out.append((f"random{n}-{nparts}", A, [0] + [int(c) for c in cuts] + [n]))
    return out


CASES = cases()


def lexicographic(A, b, x, omega, its):
    """SORLocalForwardSweepIS over all rows (the C restatement, oracle/pmg_oracle.c) with PCPARSOR's idiag = omega / d
    (LocalMatInvertDiagonalForSOR, src/pc_parsor.c:69-81; MCSOR's is (1/d) * omega, src/mc_sor.c:114-124)"""
    x = np.array(x, copy=True)
    dp = O.diag_pointers(A)
    idg = (1.0 / A.vals[dp]) if omega == 1.0 else omega / A.vals[dp]
    rows = np.arange(A.n, dtype=np.int32)
    for _ in range(its):
        O.lib().orc_parsor_rows(A.n, rows, A.rowptr, A.colidx, A.vals, dp, np.ascontiguousarray(idg), omega, np.ascontiguousarray(b), x, None, None, None, None)
    return x


def test_emulation_properties():
    """one rank == the lexicographic sweep of MatSOR / MCSOR with one colour; A^-1 b is a fixed point for any partition;
    the iteration converges; every class of row occurs somewhere in the cases"""
    seen = np.zeros(4, int)
    for name, A, parts in CASES:
        rng = np.random.default_rng(len(parts))
        b, x = rng.standard_normal(A.n), rng.standard_normal(A.n)
        one = PS.parsor_apply(A, [0, A.n], b, x, 1.0, 2)
        ref = x.copy()
        for _ in range(2):
            ref = O.mcsor_apply(A, np.zeros(A.n, np.int32), b, ref, 1.0, O.SOR_FORWARD)
        assert np.array_equal(one, ref), name
        assert np.array_equal(PS.parsor_apply(A, [0, A.n], b, x, 1.2, 2), lexicographic(A, b, x, 1.2, 2)), name
        xs = np.linalg.solve(A.dense(), b)
        assert np.abs(PS.parsor_apply(A, parts, b, xs, 1.0, 1) - xs).max() < 1e-13, name
        assert np.abs(PS.parsor_apply(A, parts, b, x, 1.0, 300) - xs).max() < 1e-10, name
        seen += np.bincount(PS.node_classes(A, parts), minlength=4)
    assert (seen > 0).all(), seen


def level_sweep_on_host(n, rp, ci, va, colors, nlevels, b, x, omega, its, zero):
    """what the device does with the 2n-row operator: snapshot, then the levels in ascending order (rows of a level in
    any order -- they are independent); row sums in storage order"""
    x = np.zeros(n) if zero else np.array(x, copy=True)
    for _ in range(its):
        xe = np.concatenate([x, x])
        for lev in range(nlevels):
            rows = [i for i in range(n) if colors[i] == lev]
            new = {}
            for i in rows:
                s, d = b[i], None
                for k in range(rp[i], rp[i + 1]):
                    if ci[k] == i:
                        d = va[k]
                    else:
                        s = s - va[k] * xe[ci[k]]
                new[i] = (1.0 - omega) * xe[i] + s * (omega / d)
            for i, v in new.items():
                xe[i] = v
        x = xe[:n].copy()
    return x


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_dataflow_builder_reproduces_the_emulation_on_the_host(case):
    """host logic only (no GPU): the level schedule + snapshot redirection built by pmg_parsor_build_dataflow, executed in
    numpy, equals the literal emulation bit for bit"""
    from parmgmc_amd import capi

    name, A, parts = case
    n, nparts = A.n, len(parts) - 1
    fn = capi.lib.pmg_parsor_build_dataflow
    fn.restype = C.c_int
    P32, PD = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    rp, ci, va, co = P32(), P32(), PD(), P32()
    nl = C.c_int32()
    rs = np.asarray(parts, np.int32)
    pcs, cls = np.zeros(nparts, np.int32), np.zeros(n, np.int32)
    st = fn(C.c_int32(n), A.rowptr.ctypes.data_as(P32), A.colidx.ctypes.data_as(P32), A.vals.ctypes.data_as(PD), C.c_int32(nparts), rs.ctypes.data_as(P32), None, C.byref(rp), C.byref(ci), C.byref(va), C.byref(co), C.byref(nl), pcs.ctypes.data_as(P32), cls.ctypes.data_as(P32))
    assert st == 0
    assert np.array_equal(pcs, PS.color_processors(A, parts))
    assert np.array_equal(cls, PS.node_classes(A, parts))
    rpn = np.ctypeslib.as_array(rp, (2 * n + 1,)).copy()
    nnz = int(rpn[-1])
    cin, van, con = np.ctypeslib.as_array(ci, (nnz,)).copy(), np.ctypeslib.as_array(va, (nnz,)).copy(), np.ctypeslib.as_array(co, (2 * n,)).copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for p in (rp, ci, va, co):
        libc.free(C.cast(p, C.c_void_p))
    assert nnz == A.rowptr[-1] + n and (con[n:] == nl.value).all() and con[:n].max() == nl.value - 1
    rng = np.random.default_rng(7)
    b, x = rng.standard_normal(n), rng.standard_normal(n)
    for omega, its, zero in [(1.0, 1, False), (1.3, 2, False), (0.9, 2, True)]:
        want = PS.parsor_apply(A, parts, b, x, omega, its, zero)
        got = level_sweep_on_host(n, rpn, cin, van, con, nl.value, b, x, omega, its, zero)
        assert np.array_equal(got, want), (name, omega, its, zero, np.abs(got - want).max())


def test_rejects_bad_partitions():
    from parmgmc_amd import capi

    A = O.shifted_laplace(5, 4, 1, 1.0)
    fn = capi.lib.pmg_parsor_build_dataflow
    fn.restype = C.c_int
    P32, PD = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    rp, ci, va, co, nl = P32(), P32(), PD(), P32(), C.c_int32()

    def call(parts, cols=None):
        rs = np.asarray(parts, np.int32)
        pc = None if cols is None else np.asarray(cols, np.int32).ctypes.data_as(P32)
        return fn(C.c_int32(A.n), A.rowptr.ctypes.data_as(P32), A.colidx.ctypes.data_as(P32), A.vals.ctypes.data_as(PD), C.c_int32(len(parts) - 1), rs.ctypes.data_as(P32), pc, C.byref(rp), C.byref(ci), C.byref(va), C.byref(co), C.byref(nl), None, None)

    assert call([0, 10, 10, 20]) != 0  # a rank without rows
    assert call([0, 10, 19]) != 0  # does not cover the matrix
    assert call([0, 10, 20], cols=[1, 1]) != 0  # adjacent ranks with one colour


GPU_CASES = CASES[::3] + CASES[-2:]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_parsor_pc_with_partition_matches_the_emulation_on_device(case):
    import torch

    import parmgmc_amd.pc as P

    name, A, parts = case
    P.initialize()
    try:
        mat = P.Mat.csr(A.rowptr, A.colidx, A.vals)
        pc = P.PC("parsor")
        pc.set_operators(mat)
        pc.parsor_set_partition(parts)
        rng = np.random.default_rng(3)
        b, x = rng.standard_normal(A.n), rng.standard_normal(A.n)
        for omega, its, zero in [(1.0, 1, False), (1.3, 3, False), (1.0, 2, True)]:
            pc.parsor_set_omega(omega)
            nl, pcs, cls = pc.parsor_partition_info()
            assert np.array_equal(pcs, PS.color_processors(A, parts)) and np.array_equal(cls, PS.node_classes(A, parts)) and nl >= 1
            xd = torch.as_tensor(x, device="cuda").clone()
            pc.parsor_apply_sor(torch.as_tensor(b, device="cuda"), its, zero, xd)
            want = PS.parsor_apply(A, parts, b, x, omega, its, zero)
            assert np.array_equal(xd.cpu().numpy(), want), (name, omega, its, zero, np.abs(xd.cpu().numpy() - want).max())
        # PCApply = its sweeps from a zero guess (src/pc_parsor.c:880-890); no partition = the lexicographic order again
        pc.parsor_set_partition([0, A.n])
        pc.parsor_set_omega(1.0)
        yd = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        pc.apply(torch.as_tensor(b, device="cuda"), yd)
        assert np.array_equal(yd.cpu().numpy(), lexicographic(A, b, np.zeros(A.n), 1.0, 1))
        pc.parsor_set_partition([])  # back to the plain single-rank PC (dependency levels of the natural order)
        pc.parsor_set_omega(1.3)
        xd = torch.as_tensor(x, device="cuda").clone()
        pc.parsor_apply_sor(torch.as_tensor(b, device="cuda"), 2, False, xd)
        assert np.array_equal(xd.cpu().numpy(), lexicographic(A, b, x, 1.3, 2))
    finally:
        P.finalize()
