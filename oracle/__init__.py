"""CPU oracle for the ParMGMC Gibbs/SOR hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package.  Nothing under ``parmgmc_amd/`` imports it; the product path is the HIP library and fails
loudly when that library is missing.

The scalar loops live in ``pmg_oracle.c`` (plain C, each function cites the reference file:line it
restates); this module is the ctypes/numpy glue plus the pieces whose arithmetic the reference
delegates to PETSc (Q1 interpolation, Galerkin products, the PCMG V-cycle), restated here from PETSc's
documented semantics with scipy.sparse.  ``/root/reference`` is never read at run time.

Parity status: the reference cannot be built here (needs PETSc), so the oracle is pinned by the
reference's own known-answer tests -- ``examples/ex5.c`` (symmetric == forward then backward, <1e-15),
``examples/ex1.c`` (sample mean -> A^-1 b), ``examples/ex6.c`` + ``src/stats.c`` (covariance metric) --
see ``tests/test_oracle_*.py``.  What PETSc computes internally (JP colouring, GAMG aggregates,
PetscRandom/MKL streams, the entries of DMDA interpolation) is "parity unpinned" by any reference
fixture; those pieces follow PETSc's documentation and our own stated rules.  ``parsor.py`` is the literal
multi-rank emulation of PCPARSOR's schedule (src/pc_parsor.c:703-878), pinned by properties only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
import scipy.sparse as sp

_HERE = Path(__file__).resolve().parent

SOR_FORWARD, SOR_BACKWARD, SOR_SYMMETRIC = 1, 2, 3


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    so = _HERE / "libpmg_oracle.so"
    src = _HERE / "pmg_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "all"], stdout=subprocess.DEVNULL)


def _load(native: bool = False) -> C.CDLL:
    build()
    return C.CDLL(str(_HERE / ("libpmg_oracle_native.so" if native else "libpmg_oracle.so")))


_lib = None
_lib_native = None


def lib(native: bool = False) -> C.CDLL:
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            _lib_native = _load(True)
            _declare(_lib_native)
        return _lib_native
    if _lib is None:
        _lib = _load(False)
        _declare(_lib)
    return _lib


_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def _declare(L: C.CDLL) -> None:
    L.orc_laplace_nnz.restype = C.c_int64
    L.orc_laplace_nnz.argtypes = [C.c_int] * 3
    L.orc_assemble_shifted_laplace.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _i32p, _i32p, _f64p]
    L.orc_assemble_ex6.argtypes = [C.c_int, C.c_double, _i32p, _i32p, _f64p]
    L.orc_diag_pointers.argtypes = [C.c_int, _i32p, _i32p, _i32p]
    L.orc_idiag.argtypes = [C.c_int, _i32p, _f64p, C.c_double, _f64p]
    L.orc_sqrtdiag.argtypes = [C.c_int, _i32p, _f64p, C.c_double, C.c_int, _f64p]
    L.orc_mcsor_sweep_seq.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p, _i32p, _f64p, C.c_double, _f64p, _f64p]
    L.orc_mcsor_apply.argtypes = L.orc_mcsor_sweep_seq.argtypes
    L.orc_mcsor_rank_color.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, _f64p, _f64p]
    L.orc_parsor_rows.argtypes = [C.c_int, _i32p, _i32p, _i32p, _f64p, _i32p, _f64p, C.c_double, _f64p, _f64p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.orc_normal_pair.argtypes = [_u32p, _u32p, _f64p]
    L.orc_vec_set_random_standard_normal.argtypes = [C.c_int, _f64p, _f64p]
    L.orc_noise_rows.argtypes = [C.c_int64, C.c_uint64, C.c_uint64, _f64p]
    L.orc_noise_grid.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _f64p]
    L.orc_prepare_rhs.argtypes = [C.c_int64, _f64p, _f64p, C.c_void_p, _f64p]
    L.orc_gibbs_sample_serial.argtypes = [C.c_int, _i32p, _i32p, _f64p, _i32p, _f64p, _f64p, C.c_double, _f64p, _f64p, _f64p, C.c_uint64, C.c_uint64]
    L.orc_gibbs_sample_colored_parallel.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p, _i32p, _f64p, _f64p, C.c_double, _f64p, _f64p, _f64p, C.c_uint64, C.c_uint64]
    L.orc_gibbs_sample_colored_parallel.restype = None
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_potrf_lower.restype = C.c_int
    L.orc_potrf_lower.argtypes = [C.c_int, _f64p]
    L.orc_chol_sample.argtypes = [C.c_int, _f64p, _f64p, _f64p, _f64p]
    L.orc_spmv.argtypes = [C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p]
    _i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    L.orc_grid7_rows_sweep.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, _i64p, _f64p, _f64p, _f64p, _f64p]
    L.orc_grid7_rows_residual.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int64, _i64p, _f64p, _f64p, _f64p]
    L.orc_st27_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _f64p, _f64p, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, _i64p, _f64p, _f64p, _f64p, _f64p]
    L.orc_q1_rows.argtypes = [C.c_int, _i32p, _i32p, C.c_int64, _i64p, _f64p, _f64p, _f64p]
    for f in (L.orc_grid7_rows_sweep, L.orc_grid7_rows_residual, L.orc_st27_rows, L.orc_q1_rows):
        f.restype = None


# ------------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------------
class CSR:
    """Borrowed-array CSR triple (rowptr int32[n+1], colidx int32[nnz], vals f64[nnz]), square."""

    def __init__(self, rowptr, colidx, vals, shape=None):
        self.rowptr = np.ascontiguousarray(rowptr, np.int32)
        self.colidx = np.ascontiguousarray(colidx, np.int32)
        self.vals = np.ascontiguousarray(vals, np.float64)
        self.n = len(self.rowptr) - 1
        self.shape = shape or (self.n, self.n)

    def scipy(self) -> sp.csr_matrix:
        return sp.csr_matrix((self.vals, self.colidx, self.rowptr), shape=self.shape)

    @staticmethod
    def from_scipy(m) -> "CSR":
        m = sp.csr_matrix(m)
        m.sort_indices()
        return CSR(m.indptr, m.indices, m.data, m.shape)

    def dense(self) -> np.ndarray:
        return self.scipy().toarray()


def shifted_laplace(nx: int, ny: int, nz: int = 1, kappa: float = 1.0) -> CSR:
    """reference src/problems.c:14-75 (2-D when nz == 1) and its 3-D analogue."""
    L = lib()
    n = nx * ny * nz
    nnz = L.orc_laplace_nnz(nx, ny, nz)
    rp, ci, v = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    L.orc_assemble_shifted_laplace(nx, ny, nz, kappa, rp, ci, v)
    return CSR(rp, ci, v)


def ex6_matrix(n: int, kappa: float) -> CSR:
    """reference examples/ex6.c:69-127."""
    L = lib()
    nnz = L.orc_laplace_nnz(n, n, 1)
    rp, ci, v = np.zeros(n * n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    L.orc_assemble_ex6(n, kappa, rp, ci, v)
    return CSR(rp, ci, v)


def diag_pointers(A: CSR) -> np.ndarray:
    d = np.full(A.n, -1, np.int32)
    lib().orc_diag_pointers(A.n, A.rowptr, A.colidx, d)
    return d


def idiag(A: CSR, omega: float) -> np.ndarray:
    out = np.zeros(A.n)
    lib().orc_idiag(A.n, diag_pointers(A), A.vals, omega, out)
    return out


def sqrtdiag(A: CSR, omega: float = 1.0, scale: bool = True) -> np.ndarray:
    out = np.zeros(A.n)
    lib().orc_sqrtdiag(A.n, diag_pointers(A), A.vals, omega, int(scale), out)
    return out


# ------------------------------------------------------------------------------------------------
# colourings.  The reference's serial colouring is "every row colour 0" (src/mc_sor.c:397-410); its
# parallel one is PETSc's randomised JP (src/mc_sor.c:383-395, third party, parity unpinned).  The
# deterministic rules below are the build's own and are what "bit-exact colouring" is tested against.
# ------------------------------------------------------------------------------------------------
def coloring_single(n: int) -> np.ndarray:
    """MatCreateISColoring_Seq, reference src/mc_sor.c:397-410."""
    return np.zeros(n, np.int32)


def coloring_redblack(nx: int, ny: int, nz: int = 1) -> np.ndarray:
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    c = ((i + j + k) & 1).astype(np.int32)
    return np.ascontiguousarray(c.transpose(2, 1, 0)).ravel()  # natural order i fastest


def coloring_parity8(nx: int, ny: int, nz: int = 1) -> np.ndarray:
    """(i&1) + 2(j&1) + 4(k&1): valid for any stencil contained in the 27-point box."""
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    c = ((i & 1) + 2 * (j & 1) + 4 * (k & 1)).astype(np.int32)
    c = np.ascontiguousarray(c.transpose(2, 1, 0)).ravel()
    _, inv = np.unique(c, return_inverse=True)  # compress (2-D grids use 4 colours)
    return inv.astype(np.int32)


def coloring_greedy(A: CSR) -> np.ndarray:
    """First-fit in natural row order over the structural neighbours (distance 1)."""
    col = np.full(A.n, -1, np.int32)
    for r in range(A.n):
        used = set()
        for k in range(A.rowptr[r], A.rowptr[r + 1]):
            c = A.colidx[k]
            if c != r and col[c] >= 0:
                used.add(int(col[c]))
        c = 0
        while c in used:
            c += 1
        col[r] = c
    return col


def coloring_iterated(A: CSR) -> np.ndarray:
    """First-fit, then first-fit once more with the rows visited class by class, the LAST class first, rows ascending inside
    a class (one round of iterated greedy; the library's PMG_COLORING_ITERATED).  Never more classes than first-fit."""
    col0 = coloring_greedy(A)
    nc0 = int(col0.max()) + 1 if A.n else 0
    if nc0 <= 2:
        return col0
    order = np.lexsort((np.arange(A.n), nc0 - 1 - col0))
    col = np.full(A.n, -1, np.int32)
    for r in order:
        used = set()
        for k in range(A.rowptr[r], A.rowptr[r + 1]):
            c = A.colidx[k]
            if c != r and col[c] >= 0:
                used.add(int(col[c]))
        c = 0
        while c in used:
            c += 1
        col[r] = c
    return col


def coloring_lexlevels(A: CSR) -> np.ndarray:
    """Dependency levels of the lexicographic sweep: level(r) = 1 + max level(c), c<r a neighbour.
    A multicolour sweep over these classes in ascending order reproduces plain Gauss-Seidel (the
    reference's serial path and PETSc MatSOR forward sweep) update for update."""
    lev = np.zeros(A.n, np.int32)
    for r in range(A.n):
        m = -1
        for k in range(A.rowptr[r], A.rowptr[r + 1]):
            c = A.colidx[k]
            if c < r and lev[c] > m:
                m = lev[c]
        lev[r] = m + 1
    return lev


def color_lists(colors: np.ndarray):
    """ISColoringGetIS: (ncolors, colorptr, colorrows) with rows ascending inside a colour."""
    colors = np.asarray(colors, np.int32)
    nc = int(colors.max()) + 1 if len(colors) else 0
    order = np.argsort(colors, kind="stable").astype(np.int32)
    ptr = np.zeros(nc + 1, np.int32)
    np.cumsum(np.bincount(colors, minlength=nc), out=ptr[1:])
    return nc, ptr, order


def coloring_is_valid(A: CSR, colors: np.ndarray) -> bool:
    rows = np.repeat(np.arange(A.n), np.diff(A.rowptr))
    off = rows != A.colidx
    return not np.any(colors[rows[off]] == colors[A.colidx[off]])


# ------------------------------------------------------------------------------------------------
# sweeps
# ------------------------------------------------------------------------------------------------
def mcsor_apply(A: CSR, colors: np.ndarray, b: np.ndarray, y: np.ndarray, omega: float = 1.0, sweep: int = SOR_FORWARD) -> np.ndarray:
    """MCSORApply (reference src/mc_sor.c:216-296) on a copy of y; returns the new y."""
    nc, ptr, rows = color_lists(colors)
    y = np.array(y, np.float64, copy=True)
    b = np.ascontiguousarray(b, np.float64)
    lib().orc_mcsor_apply(sweep, nc, ptr, rows, A.rowptr, A.colidx, A.vals, diag_pointers(A), idiag(A, omega), omega, b, y)
    return y


def split_domain(A: CSR, lo: int, hi: int):
    """MatMPIAIJGetSeqAIJ: diagonal block (local columns), off-diagonal block (compressed ghost
    columns, ascending global order) and colmap, for the rank owning rows [lo, hi)."""
    M = A.scipy()[lo:hi].tocsr()
    M.sort_indices()
    cols = M.indices
    local = (cols >= lo) & (cols < hi)
    rows = np.repeat(np.arange(hi - lo), np.diff(M.indptr))
    ad = sp.csr_matrix((M.data[local], (rows[local], cols[local] - lo)), shape=(hi - lo, hi - lo))
    ghost_cols = np.unique(cols[~local])
    remap = {int(g): q for q, g in enumerate(ghost_cols)}
    ao = sp.csr_matrix((M.data[~local], (rows[~local], np.array([remap[int(c)] for c in cols[~local]], dtype=np.int64))), shape=(hi - lo, max(len(ghost_cols), 1)))
    ad.sort_indices()
    ao.sort_indices()
    return CSR.from_scipy(ad), CSR(ao.indptr, ao.indices, ao.data, ao.shape), ghost_cols.astype(np.int32)


def mcsor_sweep_domains(A: CSR, own: list[int], colors: np.ndarray, b, y, omega=1.0, sweep=SOR_FORWARD) -> np.ndarray:
    """MCSORApply_MPIAIJ + MatCreateScatters (reference src/mc_sor.c:298-381, :152-214) emulated for
    len(own)-1 row-block ranks in one process.  For every colour all ranks first gather their ghost
    buffers from the current global y (the scatter of :318-319 completes everywhere before any rank
    sweeps), then each rank sweeps its rows of that colour."""
    L = lib()
    y = np.array(y, np.float64, copy=True)
    b = np.ascontiguousarray(b, np.float64)
    idg = idiag(A, omega)
    ndom = len(own) - 1
    parts = []
    for d in range(ndom):
        lo, hi = own[d], own[d + 1]
        ad, ao, colmap = split_domain(A, lo, hi)
        nc_d, ptr, rows = color_lists(colors[lo:hi])
        parts.append((ad, ao, colmap, ptr, rows, diag_pointers(ad)))
    nc = int(colors.max()) + 1

    def one(direction):
        crange = range(nc) if direction == SOR_FORWARD else range(nc - 1, -1, -1)
        for c in crange:
            ghosts = []
            for d in range(ndom):
                ad, ao, colmap, ptr, rows, dg = parts[d]
                rc = rows[ptr[c]:ptr[c + 1]] if c + 1 < len(ptr) else rows[:0]
                idx = [colmap[ao.colidx[k]] for r in rc for k in range(ao.rowptr[r], ao.rowptr[r + 1])]
                ghosts.append(np.ascontiguousarray(y[np.array(idx, dtype=np.int64)] if idx else np.zeros(1)))
            for d in range(ndom):
                ad, ao, colmap, ptr, rows, dg = parts[d]
                lo, hi = own[d], own[d + 1]
                rc = np.ascontiguousarray(rows[ptr[c]:ptr[c + 1]] if c + 1 < len(ptr) else rows[:0])
                yl = np.ascontiguousarray(y[lo:hi])
                L.orc_mcsor_rank_color(direction, len(rc), rc if len(rc) else np.zeros(1, np.int32), ad.rowptr, ad.colidx if len(ad.colidx) else np.zeros(1, np.int32), ad.vals if len(ad.vals) else np.zeros(1), dg, ao.rowptr, ao.vals if len(ao.vals) else np.zeros(1), ghosts[d], np.ascontiguousarray(idg[lo:hi]), omega, np.ascontiguousarray(b[lo:hi]), yl)
                y[lo:hi] = yl

    if sweep == SOR_SYMMETRIC:
        one(SOR_FORWARD)
        one(SOR_BACKWARD)
    else:
        one(sweep)
    return y


# ------------------------------------------------------------------------------------------------
# noise
# ------------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key) -> np.ndarray:
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def normal_pair(ctr, key) -> np.ndarray:
    out = np.zeros(2)
    lib().orc_normal_pair(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def box_muller_vec(n: int, uniforms: np.ndarray) -> np.ndarray:
    """VecSetRandomStandardNormal, Box-Muller branch (reference src/parmgmc.c:99-110)."""
    out = np.zeros(n)
    lib().orc_vec_set_random_standard_normal(n, np.ascontiguousarray(uniforms, np.float64), out)
    return out


def noise_rows(n: int, seed: int, sweep: int) -> np.ndarray:
    xi = np.zeros(n)
    lib().orc_noise_rows(n, seed, sweep, xi)
    return xi


def noise_grid(nx: int, ny: int, nz: int, seed: int, sweep: int) -> np.ndarray:
    xi = np.zeros(nx * ny * nz)
    lib().orc_noise_grid(nx, ny, nz, seed, sweep, xi)
    return xi


def prepare_rhs(xi, sqrtd, b) -> np.ndarray:
    """PrepareRHS_Default (reference src/pc_mcgibbs.c:119-128)."""
    w = np.zeros(len(xi))
    bb = None if b is None else np.ascontiguousarray(b, np.float64)
    lib().orc_prepare_rhs(len(xi), np.ascontiguousarray(xi), np.ascontiguousarray(sqrtd), None if bb is None else bb.ctypes.data, w)
    return w


def gibbs_samples(A: CSR, colors, b, y0, its: int, noise_fn, omega=1.0, sweep=SOR_FORWARD, scaled=True, callback=None) -> np.ndarray:
    """PCApplyRichardson_MulticolorGibbs (reference src/pc_mcgibbs.c:155-188; scaled=True) or
    PCApplyRichardson_SORGibbs (src/pc_sorgibbs.c:115-134; scaled=False, omega must be 1).
    noise_fn(draw_index) -> xi supplies the standard normals of each PrepareRHS call; a symmetric
    sweep draws twice per sample (pc_mcgibbs.c:172-181)."""
    sd = sqrtdiag(A, omega, scaled)
    y = np.array(y0, np.float64, copy=True)
    draw = 0
    for it in range(its):
        if sweep in (SOR_FORWARD, SOR_BACKWARD):
            w = prepare_rhs(noise_fn(draw), sd, b)
            draw += 1
            y = mcsor_apply(A, colors, w, y, omega, sweep)
        else:
            for d in (SOR_FORWARD, SOR_BACKWARD):
                w = prepare_rhs(noise_fn(draw), sd, b)
                draw += 1
                y = mcsor_apply(A, colors, w, y, omega, d)
        if callback is not None:
            callback(it, y)
    return y


# ------------------------------------------------------------------------------------------------
# statistics (reference src/stats.c)
# ------------------------------------------------------------------------------------------------
def covariance_error(A: CSR, samples: np.ndarray) -> float:
    """EstimateCovarianceMatErrors for one sample index (reference src/stats.c:63-117): unbiased
    1/(n-1) sample covariance across chains, ||C - A^-1||_F / ||A^-1||_F.  samples: (chains, N)."""
    Q = np.linalg.inv(A.dense())
    m = samples.mean(axis=0)
    D = samples - m
    Cm = D.T @ D / (samples.shape[0] - 1)
    return float(np.linalg.norm(Cm - Q) / np.linalg.norm(Q))


def stationary_covariance(G: np.ndarray, Nn: np.ndarray) -> np.ndarray:
    """Covariance of the stationary law of y <- G y + Nn xi + c, xi ~ N(0,I): the solution of the
    discrete Lyapunov equation S = G S G^T + Nn Nn^T."""
    from scipy.linalg import solve_discrete_lyapunov

    return solve_discrete_lyapunov(G, Nn @ Nn.T)


# ------------------------------------------------------------------------------------------------
# PCMG pieces (PETSc, third party: restated from its documented semantics -- parity unpinned)
# ------------------------------------------------------------------------------------------------
def q1_interp_1d(nc: int) -> sp.csr_matrix:
    """DMDA Q1 interpolation in one direction for a non-periodic vertex grid refined 2:1:
    nf = 2(nc-1)+1 fine points; fine 2I coincides with coarse I (weight 1), fine 2I+1 lies midway
    between coarse I and I+1 (weights 1/2, 1/2)."""
    nf = 2 * (nc - 1) + 1
    rows, cols, vals = [], [], []
    for f in range(nf):
        if f % 2 == 0:
            rows.append(f); cols.append(f // 2); vals.append(1.0)
        else:
            rows += [f, f]; cols += [f // 2, f // 2 + 1]; vals += [0.5, 0.5]
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf, nc))


def q1_interp(ncx: int, ncy: int, ncz: int = 1) -> sp.csr_matrix:
    """Tensor-product Q1 (bi/tri-linear) interpolation, natural ordering i fastest; a direction
    with a single point is not coarsened."""
    Px = q1_interp_1d(ncx)
    Py = q1_interp_1d(ncy)
    Pz = q1_interp_1d(ncz) if ncz > 1 else sp.identity(1, format="csr")
    return sp.kron(Pz, sp.kron(Py, Px)).tocsr()


def galerkin(A: sp.spmatrix, P: sp.spmatrix) -> sp.csr_matrix:
    """-pc_mg_galerkin both (reference src/pc_gamgmc.c:345-349): A_c = P^T A P."""
    M = (P.T @ A @ P).tocsr()
    M.sort_indices()
    return M


def vcycle(levels, lvl: int, b: np.ndarray, x: np.ndarray, smooth, coarse) -> np.ndarray:
    """PCMG multiplicative V-cycle (PETSc PCMGMCycle_Private semantics): pre-smooth, residual
    r = b - A x, restrict b_c = P^T r, recurse from a ZERO coarse guess, x += P e_c, post-smooth.
    levels[l] = dict(A=scipy csr, P=interp from l-1 to l or None); smooth(l, b, x, leg) and
    coarse(b) are callables returning the new iterate (one KSPSolve with max_it sweeps each)."""
    if lvl == 0:
        return coarse(b)
    L = levels[lvl]
    x = smooth(lvl, b, x, 0)
    r = b - L["A"] @ x
    bc = L["P"].T @ r
    ec = vcycle(levels, lvl - 1, bc, np.zeros(L["P"].shape[1]), smooth, coarse)
    x = x + L["P"] @ ec
    x = smooth(lvl, b, x, 1)
    return x


def gamgmc_richardson(levels, b, y, its: int, guesszero: bool, smooth, coarse, callback=None) -> np.ndarray:
    """PCApplyRichardson_GAMGMC (reference src/pc_gamgmc.c:227-264): first iteration from a zero
    guess is y = MG(b); afterwards w = b - A y, work = MG(w), y += work."""
    top = len(levels) - 1
    y = np.array(y, np.float64, copy=True)
    for it in range(its):
        if it == 0 and guesszero:
            y = vcycle(levels, top, b, np.zeros_like(b), smooth, coarse)
        else:
            w = b - levels[top]["A"] @ y
            work = vcycle(levels, top, w, np.zeros_like(b), smooth, coarse)
            y = y + work
        if callback is not None:
            callback(it, y)
    return y


def potrf_lower(a: np.ndarray) -> np.ndarray:
    """dense lower Cholesky, column-major in/out as a Fortran-ordered copy."""
    n = a.shape[0]
    f = np.asfortranarray(a, dtype=np.float64).copy(order="F")
    flat = np.ascontiguousarray(f.ravel(order="F"))
    info = lib().orc_potrf_lower(n, flat)
    if info:
        raise np.linalg.LinAlgError(f"leading minor {info} not positive definite")
    return np.tril(flat.reshape((n, n), order="F"))


def chol_sample(L: np.ndarray, x: np.ndarray, xi: np.ndarray) -> np.ndarray:
    """PCApply_CholSampler (reference src/pc_chols.c:262-291): y = L^-T (L^-1 x + xi)."""
    n = L.shape[0]
    flat = np.ascontiguousarray(np.asfortranarray(L).ravel(order="F"))
    y = np.zeros(n)
    lib().orc_chol_sample(n, flat, np.ascontiguousarray(x, np.float64), np.ascontiguousarray(xi, np.float64), y)
    return y


# ------------------------------------------------------------------------------------------------
# low-rank updates (MATLRC): A_post = A + B diag(S) B^T
# ------------------------------------------------------------------------------------------------
def lrc_build_correction(A: CSR, colors, B: np.ndarray, S: np.ndarray, omega: float, direction: int) -> np.ndarray:
    """MCSORBuildLRCCorrection (reference src/mc_sor.c:480-544): C = M_A^-1 B column by column with one
    deterministic sweep from a zero guess (:499-510), T = B^T C + S^-1 (:514-527), Sb = T^-1 (:528-533),
    Bb = C Sb (:535)."""
    n, k = B.shape
    Cm = np.stack([mcsor_apply(A, colors, B[:, i], np.zeros(n), omega, direction) for i in range(k)], 1)
    T = B.T @ Cm + np.diag(1.0 / S)
    return Cm @ np.linalg.inv(T)


def lrc_mcsor_apply(A: CSR, colors, B, Bb_f, Bb_b, b, y, omega=1.0, sweep=SOR_FORWARD) -> np.ndarray:
    """MCSORApply on a MATLRC operator (reference src/mc_sor.c:216-239 with postsor = MCSORPostSOR_LRC :101-112):
    every directional sweep is followed by y -= Bb (B^T y)."""
    dirs = [SOR_FORWARD, SOR_BACKWARD] if sweep == SOR_SYMMETRIC else [sweep]
    for d in dirs:
        y = mcsor_apply(A, colors, b, y, omega, d)
        Bb = Bb_f if d == SOR_FORWARD else Bb_b
        y = y - Bb @ (B.T @ y)
    return y


def lrc_gibbs_samples(A: CSR, colors, B, S, b, y0, its, noise_fn, eta_fn, omega=1.0, sweep=SOR_FORWARD, scaled=True) -> np.ndarray:
    """PCApplyRichardson_MulticolorGibbs with prepare_rhs = PrepareRHS_LRC (reference src/pc_mcgibbs.c:130-140,
    :155-188) / PCSORGibbsSample's LRC branch (src/pc_sorgibbs.c:86-101): w = xi*sqrtdiag + b + B (sqrt(S) o eta),
    sweep on A, then y -= Bb (B^T y).  noise_fn(draw) -> xi (N), eta_fn(draw) -> eta (k)."""
    Bb_f = lrc_build_correction(A, colors, B, S, omega, SOR_FORWARD)
    Bb_b = lrc_build_correction(A, colors, B, S, omega, SOR_BACKWARD)
    sd = sqrtdiag(A, omega, scaled)
    sqrtS = np.sqrt(np.abs(S))
    y = np.array(y0, np.float64, copy=True)
    draw = 0
    for _ in range(its):
        for d in ([SOR_FORWARD, SOR_BACKWARD] if sweep == SOR_SYMMETRIC else [sweep]):
            w = prepare_rhs(noise_fn(draw), sd, b) + B @ (sqrtS * eta_fn(draw))
            draw += 1
            y = lrc_mcsor_apply(A, colors, B, Bb_f, Bb_b, w, y, omega, d)
    return y


class LRCOperator:
    """MATLRC: the operator A + B diag(S) B^T applied as MatMult_LRC does (A x first, then the rank-k term);
    stands in for levels[l]["A"] in vcycle / gamgmc_richardson."""

    def __init__(self, A, B: np.ndarray, S: np.ndarray):
        self.A, self.B, self.S = A, np.asarray(B, np.float64), np.asarray(S, np.float64)
        self.shape = A.shape

    def __matmul__(self, x):
        return self.A @ x + self.B @ (self.S * (self.B.T @ x))

    def dense(self) -> np.ndarray:
        A = self.A.toarray() if sp.issparse(self.A) else np.asarray(self.A)
        return A + self.B @ np.diag(self.S) @ self.B.T


def lrc_level_factors(levels, B: np.ndarray):
    """PCGAMGMC_SetUpHierarchy (reference src/pc_gamgmc.c:166-180): B_{l-1} = P_l^T B_l from the finest level down;
    returns [B_0, ..., B_top]."""
    top = len(levels) - 1
    out = [None] * (top + 1)
    out[top] = np.asarray(B, np.float64)
    for l in range(top, 0, -1):
        out[l - 1] = levels[l]["P"].T @ out[l]
    return out


# ------------------------------------------------------------------------------------------------
# chain diagnostics
# ------------------------------------------------------------------------------------------------
def autocorrelation(x: np.ndarray) -> np.ndarray:
    """Autocorrelation (reference src/iact.c:17-47): zero-pad x - mean to 2 * nextpow2(n), |FFT|^2, inverse FFT,
    normalise by lag 0."""
    x = np.asarray(x, np.float64)
    n = len(x)
    N = 1
    while N < n:
        N <<= 1
    f = np.fft.fft(x - x.mean(), 2 * N)
    c = np.fft.ifft(f * np.conj(f)).real[:n]
    return c / c[0]


def iact(x: np.ndarray):
    """IACT + AutoWindow(c=5) (reference src/iact.c:49-92): returns (tau, valid)."""
    n = len(x)
    taus = 2 * np.cumsum(autocorrelation(x)) - 1
    idx = np.arange(n)
    if np.any(idx < 5 * taus):
        ok = np.nonzero(idx >= 5 * taus)[0]
        w = ok[0] if len(ok) else 0
    else:
        w = n - 1
    return float(taus[w]), bool(500 * taus[w] <= n)


def covariance_errors(A: CSR, samples: np.ndarray, chains: int) -> np.ndarray:
    """EstimateCovarianceMatErrors (reference src/stats.c:94-117): samples (samples_per_chain * chains, n),
    sample-major; unbiased covariance over chains per sample index vs A^-1, relative Frobenius error."""
    Q = np.linalg.inv(A.dense())
    spc = samples.shape[0] // chains
    out = np.empty(spc)
    for i in range(spc):
        S = samples[i * chains:(i + 1) * chains]
        W = S - S.mean(0)
        out[i] = np.linalg.norm(W.T @ W / (chains - 1) - Q) / np.linalg.norm(Q)
    return out


# ------------------------------------------------------------------------------------------------
# sampled-row restatements (full-size parity: 256^3 ... 513^3, where a whole CPU sweep takes too long)
# ------------------------------------------------------------------------------------------------
def _rows64(rows) -> np.ndarray:
    return np.ascontiguousarray(rows, np.int64)


def grid7_rows_sweep(nx, ny, nz, kappa, rows, b, y0, y1, omega=1.0, backward=False, noisy=False, scaled=True, seed=0, sweep=0) -> np.ndarray:
    """Values the reference loop (src/mc_sor.c:256-271, red-black colouring) gives the sampled `rows` of the 7-point
    grid operator in one directional sweep that took y0 to y1 (natural-order host vectors)."""
    rows = _rows64(rows)
    out = np.zeros(len(rows))
    lib().orc_grid7_rows_sweep(nx, ny, nz, kappa, omega, int(backward), int(noisy), int(scaled), seed, sweep, len(rows), rows, np.ascontiguousarray(b), np.ascontiguousarray(y0), np.ascontiguousarray(y1), out)
    return out


def grid7_rows_residual(nx, ny, nz, kappa, rows, b, y) -> np.ndarray:
    rows = _rows64(rows)
    out = np.zeros(len(rows))
    lib().orc_grid7_rows_residual(nx, ny, nz, kappa, len(rows), rows, np.ascontiguousarray(b), np.ascontiguousarray(y), out)
    return out


def st27_rows_sweep(nx, ny, nz, coef, sqrtd, rows, b, y0, y1, omega=1.0, backward=False, noisy=False, seed=0, sweep=0) -> np.ndarray:
    """same for a 27-point class-stencil operator under the 8-colour parity colouring (coef: 27 x 27 class table)"""
    rows = _rows64(rows)
    out = np.zeros(len(rows))
    lib().orc_st27_rows(0, nx, ny, nz, np.ascontiguousarray(coef, np.float64).ravel(), np.ascontiguousarray(sqrtd, np.float64), omega, int(backward), int(noisy), seed, sweep, len(rows), rows, np.ascontiguousarray(b), np.ascontiguousarray(y0), np.ascontiguousarray(y1), out)
    return out


def st27_rows_residual(nx, ny, nz, coef, rows, b, y) -> np.ndarray:
    rows = _rows64(rows)
    out = np.zeros(len(rows))
    yy = np.ascontiguousarray(y)
    lib().orc_st27_rows(1, nx, ny, nz, np.ascontiguousarray(coef, np.float64).ravel(), np.zeros(27), 1.0, 0, 0, 0, 0, len(rows), rows, np.ascontiguousarray(b), yy, yy, out)
    return out


def q1_rows_restrict(nf, nc, rows, r_fine) -> np.ndarray:
    rows = _rows64(rows)
    out = np.zeros(len(rows))
    rf = np.ascontiguousarray(r_fine)
    lib().orc_q1_rows(0, np.asarray(nf, np.int32), np.asarray(nc, np.int32), len(rows), rows, rf, rf, out)
    return out


def q1_rows_prolong_add(nf, nc, rows, x_fine, e_coarse) -> np.ndarray:
    rows = _rows64(rows)
    out = np.zeros(len(rows))
    lib().orc_q1_rows(1, np.asarray(nf, np.int32), np.asarray(nc, np.int32), len(rows), rows, np.ascontiguousarray(x_fine), np.ascontiguousarray(e_coarse), out)
    return out


def st27_table_from_csr(nx, ny, nz, A: "CSR"):
    """27 x 27 class table of a structured 27-point matrix (first point of each position class); also returns whether
    EVERY row equals its class stencil bit for bit."""
    coef = np.zeros((27, 27))
    have = np.zeros(27, bool)
    exact = True
    cls_of = lambda i, n: 0 if i == 0 else (2 if i == n - 1 else 1)  # noqa: E731
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                row = i + nx * (j + ny * k)
                cls = cls_of(i, nx) + 3 * cls_of(j, ny) + 9 * cls_of(k, nz)
                loc = np.zeros(27)
                for q in range(A.rowptr[row], A.rowptr[row + 1]):
                    c = int(A.colidx[q])
                    dx, dy, dz = c % nx - i, (c // nx) % ny - j, c // (nx * ny) - k
                    loc[9 * (dz + 1) + 3 * (dy + 1) + dx + 1] = A.vals[q]
                if not have[cls]:
                    coef[cls], have[cls] = loc, True
                elif not np.array_equal(coef[cls], loc):
                    exact = False
    return coef, have, exact
