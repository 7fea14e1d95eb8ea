"""Thin Python mirrors of the C-ABI objects, for tests and benchmarks.

Device vectors are torch CUDA tensors (float64); only their ``data_ptr()`` and the current HIP stream cross
the boundary.  The method names follow the reference API (include/parmgmc/mc_sor.h:21-30)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import check, lib


def _stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    import torch

    assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "need a contiguous float64 CUDA tensor"
    return C.c_void_p(t.data_ptr())


class MCSOR:
    """MCSOR on an assembled AIJ matrix (reference include/parmgmc/mc_sor.h:21-30)."""

    def __init__(self, rowptr, colidx, vals, coloring=capi.COLORING_GREEDY, user_colors=None, idx_width=32):
        """idx_width 64: the index arrays cross the boundary as 64-bit PetscInt (pmg_mcsor_create_csr_idx)"""
        it = np.int64 if idx_width == 64 else np.int32
        self._rowptr = np.ascontiguousarray(rowptr, it)
        self._colidx = np.ascontiguousarray(colidx, it)
        self._vals = np.ascontiguousarray(vals, np.float64)
        self.n = len(self._rowptr) - 1
        self._h = C.c_void_p()
        if idx_width == 32:
            check(lib.pmg_mcsor_create_csr(self.n, self._rowptr.ctypes.data, self._colidx.ctypes.data, self._vals.ctypes.data, C.byref(self._h)))
        else:
            check(lib.pmg_mcsor_create_csr_idx(self.n, self._rowptr.ctypes.data, self._colidx.ctypes.data, self._vals.ctypes.data, idx_width, C.byref(self._h)))
        uc = None
        if user_colors is not None:
            uc = np.ascontiguousarray(user_colors, np.int32)
            coloring = capi.COLORING_USER
        try:
            check(lib.pmg_mcsor_set_coloring(self._h, coloring, None if uc is None else uc.ctypes.data))
        except Exception:
            self.destroy()
            raise

    def setup(self):
        check(lib.pmg_mcsor_setup(self._h))
        return self

    def set_omega(self, omega: float):
        check(lib.pmg_mcsor_set_omega(self._h, omega))

    def set_sweep_type(self, t: int):
        check(lib.pmg_mcsor_set_sweep_type(self._h, t))

    def get_sweep_type(self) -> int:
        t = C.c_int()
        check(lib.pmg_mcsor_get_sweep_type(self._h, C.byref(t)))
        return t.value

    def get_num_colors(self) -> int:
        n = C.c_int32()
        check(lib.pmg_mcsor_get_num_colors(self._h, C.byref(n)))
        return n.value

    def get_coloring(self) -> np.ndarray:
        out = np.zeros(max(self.n, 1), np.int32)
        check(lib.pmg_mcsor_get_coloring(self._h, out.ctypes.data))
        return out[: self.n]

    def apply(self, b, y):
        check(lib.pmg_mcsor_apply(self._h, _ptr(b), _ptr(y), _stream()))

    def sample(self, b, y, its: int, seed: int, counter0: int = 0, scaled: bool = True) -> int:
        out = C.c_uint64()
        check(lib.pmg_mcsor_sample(self._h, _ptr(b), _ptr(y), its, int(scaled), seed, counter0, C.byref(out), _stream()))
        return out.value

    def residual(self, b, y, r):
        check(lib.pmg_mcsor_residual(self._h, _ptr(b), _ptr(y), _ptr(r), _stream()))

    # --- storage layout and per-colour sweeps (building blocks of the row-block distributed sampler) ---
    def layout_len(self) -> int:
        n = C.c_int32()
        check(lib.pmg_mcsor_layout_len(self._h, C.byref(n)))
        return n.value

    def get_layout(self) -> np.ndarray:
        """position of every row in the library's storage layout"""
        out = np.zeros(max(self.n, 1), np.int32)
        check(lib.pmg_mcsor_get_layout(self._h, out.ctypes.data))
        return out[: self.n]

    def set_noise_row_offset(self, row0: int):
        check(lib.pmg_mcsor_set_noise_row_offset(self._h, row0))

    def sweep_color_layout(self, color: int, b_lay, y_lay, noisy: bool = True, scaled: bool = True, seed: int = 0, counter: int = 0):
        check(lib.pmg_mcsor_sweep_color_layout(self._h, color, int(noisy), int(scaled), seed, counter, _ptr(b_lay), _ptr(y_lay), _stream()))

    def set_lowrank(self, B, S):
        """MATLRC A + B diag(S) B^T: B (n x k), S (k) host arrays (reference src/mc_sor.c:572-595)."""
        B = np.asfortranarray(B, dtype=np.float64)
        S = np.ascontiguousarray(S, np.float64)
        check(lib.pmg_mcsor_set_lowrank(self._h, B.shape[1], B.ctypes.data, S.ctypes.data))

    def destroy(self):
        if self._h:
            check(lib.pmg_mcsor_destroy(C.byref(self._h)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class GridMCSOR:
    """Matrix-free MCSOR on a DMDA grid for the operator of MatAssembleShiftedLaplaceFD
    (reference src/problems.c:14-75); owns planes [kz0, kz0+nz) of nx*ny*nzg."""

    def __init__(self, nx, ny, nz=1, kappa=1.0, kz0=0, nz_owned=None):
        self.nx, self.ny, self.nzg = nx, ny, nz
        self.kz0, self.nz = kz0, (nz if nz_owned is None else nz_owned)
        self._h = C.c_void_p()
        check(lib.pmg_grid_create(nx, ny, nz, self.kz0, self.nz, kappa, C.byref(self._h)))
        n = C.c_int64()
        check(lib.pmg_grid_cvec_len(self._h, C.byref(n)))
        self.cvec_len = n.value
        self.n = nx * ny * self.nz

    def new_cvec(self):
        import torch

        return torch.zeros(self.cvec_len, dtype=torch.float64, device="cuda")

    def to_cvec(self, nat, out=None):
        out = self.new_cvec() if out is None else out
        check(lib.pmg_grid_to_cvec(self._h, _ptr(nat), _ptr(out), _stream()))
        return out

    def from_cvec(self, cvec, out=None):
        import torch

        out = torch.empty(self.n, dtype=torch.float64, device="cuda") if out is None else out
        check(lib.pmg_grid_from_cvec(self._h, _ptr(cvec), _ptr(out), _stream()))
        return out

    def set_omega(self, omega: float):
        check(lib.pmg_grid_set_omega(self._h, omega))

    def set_sweep_type(self, t: int):
        check(lib.pmg_grid_set_sweep_type(self._h, t))

    def get_sweep_type(self) -> int:
        t = C.c_int()
        check(lib.pmg_grid_get_sweep_type(self._h, C.byref(t)))
        return t.value

    def get_num_colors(self) -> int:
        n = C.c_int32()
        check(lib.pmg_grid_get_num_colors(self._h, C.byref(n)))
        return n.value

    def get_coloring(self) -> np.ndarray:
        out = np.zeros(self.n, np.int32)
        check(lib.pmg_grid_get_coloring(self._h, out.ctypes.data))
        return out

    def apply(self, b, y):
        check(lib.pmg_grid_apply(self._h, _ptr(b), _ptr(y), _stream()))

    def apply_cvec(self, b, y):
        check(lib.pmg_grid_apply_cvec(self._h, _ptr(b), _ptr(y), _stream()))

    def sample(self, b, y, its: int, seed: int, counter0: int = 0, scaled: bool = True) -> int:
        out = C.c_uint64()
        check(lib.pmg_grid_sample(self._h, _ptr(b), _ptr(y), its, int(scaled), seed, counter0, C.byref(out), _stream()))
        return out.value

    def sample_cvec(self, b, y, its: int, seed: int, counter0: int = 0, scaled: bool = True) -> int:
        out = C.c_uint64()
        check(lib.pmg_grid_sample_cvec(self._h, _ptr(b), _ptr(y), its, int(scaled), seed, counter0, C.byref(out), _stream()))
        return out.value

    def sweep_color_cvec(self, color: int, b, y, noisy: bool = False, scaled: bool = True, seed: int = 0, counter: int = 0):
        check(lib.pmg_grid_sweep_color_cvec(self._h, color, int(noisy), int(scaled), seed, counter, _ptr(b), _ptr(y), _stream()))

    def sweep_color_planes_cvec(self, color: int, kbegin: int, kcount: int, b, y, noisy: bool = False, scaled: bool = True, seed: int = 0, counter: int = 0):
        check(lib.pmg_grid_sweep_color_planes_cvec(self._h, color, kbegin, kcount, int(noisy), int(scaled), seed, counter, _ptr(b), _ptr(y), _stream()))

    def halo_plane(self, color: int, side: int):
        """(owned_offset, ghost_offset, count) in doubles inside a cvec."""
        a, b, n = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.pmg_grid_halo_plane(self._h, color, side, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def set_lowrank(self, B, S):
        B = np.asfortranarray(B, dtype=np.float64)
        S = np.ascontiguousarray(S, np.float64)
        check(lib.pmg_grid_set_lowrank(self._h, B.shape[1], B.ctypes.data, S.ctypes.data))

    def residual_cvec(self, b, y, r):
        check(lib.pmg_grid_residual_cvec(self._h, _ptr(b), _ptr(y), _ptr(r), _stream()))

    def destroy(self):
        if self._h:
            check(lib.pmg_grid_destroy(C.byref(self._h)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class CholSampler:
    """Exact coarse sampler (reference PCCHOLSAMPLER dense path, src/pc_chols.c:174-291)."""

    def __init__(self, rowptr, colidx, vals, B=None, S=None, idx_width=32):
        """B (n x k), S (k): factor the MATLRC operator A + B diag(S) B^T instead (src/pc_chols.c:119-153)."""
        it = np.int64 if idx_width == 64 else np.int32
        rp, ci, v = np.ascontiguousarray(rowptr, it), np.ascontiguousarray(colidx, it), np.ascontiguousarray(vals, np.float64)
        self.n = len(rp) - 1
        self._h = C.c_void_p()
        if idx_width != 32:
            k = 0 if B is None else np.asarray(B).shape[1]
            Bf = None if B is None else np.asfortranarray(B, np.float64)
            Sf = None if S is None else np.ascontiguousarray(S, np.float64)
            check(lib.pmg_chol_create_csr_idx(self.n, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, idx_width, k, None if Bf is None else Bf.ctypes.data, None if Sf is None else Sf.ctypes.data, C.byref(self._h)))
        elif B is None:
            check(lib.pmg_chol_create_csr(self.n, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, C.byref(self._h)))
        else:
            B = np.asfortranarray(B, np.float64)
            S = np.ascontiguousarray(S, np.float64)
            assert B.shape == (self.n, len(S))
            check(lib.pmg_chol_create_csr_lowrank(self.n, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, B.shape[1], B.ctypes.data, S.ctypes.data, C.byref(self._h)))

    def factor(self) -> np.ndarray:
        out = np.zeros(self.n * self.n)
        check(lib.pmg_chol_get_factor(self._h, out.ctypes.data))
        return out.reshape((self.n, self.n), order="F")

    def sample(self, b, y, seed: int = 0, counter: int = 0, noisy: bool = True):
        check(lib.pmg_chol_sample(self._h, _ptr(b), _ptr(y), int(noisy), seed, counter, _stream()))

    def destroy(self):
        if self._h:
            check(lib.pmg_chol_destroy(C.byref(self._h)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class MGMC:
    """Multigrid Monte Carlo on a DMDA hierarchy (reference PCGAMGMC, src/pc_gamgmc.c, with -pc_gamgmc_mg_type mg)."""

    def __init__(self, nx, ny, nz, kappa, levels, keep_host=False):
        self.nx, self.ny, self.nz, self.n = nx, ny, nz, nx * ny * nz
        self._h = C.c_void_p()
        check(lib.pmg_mgmc_create_dmda(nx, ny, nz, kappa, levels, C.byref(self._h)))
        check(lib.pmg_mgmc_set_keep_host(self._h, int(keep_host)))
        self.levels = levels

    @classmethod
    def from_hierarchy(cls, operators, interpolations, idx_width=32):
        """operators[l] = (rowptr, colidx, vals) of level l (0 = coarsest); interpolations[l] (l >= 1) = CSR triple of
        the prolongation from level l-1 to level l (reference src/pc_gamgmc.c:165-176: what PCMG holds).  idx_width 64:
        the index arrays cross the boundary as 64-bit PetscInt and need not outlive the calls."""
        self = cls.__new__(cls)
        levels = len(operators)
        self._h = C.c_void_p()
        self.levels = levels
        self._keep = []
        it = np.int64 if idx_width == 64 else np.int32
        check(lib.pmg_mgmc_create_hierarchy(levels, C.byref(self._h)))
        for l, (rp, ci, v) in enumerate(operators):
            rp, ci, v = np.ascontiguousarray(rp, it), np.ascontiguousarray(ci, it), np.ascontiguousarray(v, np.float64)
            self._keep.append((rp, ci, v) if idx_width == 32 else (v,))
            if idx_width == 32:
                check(lib.pmg_mgmc_set_level_operator(self._h, l, len(rp) - 1, rp.ctypes.data, ci.ctypes.data, v.ctypes.data))
            else:
                check(lib.pmg_mgmc_set_level_operator_idx(self._h, l, len(rp) - 1, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, idx_width))
        for l in range(1, levels):
            rp, ci, v = interpolations[l]
            rp, ci, v = np.ascontiguousarray(rp, it), np.ascontiguousarray(ci, it), np.ascontiguousarray(v, np.float64)
            self._keep.append((rp, ci, v) if idx_width == 32 else (v,))
            nc = len(operators[l - 1][0]) - 1
            if idx_width == 32:
                check(lib.pmg_mgmc_set_level_interpolation(self._h, l, len(rp) - 1, nc, rp.ctypes.data, ci.ctypes.data, v.ctypes.data))
            else:
                check(lib.pmg_mgmc_set_level_interpolation_idx(self._h, l, len(rp) - 1, nc, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, idx_width))
        self.n = len(operators[-1][0]) - 1
        return self

    def set_smoother(self, scaled: bool, omega: float = 1.0, sweep_type: int = capi.SOR_FORWARD_SWEEP, its: int = 1):
        check(lib.pmg_mgmc_set_smoother(self._h, int(scaled), omega, sweep_type, its))

    def set_correction_form(self, literal: bool):
        check(lib.pmg_mgmc_set_correction_form(self._h, int(literal)))

    def set_fused_transfers(self, on: bool):
        """False: residual and restriction as two kernels, a low-rank term subtracted before the restriction (the
        reference's operation order); True (default): the fused kernel and the restricted low-rank term."""
        check(lib.pmg_mgmc_set_fused_transfers(self._h, int(on)))

    def set_coloring(self, rule: int):
        """colouring rule of the AIJ levels (capi.COLORING_GREEDY, the default, or capi.COLORING_ITERATED); before setup"""
        check(lib.pmg_mgmc_set_coloring(self._h, int(rule)))

    def set_lowrank(self, B, S):
        """MATLRC fine operator A + B diag(S) B^T, propagated to every level (reference src/pc_gamgmc.c:157-196)."""
        B = np.asfortranarray(B, np.float64)
        S = np.ascontiguousarray(S, np.float64)
        assert B.ndim == 2 and B.shape[1] == len(S)
        check(lib.pmg_mgmc_set_lowrank(self._h, B.shape[1], B.ctypes.data, S.ctypes.data))

    def set_coarse(self, kind: str = "cholsampler", its: int = 1):
        check(lib.pmg_mgmc_set_coarse(self._h, {"cholsampler": 0, "gibbs": 1}[kind], its))

    def setup(self):
        check(lib.pmg_mgmc_setup(self._h))
        return self

    def level_dims(self, level: int):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.pmg_mgmc_get_level_dims(self._h, level, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def level_matrix(self, level: int, which: str):
        w = {"A": 0, "P": 1}[which]
        nr, nnz = C.c_int32(), C.c_int32()
        check(lib.pmg_mgmc_get_level_matrix(self._h, level, w, C.byref(nr), C.byref(nnz), None, None, None))
        rp, ci, v = np.zeros(nr.value + 1, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nnz.value)
        check(lib.pmg_mgmc_get_level_matrix(self._h, level, w, C.byref(nr), C.byref(nnz), rp.ctypes.data, ci.ctypes.data, v.ctypes.data))
        return rp, ci, v

    # --- one kernel of the V-cycle on caller vectors in the level's own layout (diagnostics for full-size parity tests) ---
    def level_layout(self, level: int):
        """(kind, ld, off): kind 0 = grid cvec, 1 = class stencil, 2 = sliced-ELL, 3 = dense coarsest; natural index q of
        kinds 1 and 3 sits at off + q"""
        k, ld, off = C.c_int32(), C.c_int64(), C.c_int64()
        check(lib.pmg_mgmc_get_level_layout(self._h, level, C.byref(k), C.byref(ld), C.byref(off)))
        return k.value, ld.value, off.value

    def level_stencil(self, level: int):
        coef, sq = np.zeros((27, 27)), np.zeros(27)
        check(lib.pmg_mgmc_get_level_stencil(self._h, level, coef.ctypes.data, sq.ctypes.data))
        return coef, sq

    def level_sweep(self, level: int, b, x, backward: bool = False, noisy: bool = False, seed: int = 0, counter: int = 0):
        check(lib.pmg_mgmc_level_sweep(self._h, level, int(backward), int(noisy), seed, counter, _ptr(b), _ptr(x), _stream()))

    def level_residual(self, level: int, b, x, r):
        check(lib.pmg_mgmc_level_residual(self._h, level, _ptr(b), _ptr(x), _ptr(r), _stream()))

    def level_restrict(self, level: int, r_fine, b_coarse):
        check(lib.pmg_mgmc_level_restrict(self._h, level, _ptr(r_fine), _ptr(b_coarse), _stream()))

    def level_residual_restrict(self, level: int, b, x, b_coarse):
        check(lib.pmg_mgmc_level_residual_restrict(self._h, level, _ptr(b), _ptr(x), _ptr(b_coarse), _stream()))

    def level_prolong_add(self, level: int, e_coarse, x_fine):
        check(lib.pmg_mgmc_level_prolong_add(self._h, level, _ptr(e_coarse), _ptr(x_fine), _stream()))

    def algorithmic_bytes(self):
        """(total, per_level): algorithmic bytes of one sample as the cycle is built (pmg_mgmc_get_algorithmic_bytes)"""
        tot = C.c_double()
        per = np.zeros(self.levels)
        check(lib.pmg_mgmc_get_algorithmic_bytes(self._h, C.byref(tot), per.ctypes.data))
        return tot.value, per

    def level_lowrank_factors(self, level: int):
        """(rows, B, Bb_fwd, Bb_bwd): layout positions of the support rows and the ns x k blocks the kernels use"""
        k, ns = C.c_int32(), C.c_int64()
        check(lib.pmg_mgmc_level_lowrank_factors(self._h, level, C.byref(k), C.byref(ns), None, None, None, None))
        rows = np.zeros(ns.value, np.int64)
        B, Bf, Bb = (np.zeros((ns.value, k.value), order="F") for _ in range(3))
        check(lib.pmg_mgmc_level_lowrank_factors(self._h, level, C.byref(k), C.byref(ns), rows.ctypes.data, B.ctypes.data, Bf.ctypes.data, Bb.ctypes.data))
        return rows, B, Bf, Bb

    def level_lowrank_post(self, level: int, y, backward: bool = False):
        check(lib.pmg_mgmc_level_lowrank_post(self._h, level, int(backward), _ptr(y), _stream()))

    def level_lowrank_residual_sub(self, level: int, x, out, restricted: bool = False):
        check(lib.pmg_mgmc_level_lowrank_residual_sub(self._h, level, int(restricted), _ptr(x), _ptr(out), _stream()))

    def sample(self, b, y, its: int, seed: int, counter0: int = 0, guesszero: bool = False, callback=None) -> int:
        out = C.c_uint64()
        if callback is None:
            cb = None
        else:
            import torch

            def _cb(it, ptr, n, _ctx):
                try:
                    callback(it, y)  # y is the tensor the library writes the sample into
                    return 0
                except Exception:  # pragma: no cover
                    import traceback

                    traceback.print_exc()
                    return 77

            cb = capi.SAMPLE_CALLBACK(_cb)
        check(lib.pmg_mgmc_sample(self._h, _ptr(b), _ptr(y), its, int(guesszero), seed, counter0, C.byref(out), cb, None, _stream()))
        return out.value

    def destroy(self):
        if self._h:
            check(lib.pmg_mgmc_destroy(C.byref(self._h)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class WoodburySampler:
    """PCWOODBURY (reference src/woodbury.c): posterior sampler for A + B diag(S) B^T from ANY sampler of A plus a solver --
    on one device or on rows distributed over ranks (z-slabs, row blocks): B_rows = this rank's rows of B (n x k), vectors in
    natural order over those rows; `dist_handle` = the C handle of a transport of the ranks (ctypes void pointer) or None.
      solve(b, x)            x <- (approximately) A^-1 b from the zero guess x holds (device tensors), collective
      sample(w, y, counter)  one sample of the A-sampler on right-hand side w, y updated in place, collective"""

    def __init__(self, B_rows, S, solve, sample, dist_handle=None):
        import torch

        B = np.asfortranarray(B_rows, np.float64)
        S = np.ascontiguousarray(S, np.float64)
        self.n, self.k = B.shape
        assert len(S) == self.k
        self._h = C.c_void_p()
        self._sample = sample
        check(lib.pmg_woodbury_create(self.n, self.k, B.ctypes.data, max(self.n, 1), S.ctypes.data, dist_handle, C.byref(self._h)))
        for c in range(self.k):  # C = solver(B) column by column from a zero guess (src/woodbury.c:35-50)
            b = torch.as_tensor(np.ascontiguousarray(B[:, c]), device="cuda")
            x = torch.zeros(self.n, dtype=torch.float64, device="cuda")
            solve(b, x)
            check(lib.pmg_woodbury_set_c_column(self._h, c, _ptr(x), _stream()))
        torch.cuda.synchronize()
        check(lib.pmg_woodbury_finish(self._h))
        self._w = torch.zeros(self.n, dtype=torch.float64, device="cuda")

    def correction(self):
        G = np.zeros((self.n, self.k), order="F")
        check(lib.pmg_woodbury_get_correction(self._h, G.ctypes.data))
        return G

    def run(self, b, y, its: int, seed: int, counter0: int = 0, callback=None) -> int:
        """PCApplyRichardson_Woodbury (src/woodbury.c:263-289); sample `it` uses noise counter counter0 + it for the k-vector
        and hands the same counter to the A-sampler"""
        for it in range(its):
            check(lib.pmg_woodbury_noisy_rhs(self._h, _ptr(b), _ptr(self._w), seed ^ 0x5851F42D4C957F2D, counter0 + it, _stream()))
            self._sample(self._w, y, counter0 + it)
            check(lib.pmg_woodbury_correct(self._h, _ptr(y), _stream()))
            if callback is not None:
                callback(it, y)
        return counter0 + its

    def destroy(self):
        if self._h:
            check(lib.pmg_woodbury_destroy(C.byref(self._h)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def vec_set_random_standard_normal(x, seed: int, counter: int = 0):
    """VecSetRandomStandardNormal (reference src/parmgmc.c:70-116) on the counter-based source."""
    check(lib.pmg_vec_set_random_standard_normal(x.numel(), _ptr(x), seed, counter, _stream()))
    return x


def autocorrelation(x):
    """Autocorrelation (reference src/iact.c:17-47) of a scalar series; host arrays."""
    x = np.ascontiguousarray(x, np.float64)
    acf = np.empty_like(x)
    check(lib.pmg_autocorrelation(len(x), x.ctypes.data, acf.ctypes.data))
    return acf


def iact(x):
    """IACT (reference src/iact.c:73-92): returns (tau, valid)."""
    x = np.ascontiguousarray(x, np.float64)
    tau, valid = C.c_double(), C.c_int()
    check(lib.pmg_iact(len(x), x.ctypes.data, C.byref(tau), None, C.byref(valid)))
    return tau.value, bool(valid.value)


def estimate_covariance_errors(rowptr, colidx, vals, samples, chains: int):
    """EstimateCovarianceMatErrors (reference src/stats.c:94-117); samples: (samples_per_chain * chains, n) host rows
    ordered sample-major."""
    rp, ci, v = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colidx, np.int32), np.ascontiguousarray(vals, np.float64)
    S = np.ascontiguousarray(samples, np.float64)
    n = len(rp) - 1
    assert S.ndim == 2 and S.shape[1] == n and S.shape[0] % chains == 0
    spc = S.shape[0] // chains
    errs = np.empty(spc)
    check(lib.pmg_estimate_covariance_errors(n, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, chains, spc, S.ctypes.data, errs.ctypes.data))
    return errs


def make_observation_mats(nx, ny, nz, coords, radii, obsvals, sigma2, kz0=0, nz_owned=None):
    """MakeObservationMats (reference src/obs.c:135-180) on the unit-cube DMDA: returns (B, S, f) as host arrays, rows =
    the planes [kz0, kz0 + nz_owned) in natural order."""
    nz_owned = nz if nz_owned is None else nz_owned
    coords = np.ascontiguousarray(coords, np.float64).ravel()
    radii = np.ascontiguousarray(radii, np.float64)
    vals = np.ascontiguousarray(obsvals, np.float64)
    k, n = len(radii), nx * ny * nz_owned
    B = np.zeros((n, k), order="F")
    S, f = np.zeros(k), np.zeros(n)
    check(lib.pmg_make_observation_mats_dmda(nx, ny, nz, kz0, nz_owned, k, sigma2, coords.ctypes.data, radii.ctypes.data, vals.ctypes.data, B.ctypes.data, S.ctypes.data, f.ctypes.data))
    return B, S, f
