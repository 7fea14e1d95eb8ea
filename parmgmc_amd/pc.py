"""Python mirror of the PC layer (include/parmgmc_hip.h, "registration boundary" section): what a user of the
reference writes against PETSc's KSP/PC API (reference examples/ex1.c, ex3.c, ex8.c) written against the C-ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import check, lib
from .wrappers import _ptr, _stream


def initialize():
    """ParMGMCInitialize (reference src/parmgmc.c:118-127)."""
    check(lib.pmg_initialize())


def finalize():
    check(lib.pmg_finalize())


def options_set_value(name: str, value: str = ""):
    check(lib.pmg_options_set_value(name.encode(), str(value).encode()))


def options_clear():
    check(lib.pmg_options_clear())


def set_seed(seed: int):
    check(lib.pmg_set_seed(seed))


class Mat:
    def __init__(self, handle, keep=None):
        self._h, self._keep = handle, keep

    @staticmethod
    def csr(rowptr, colidx, vals) -> "Mat":
        rp, ci, v = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colidx, np.int32), np.ascontiguousarray(vals, np.float64)
        h = C.c_void_p()
        check(lib.pmg_mat_create_csr(len(rp) - 1, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, C.byref(h)))
        return Mat(h, (rp, ci, v))

    @staticmethod
    def dmda(nx, ny, nz, kappa) -> "Mat":
        h = C.c_void_p()
        check(lib.pmg_mat_create_dmda(nx, ny, nz, kappa, C.byref(h)))
        return Mat(h)

    def lrc(self, B, S) -> "Mat":
        """MatCreateLRC(A, B, S): A + B diag(S) B^T (reference examples/ex4.c)."""
        B = np.asfortranarray(B, np.float64)
        S = np.ascontiguousarray(S, np.float64)
        assert B.ndim == 2 and B.shape == (self.size, len(S))
        h = C.c_void_p()
        check(lib.pmg_mat_create_lrc(self._h, B.shape[1], B.ctypes.data, S.ctypes.data, C.byref(h)))
        return Mat(h, (self, B, S))

    @property
    def size(self) -> int:
        n = C.c_int32()
        check(lib.pmg_mat_get_size(self._h, C.byref(n)))
        return n.value


class PC:
    def __init__(self, pc_type: str | None = None, prefix: str = ""):
        self._h = C.c_void_p()
        check(lib.pmg_pc_create(C.byref(self._h)))
        self._cb_keep = []
        if prefix:
            check(lib.pmg_pc_set_options_prefix(self._h, prefix.encode()))
        if pc_type:
            self.set_type(pc_type)

    def set_type(self, t: str):
        check(lib.pmg_pc_set_type(self._h, t.encode()))

    def get_type(self) -> str:
        buf = C.create_string_buffer(64)
        check(lib.pmg_pc_get_type(self._h, buf, 64))
        return buf.value.decode()

    def set_operators(self, mat: Mat):
        self._mat = mat
        check(lib.pmg_pc_set_operators(self._h, mat._h))

    def set_from_options(self):
        check(lib.pmg_pc_set_from_options(self._h))

    def setup(self):
        check(lib.pmg_pc_setup(self._h))

    def view(self) -> str:
        buf = C.create_string_buffer(512)
        check(lib.pmg_pc_view(self._h, buf, 512))
        return buf.value.decode()

    def apply(self, b, y):
        check(lib.pmg_pc_apply(self._h, _ptr(b), _ptr(y), _stream()))

    def apply_richardson(self, b, y, its: int, guesszero: bool = False):
        outits, reason = C.c_int32(), C.c_int32()
        check(lib.pmg_pc_apply_richardson(self._h, _ptr(b), _ptr(y), its, int(guesszero), C.byref(outits), C.byref(reason), _stream()))
        return outits.value, reason.value

    def ksp_solve(self, b, y, max_it: int, guess_nonzero: bool = True):
        """KSPSolve, -ksp_type richardson (reference examples/ex1.c:123-129)."""
        check(lib.pmg_ksp_richardson_solve(self._h, _ptr(b), _ptr(y), max_it, int(guess_nonzero), _stream()))

    def set_sample_callback(self, fn, y_tensor, deleter=None):
        """fn(it, y_tensor) after every sample; y_tensor is the solution tensor passed to the solve."""

        def _cb(it, _ptr_, _n, _ctx):
            try:
                fn(it, y_tensor)
                return 0
            except Exception:  # pragma: no cover
                import traceback

                traceback.print_exc()
                return 77

        cb = capi.SAMPLE_CALLBACK(_cb)
        dl = capi.DELETER(lambda _c: (deleter() if deleter else None) or 0)
        self._cb_keep.append((cb, dl))
        check(lib.pmg_pc_set_sample_callback(self._h, cb, None, dl if deleter else None))

    def noise_state(self):
        """(effective Philox seed of this PC's stream, counter of its next draw)."""
        s, c = C.c_uint64(), C.c_uint64()
        check(lib.pmg_pc_get_noise_state(self._h, C.byref(s), C.byref(c)))
        return s.value, c.value

    def set_noise_counter(self, counter: int):
        check(lib.pmg_pc_set_noise_counter(self._h, counter))

    def woodbury_set_solver(self, solver: "PC"):
        """PCWoodburySetSolver (reference src/woodbury.c:185-198); the woodbury PC takes the inner PC over."""
        check(lib.pmg_pc_woodbury_set_solver(self._h, solver._h))
        solver._borrowed = True
        self._inner = getattr(self, "_inner", []) + [solver]

    def woodbury_set_sampler(self, sampler: "PC"):
        """PCWoodburySetSampler (reference src/woodbury.c:200-213)."""
        check(lib.pmg_pc_woodbury_set_sampler(self._h, sampler._h))
        sampler._borrowed = True
        self._inner = getattr(self, "_inner", []) + [sampler]

    # --- PCPARSOR (reference include/parmgmc/pc/pc_parsor.h) ---
    def parsor_set_omega(self, omega: float):
        check(lib.pmg_pc_parsor_set_omega(self._h, omega))

    def parsor_set_iterations(self, its: int):
        check(lib.pmg_pc_parsor_set_iterations(self._h, its))

    def parsor_apply_sor(self, b, its: int, zero_initial_guess: bool, x):
        """PCPARSORApplySOR (reference src/pc_parsor.c:892-904)"""
        check(lib.pmg_pc_parsor_apply_sor(self._h, _ptr(b), its, int(zero_initial_guess), _ptr(x), _stream()))

    def parsor_set_partition(self, row_starts, proc_colors=None):
        """reproduce the sweep of len(row_starts)-1 MPI ranks owning contiguous row blocks (src/pc_parsor.c:703-878)"""
        rs = np.ascontiguousarray(row_starts, np.int32)
        nparts = max(len(rs) - 1, 0)  # an empty list returns to the single-rank order
        pcs = None if proc_colors is None else np.ascontiguousarray(proc_colors, np.int32)
        check(lib.pmg_pc_parsor_set_partition(self._h, nparts, rs.ctypes.data if nparts else None, None if pcs is None else pcs.ctypes.data))
        self._parsor_nparts = nparts

    def parsor_partition_info(self):
        """(number of dependency levels, rank colours, row classes 0 INT / 1 TOP / 2 MID / 3 BOT); sets the PC up"""
        nl = C.c_int32()
        pcs, cls = np.zeros(self._parsor_nparts, np.int32), np.zeros(self._mat.size, np.int32)
        check(lib.pmg_pc_parsor_get_partition_info(self._h, C.byref(nl), pcs.ctypes.data, cls.ctypes.data))
        return nl.value, pcs, cls

    def reset(self):
        check(lib.pmg_pc_reset(self._h))

    def destroy(self):
        if getattr(self, "_borrowed", False):  # owned by a woodbury PC
            self._h = C.c_void_p()
            return
        if self._h:
            check(lib.pmg_pc_destroy(C.byref(self._h)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
