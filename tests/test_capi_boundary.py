"""The C-ABI library loads, exports every symbol include/parmgmc_hip.h declares, and keeps the reference's
error behaviour for calls that need no GPU.  CPU only: no compute call is made here."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from parmgmc_amd import capi
from parmgmc_amd.capi import lib


def test_library_exports_every_declared_symbol():
    names = capi.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and nothing torch/petsc leaks into the dynamic dependencies
    out = subprocess.run(["ldd", str(capi.library_path())], capture_output=True, text=True).stdout
    assert "libamdhip64" in out and "torch" not in out and "petsc" not in out


def test_every_declared_symbol_has_a_python_prototype():
    assert sorted(capi._sig) == capi.declared_symbols()


def test_version_and_arch():
    assert lib.pmg_gpu_arch() == b"gfx950"
    assert lib.pmg_version().count(b".") == 2


def test_code_object_is_gfx950():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(capi.library_path())], capture_output=True, text=True).stdout
    txt = subprocess.run(["strings", str(capi.library_path())], capture_output=True, text=True).stdout
    assert "gfx950" in out + txt


def test_sweep_type_errors_like_the_reference():
    """reference src/mc_sor.c:427: anything but forward/backward/symmetric raises PETSC_ERR_SUP (56)."""
    h = C.c_void_p()
    assert lib.pmg_grid_create(9, 9, 1, 0, 1, 10.0, C.byref(h)) == 0
    assert lib.pmg_grid_set_sweep_type(h, 4) == 56  # SOR_LOCAL_FORWARD_SWEEP is not supported by MCSOR
    assert b"Only forward, backward and symmetric sweep supported" in lib.pmg_last_error_string()
    t = C.c_int()
    for typ in (1, 2, 3):
        assert lib.pmg_grid_set_sweep_type(h, typ) == 0
        assert lib.pmg_grid_get_sweep_type(h, C.byref(t)) == 0 and t.value == typ
    n = C.c_int32()
    assert lib.pmg_grid_get_num_colors(h, C.byref(n)) == 0 and n.value == 2
    assert lib.pmg_grid_destroy(C.byref(h)) == 0 and not h.value
    assert lib.pmg_grid_destroy(C.byref(h)) == 0  # destroying NULL is a no-op (src/mc_sor.c:63)


def test_argument_errors():
    h = C.c_void_p()
    assert lib.pmg_grid_create(1, 4, 1, 0, 1, 1.0, C.byref(h)) == 63  # nx = 1 would divide by zero (problems.c:24)
    assert lib.pmg_grid_create(4, 4, 4, 2, 3, 1.0, C.byref(h)) == 63  # owned planes outside the grid
    assert lib.pmg_grid_create(4, 4, 4, 0, 4, 1.0, None) == 85
    assert lib.pmg_grid_set_omega(None, 1.0) == 85
    assert lib.pmg_mcsor_setup(None) == 85


def test_grid_coloring_is_bit_exact_red_black():
    import oracle as O

    for (nx, ny, nz) in [(9, 9, 1), (6, 5, 4), (2, 3, 5)]:
        h = C.c_void_p()
        assert lib.pmg_grid_create(nx, ny, nz, 0, nz, 1.0, C.byref(h)) == 0
        col = np.zeros(nx * ny * nz, np.int32)
        assert lib.pmg_grid_get_coloring(h, col.ctypes.data) == 0
        assert np.array_equal(col, O.coloring_redblack(nx, ny, nz))
        ln = C.c_int64()
        assert lib.pmg_grid_cvec_len(h, C.byref(ln)) == 0 and ln.value >= nx * ny * nz
        lib.pmg_grid_destroy(C.byref(h))
    # a slab of a larger grid keeps the GLOBAL parity
    h = C.c_void_p()
    assert lib.pmg_grid_create(6, 5, 4, 1, 2, 1.0, C.byref(h)) == 0
    col = np.zeros(60, np.int32)
    lib.pmg_grid_get_coloring(h, col.ctypes.data)
    assert np.array_equal(col, O.coloring_redblack(6, 5, 4)[30:90])
    lib.pmg_grid_destroy(C.byref(h))
