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
    assert lib.pmg_stream_triad(7, None, None, None, None) == 63  # odd length
    assert lib.pmg_stream_triad(8, None, None, None, None) == 85
    assert lib.pmg_stream_triad(8, 16, 32, 40, None) == 62  # 8-byte aligned only (checked before anything is launched)
    assert lib.pmg_stream_triad(0, None, None, None, None) == 0


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


def test_idx_width_entry_points_check_their_arguments():
    """64-bit PetscInt builds (reference include/parmgmc/parmgmc.h:18-24): the _idx entry points narrow the index arrays
    with range checks; no GPU is touched before set-up."""
    rp = np.array([0, 2, 4, 6], np.int64)
    ci = np.array([0, 1, 0, 1, 1, 2], np.int64)
    v = np.array([2.0, -1, -1, 2, -1, 2])
    h = C.c_void_p()
    assert lib.pmg_mcsor_create_csr_idx(3, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64, C.byref(h)) == 0
    n = C.c_int32()
    assert lib.pmg_mcsor_destroy(C.byref(h)) == 0
    assert lib.pmg_mcsor_create_csr_idx(3, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 16, C.byref(h)) == 63
    assert b"idx_width" in lib.pmg_last_error_string()
    bad = ci.copy()
    bad[3] = 1 << 33
    assert lib.pmg_mcsor_create_csr_idx(3, rp.ctypes.data, bad.ctypes.data, v.ctypes.data, 64, C.byref(h)) == 63
    assert not h.value
    rpbad = np.array([0, 4, 2, 6], np.int64)
    assert lib.pmg_mcsor_create_csr_idx(3, rpbad.ctypes.data, ci.ctypes.data, v.ctypes.data, 64, C.byref(h)) == 62
    assert lib.pmg_mcsor_create_csr_idx(1 << 40, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64, C.byref(h)) == 63
    # 32-bit arrays through the same entry point
    assert lib.pmg_mcsor_create_csr_idx(3, rp.astype(np.int32).ctypes.data, ci.astype(np.int32).ctypes.data, v.ctypes.data, 32, C.byref(h)) == 0
    assert lib.pmg_mcsor_destroy(C.byref(h)) == 0
    mg = C.c_void_p()
    assert lib.pmg_mgmc_create_hierarchy(2, C.byref(mg)) == 0
    assert lib.pmg_mgmc_set_level_operator_idx(mg, 1, 3, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64) == 0
    assert lib.pmg_mgmc_set_level_operator_idx(mg, 5, 3, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64) == 63
    prp, pci, pv = np.array([0, 1, 2, 3], np.int64), np.array([0, 0, 7], np.int64), np.ones(3)
    assert lib.pmg_mgmc_set_level_interpolation_idx(mg, 1, 3, 1, prp.ctypes.data, pci.ctypes.data, pv.ctypes.data, 64) == 63  # column 7 of 1
    assert lib.pmg_mgmc_destroy(C.byref(mg)) == 0


def test_trace_ranges_can_be_switched():
    """ROCTx ranges named like the reference's log events (src/parmgmc.c:118-127): on with PMG_TRACE=1, off with 0"""
    import os
    import sys

    code = "import ctypes; L = ctypes.CDLL(%r); print(L.pmg_trace_enabled())" % str(capi.library_path())
    for env, want in (("1", "1"), ("0", "0")):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PMG_TRACE=env))
        assert out.stdout.strip() == want, (env, out.stdout, out.stderr)
