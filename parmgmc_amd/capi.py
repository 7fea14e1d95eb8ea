"""ctypes binding of include/parmgmc_hip.h.  No fallback: a missing library is an ImportError."""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_ROOT = _PKG.parent

SOR_FORWARD_SWEEP, SOR_BACKWARD_SWEEP, SOR_SYMMETRIC_SWEEP = 1, 2, 3
COLORING_GREEDY, COLORING_LEXLEVELS, COLORING_USER, COLORING_ITERATED = 0, 1, 2, 3


class PMGError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pmg status {code}: {msg}")
        self.code = code


def library_path() -> Path:
    import os

    alt = os.environ.get("PMG_LIBRARY")  # development: an A/B build of the library (tools/ab_build.sh)
    return Path(alt) if alt else _PKG / "libparmgmc_hip.so"


def header_path() -> Path:
    return _ROOT / "include" / "parmgmc_hip.h"


def declared_symbols() -> list[str]:
    """Every function name the public header declares (used by the export test)."""
    text = header_path().read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmg_[a-z0-9_]+)\s*\((?!\*)", text)))


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname without
    version), libparmgmc_hip.so names the system one (libamdhip64.so.7): loaded naively the process gets two
    runtimes and whichever initialises second sees "no ROCm-capable device".  Callers hand us torch tensors,
    so make torch's runtime the process-global one BEFORE our library is mapped; the dynamic linker then binds
    our hip* references to it (global scope is searched before a library's own dependencies)."""
    try:
        import torch  # noqa: F401
    except Exception:
        return
    cand = Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def torch_rccl_path() -> str | None:
    """The librccl.so PyTorch bundles (so that a process beside torch keeps ONE RCCL), or None."""
    try:
        import torch
    except Exception:
        return None
    cand = Path(torch.__file__).resolve().parent / "lib" / "librccl.so"
    return str(cand) if cand.exists() else None


def _load() -> C.CDLL:
    _share_torch_hip_runtime()
    p = library_path()
    if not p.exists():
        raise ImportError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` " "(hipcc --offload-arch=gfx950). parmgmc_amd has no CPU fallback.")
    return C.CDLL(str(p))


lib = _load()

_vp, _i32, _i64, _u64, _dbl, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_int
_sig = {
    "pmg_last_error_string": (C.c_char_p, []),
    "pmg_version": (C.c_char_p, []),
    "pmg_trace_enabled": (_int, []),
    "pmg_gpu_arch": (C.c_char_p, []),
    "pmg_mcsor_create_csr": (_int, [_i32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pmg_mcsor_create_csr_idx": (_int, [_i64, _vp, _vp, _vp, _int, C.POINTER(_vp)]),
    "pmg_mgmc_set_level_operator_idx": (_int, [_vp, _i32, _i64, _vp, _vp, _vp, _int]),
    "pmg_mgmc_set_level_interpolation_idx": (_int, [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _int]),
    "pmg_chol_create_csr_idx": (_int, [_i64, _vp, _vp, _vp, _int, _i32, _vp, _vp, C.POINTER(_vp)]),
    "pmg_mcsor_set_coloring": (_int, [_vp, _int, _vp]),
    "pmg_mcsor_setup": (_int, [_vp]),
    "pmg_mcsor_set_omega": (_int, [_vp, _dbl]),
    "pmg_mcsor_set_sweep_type": (_int, [_vp, _int]),
    "pmg_mcsor_get_sweep_type": (_int, [_vp, C.POINTER(_int)]),
    "pmg_mcsor_get_num_colors": (_int, [_vp, C.POINTER(_i32)]),
    "pmg_mcsor_get_coloring": (_int, [_vp, _vp]),
    "pmg_mcsor_apply": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_mcsor_sample": (_int, [_vp, _vp, _vp, _i32, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_mcsor_residual": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "pmg_mcsor_layout_len": (_int, [_vp, C.POINTER(_i32)]),
    "pmg_mcsor_get_layout": (_int, [_vp, _vp]),
    "pmg_mcsor_to_layout": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_mcsor_from_layout": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_mcsor_apply_layout": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_mcsor_sample_layout": (_int, [_vp, _vp, _vp, _i32, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_mcsor_residual_layout": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "pmg_mcsor_set_noise_row_offset": (_int, [_vp, C.c_int64]),
    "pmg_mcsor_sweep_color_layout": (_int, [_vp, _i32, _int, _int, _u64, _u64, _vp, _vp, _vp]),
    "pmg_mcsor_set_lowrank": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_mcsor_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_grid_create": (_int, [_i32, _i32, _i32, _i32, _i32, _dbl, C.POINTER(_vp)]),
    "pmg_grid_set_omega": (_int, [_vp, _dbl]),
    "pmg_grid_set_sweep_type": (_int, [_vp, _int]),
    "pmg_grid_get_sweep_type": (_int, [_vp, C.POINTER(_int)]),
    "pmg_grid_get_num_colors": (_int, [_vp, C.POINTER(_i32)]),
    "pmg_grid_get_coloring": (_int, [_vp, _vp]),
    "pmg_grid_cvec_len": (_int, [_vp, C.POINTER(_i64)]),
    "pmg_grid_get_layout": (_int, [_vp, _vp]),
    "pmg_grid_to_cvec": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_grid_from_cvec": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_grid_apply": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_grid_apply_cvec": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_grid_sample_cvec": (_int, [_vp, _vp, _vp, _i32, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_grid_residual_cvec": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "pmg_grid_sweep_color_cvec": (_int, [_vp, _int, _int, _int, _u64, _u64, _vp, _vp, _vp]),
    "pmg_grid_sweep_color_planes_cvec": (_int, [_vp, _int, _i32, _i32, _int, _int, _u64, _u64, _vp, _vp, _vp]),
    "pmg_grid_halo_plane": (_int, [_vp, _int, _int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "pmg_grid_set_lowrank": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_grid_sample": (_int, [_vp, _vp, _vp, _i32, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_grid_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_dist_get_unique_id": (_int, [C.c_char_p, _vp]),
    "pmg_dist_create": (_int, [_vp, _i32, _i32, _vp, C.c_char_p, _int, C.POINTER(_vp)]),
    "pmg_dist_create_ipc": (_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "pmg_dist_ipc_blob_bytes": (_int, [C.POINTER(_i32)]),
    "pmg_dist_ipc_export": (_int, [_vp, _vp]),
    "pmg_dist_ipc_connect": (_int, [_vp, _vp, _vp]),
    "pmg_dist_ipc_connect_loopback": (_int, [_vp]),
    "pmg_dist_sample_cvec": (_int, [_vp, _vp, _vp, _i32, _int, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_dist_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_dist_ipc_disconnect": (_int, [_vp]),
    "pmg_distmcsor_create": (_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pmg_distmcsor_sample_layout": (_int, [_vp, _vp, _vp, _i32, _int, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_distmcsor_apply_layout": (_int, [_vp, _vp, _vp, _int, _vp]),
    "pmg_distmcsor_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_chol_create_csr": (_int, [_i32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pmg_chol_create_csr_lowrank": (_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, C.POINTER(_vp)]),
    "pmg_chol_get_factor": (_int, [_vp, _vp]),
    "pmg_chol_sample": (_int, [_vp, _vp, _vp, _int, _u64, _u64, _vp]),
    "pmg_chol_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_mgmc_create_dmda": (_int, [_i32, _i32, _i32, _dbl, _i32, C.POINTER(_vp)]),
    "pmg_mgmc_create_hierarchy": (_int, [_i32, C.POINTER(_vp)]),
    "pmg_mgmc_set_level_operator": (_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "pmg_mgmc_set_level_interpolation": (_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pmg_mgmc_set_smoother": (_int, [_vp, _int, _dbl, _int, _i32]),
    "pmg_mgmc_set_coarse": (_int, [_vp, _int, _i32]),
    "pmg_mgmc_set_keep_host": (_int, [_vp, _int]),
    "pmg_mgmc_create_dmda_slab": (_int, [_i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pmg_dist_exchange": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmg_dist_allgather": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "pmg_dist_ipc_connect_all": (_int, [_vp, _vp]),
    "pmg_dist_apply_cvec": (_int, [_vp, _vp, _vp, _int, _vp]),
    "pmg_dist_allreduce_sum": (_int, [_vp, _vp, _i32, _vp]),
    "pmg_dist_check": (_int, [_vp]),
    "pmg_dist_get_info": (_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_int64)]),
    "pmg_dist_describe": (_int, [_vp, _i32, _i32, _vp]),
    "pmg_mgmc_set_lowrank": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_mgmc_set_correction_form": (_int, [_vp, _int]),
    "pmg_mgmc_set_fused_transfers": (_int, [_vp, _int]),
    "pmg_mgmc_set_coloring": (_int, [_vp, _int]),
    "pmg_mgmc_get_algorithmic_bytes": (_int, [_vp, C.POINTER(_dbl), _vp]),
    "pmg_mgmc_level_lowrank_factors": (_int, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i64), _vp, _vp, _vp, _vp]),
    "pmg_mgmc_level_lowrank_post": (_int, [_vp, _i32, _int, _vp, _vp]),
    "pmg_mgmc_level_lowrank_residual_sub": (_int, [_vp, _i32, _int, _vp, _vp, _vp]),
    "pmg_mgmc_setup": (_int, [_vp]),
    "pmg_mgmc_get_num_levels": (_int, [_vp, C.POINTER(_i32)]),
    "pmg_mgmc_get_level_dims": (_int, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "pmg_mgmc_get_level_matrix": (_int, [_vp, _i32, _int, C.POINTER(_i32), C.POINTER(_i32), _vp, _vp, _vp]),
    "pmg_mgmc_sample": (_int, [_vp, _vp, _vp, _i32, _int, _u64, _u64, C.POINTER(_u64), _vp, _vp, _vp]),
    "pmg_mgmc_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_mgmc_get_level_layout": (_int, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "pmg_mgmc_get_level_stencil": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_mgmc_level_sweep": (_int, [_vp, _i32, _int, _int, _u64, _u64, _vp, _vp, _vp]),
    "pmg_mgmc_level_residual": (_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "pmg_mgmc_level_restrict": (_int, [_vp, _i32, _vp, _vp, _vp]),
    "pmg_mgmc_set_rowblock_transport": (_int, [_vp, _vp, _vp]),
    "pmg_mgmc_set_level_rowblock": (_int, [_vp, _i32, C.c_int64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmg_mgmc_set_level_restriction": (_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pmg_distmcsor_set_lowrank": (_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "pmg_distmcsor_set_lowrank_dev": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_distmcsor_residual_layout": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "pmg_distmcsor_refresh_layout": (_int, [_vp, _vp, _vp]),
    "pmg_mgmc_level_residual_restrict": (_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "pmg_mgmc_level_prolong_add": (_int, [_vp, _i32, _vp, _vp, _vp]),
    "pmg_initialize": (_int, []),
    "pmg_finalize": (_int, []),
    "pmg_pc_register": (_int, [C.c_char_p, _vp]),
    "pmg_set_seed": (_int, [_u64]),
    "pmg_options_set_value": (_int, [C.c_char_p, C.c_char_p]),
    "pmg_options_clear": (_int, []),
    "pmg_mat_create_csr": (_int, [_i32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pmg_mat_create_dmda": (_int, [_i32, _i32, _i32, _dbl, C.POINTER(_vp)]),
    "pmg_make_observation_mats_dmda": (_int, [_i32, _i32, _i32, _i32, _i32, _i32, _dbl, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmg_autocorrelation": (_int, [C.c_int64, _vp, _vp]),
    "pmg_stream_triad": (_int, [_i64, _vp, _vp, _vp, _vp]),
    "pmg_iact": (_int, [C.c_int64, _vp, C.POINTER(_dbl), _vp, C.POINTER(_int)]),
    "pmg_estimate_covariance_errors": (_int, [_i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "pmg_pc_woodbury_set_solver": (_int, [_vp, _vp]),
    "pmg_pc_woodbury_set_sampler": (_int, [_vp, _vp]),
    "pmg_mat_create_lrc": (_int, [_vp, _i32, _vp, _vp, C.POINTER(_vp)]),
    "pmg_mat_get_size": (_int, [_vp, C.POINTER(_i32)]),
    "pmg_mat_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_pc_create": (_int, [C.POINTER(_vp)]),
    "pmg_pc_set_type": (_int, [_vp, C.c_char_p]),
    "pmg_pc_get_type": (_int, [_vp, C.c_char_p, _i32]),
    "pmg_pc_set_options_prefix": (_int, [_vp, C.c_char_p]),
    "pmg_pc_set_operators": (_int, [_vp, _vp]),
    "pmg_pc_set_from_options": (_int, [_vp]),
    "pmg_pc_setup": (_int, [_vp]),
    "pmg_pc_view": (_int, [_vp, C.c_char_p, _i32]),
    "pmg_pc_reset": (_int, [_vp]),
    "pmg_pc_destroy": (_int, [C.POINTER(_vp)]),
    "pmg_pc_apply": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_pc_apply_richardson": (_int, [_vp, _vp, _vp, _i32, _int, C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "pmg_ksp_richardson_solve": (_int, [_vp, _vp, _vp, _i32, _int, _vp]),
    "pmg_pc_set_sample_callback": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_pc_get_noise_state": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "pmg_pc_set_noise_counter": (_int, [_vp, _u64]),
    "pmg_pc_mcgibbs_set_omega": (_int, [_vp, _dbl]),
    "pmg_pc_mcgibbs_set_sweep_type": (_int, [_vp, _int]),
    "pmg_pc_parsor_set_omega": (_int, [_vp, _dbl]),
    "pmg_pc_parsor_set_iterations": (_int, [_vp, _i32]),
    "pmg_pc_parsor_apply_sor": (_int, [_vp, _vp, _i32, _int, _vp, _vp]),
    "pmg_pc_parsor_set_partition": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_pc_parsor_get_partition_info": (_int, [_vp, _vp, _vp, _vp]),
    "pmg_pc_gamgmc_set_levels": (_int, [_vp, _i32]),
    "pmg_pc_shell_set_apply": (_int, [_vp, _vp]),
    "pmg_pc_shell_set_context": (_int, [_vp, _vp]),
    "pmg_pc_shell_get_context": (_int, [_vp, C.POINTER(_vp)]),
    "pmg_vec_set_random_standard_normal": (_int, [_i64, _vp, _u64, _u64, _vp]),
    # row-block set-up in C (pmg_rowblock.c); the pmg_host_comm argument is a pointer to HostComm
    "pmg_rowblock_merge_mpiaij": (_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    "pmg_rowblock_color_greedy": (_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "pmg_rowblock_color_iterated": (_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "pmg_rowblock_check_coloring": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "pmg_rowblock_plan_create": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _vp, C.POINTER(_vp)]),
    "pmg_rowblock_plan_get": (_int, [_vp, C.POINTER(_i32)] + [C.POINTER(_vp)] * 7),
    "pmg_rowblock_plan_destroy": (None, [C.POINTER(_vp)]),
    "pmg_rbh_create": (_int, [_vp, _i32, _i64, C.POINTER(_vp)]),
    "pmg_rbh_set_level_operator": (_int, [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _int]),
    "pmg_rbh_set_level_interpolation": (_int, [_vp, _i32, _i64, _vp, _vp, _vp, _int]),
    "pmg_rbh_set_level_coloring": (_int, [_vp, _i32, _i32, _vp]),
    "pmg_rbh_set_coloring": (_int, [_vp, _int]),
    "pmg_rbh_build": (_int, [_vp]),
    "pmg_rbh_get_info": (_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "pmg_rbh_get_level": (_int, [_vp, _i32, _vp]),
    "pmg_rbh_create_mgmc": (_int, [_vp, _vp, C.POINTER(_vp)]),
    "pmg_rbh_destroy": (None, [C.POINTER(_vp)]),
    "pmg_rowblock_sampler_create": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _i32, _vp, _dbl, C.POINTER(_vp), C.POINTER(_vp)]),
    "pmg_dist_create_comm": (_int, [_vp, C.c_char_p, _vp, C.c_char_p, C.POINTER(_vp)]),
    "pmg_dist_destroy_comm": (_int, [_vp, C.POINTER(_vp)]),
    "pmg_mcsor_get_size": (_int, [_vp, C.POINTER(_i32)]),
    "pmg_mcsor_set_idiag_by_division": (_int, [_vp, _int]),
    "pmg_distmcsor_sample": (_int, [_vp, _i32, _vp, _vp, _i32, _int, _int, _u64, _u64, C.POINTER(_u64), _vp]),
    "pmg_distmcsor_apply": (_int, [_vp, _i32, _vp, _vp, _int, _vp]),
    "pmg_woodbury_create": (_int, [_i64, _i32, _vp, _i64, _vp, _vp, C.POINTER(_vp)]),
    "pmg_woodbury_column": (_int, [_vp, _i32, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "pmg_woodbury_set_c_column": (_int, [_vp, _i32, _vp, _vp]),
    "pmg_woodbury_finish": (_int, [_vp]),
    "pmg_woodbury_noisy_rhs": (_int, [_vp, _vp, _vp, _u64, _u64, _vp]),
    "pmg_woodbury_correct": (_int, [_vp, _vp, _vp]),
    "pmg_woodbury_get_correction": (_int, [_vp, _vp]),
    "pmg_woodbury_destroy": (_int, [C.POINTER(_vp)]),
}
for _name, (_res, _args) in _sig.items():
    _f = getattr(lib, _name)
    _f.restype, _f.argtypes = _res, _args


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class HostComm(C.Structure):
    """pmg_host_comm: the byte all-gather the C set-up calls back into"""

    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("allgather", ALLGATHER_FN), ("ctx", C.c_void_p)]


class DistDescription(C.Structure):
    """pmg_dist_description"""

    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("device", C.c_int32), ("neighbour", C.c_int32 * 2), ("peer_access", C.c_int32 * 2), ("rccl_comm_count", C.c_int32), ("halo_wait_polls", C.c_uint64), ("pci_bus_id", C.c_char * 32), ("transport", C.c_char * 8)]

    def as_dict(self):
        return {"rank": self.rank, "nranks": self.nranks, "device": self.device, "pci_bus_id": self.pci_bus_id.decode(errors="replace"), "neighbour_ranks": list(self.neighbour), "peer_access_lo_hi": list(self.peer_access),
                "transport": self.transport.decode(errors="replace"), "rccl_comm_count": self.rccl_comm_count, "halo_wait_polls": int(self.halo_wait_polls)}


class RbhLevelView(C.Structure):
    """pmg_rbh_level_view"""

    _fields_ = [("n_global", C.c_int64), ("row0", C.c_int64), ("starts", C.POINTER(C.c_int64))] + [(n_, C.c_int32) for n_ in ("replicated", "nowned", "nlocal", "nghost", "ncolors", "P_nrows", "R_nrows", "ncoarse_local")] + [(n_, C.POINTER(C.c_int32)) for n_ in ("rp", "ci", "colors", "P_rp", "P_ci", "R_rp", "R_ci", "send_rows", "recv_src", "recv_rows")] + [(n_, C.POINTER(C.c_double)) for n_ in ("v", "P_v", "R_v")] + [(n_, C.POINTER(C.c_int64)) for n_ in ("ghosts", "send_ptr", "counts", "recv_ptr")]


def torch_host_comm(rank: int, world: int, group=None):
    """A pmg_host_comm whose all-gather is torch.distributed's (byte tensors on the CPU for gloo, staged through the
    device for nccl).  Returns (HostComm, keep-alive callback object): hold both while the C side may call back."""
    import numpy as np

    def _ag(_ctx, send, nbytes, recv):
        try:
            import torch
            import torch.distributed as dist

            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[: int(nbytes)]
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.from_numpy(src.copy()).to(dev)
            out = torch.empty(world * int(nbytes), dtype=torch.uint8, device=dev)
            if nbytes:
                if dev == "cuda":
                    dist.all_gather_into_tensor(out, t, group=group)
                else:
                    parts = [torch.empty(int(nbytes), dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, t, group=group)
                    out = torch.cat(parts)
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(world * int(nbytes),))
                dst[:] = out.cpu().numpy()
            else:
                dist.barrier(group=group)
            return 0
        except Exception:  # pragma: no cover
            import traceback

            traceback.print_exc()
            return 1

    cb = ALLGATHER_FN(_ag)
    return HostComm(rank, world, cb, None), cb


DELETER = C.CFUNCTYPE(C.c_int, C.c_void_p)
SHELL_APPLY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
PC_CTOR = C.CFUNCTYPE(C.c_int, C.c_void_p)
SAMPLE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p)


def check(status: int) -> None:
    """PetscCall analogue: raise on a non-zero status with the library's message."""
    if status != 0:
        raise PMGError(status, lib.pmg_last_error_string().decode())
