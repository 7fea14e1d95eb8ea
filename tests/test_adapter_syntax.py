"""The PETSc-side constructors (adapter/: PCCreate_HipSORGibbs, PCCreate_HipMulticolorGibbs, PCCreate_HipGAMGMC,
PCCreate_HipCholSampler, PCCreate_HipPARSOR, PCCreate_HipWoodbury -- all six type names the reference registers at
src/parmgmc.c:44-54 -- each with its more-than-one-rank branch: MATMPIAIJ row blocks / DMDA z-slabs) cannot be built
here: the image has no PETSc.  What CAN be checked without it:
  * the files are valid C, for 32- and 64-bit PetscInt, against a declaration-only transcription of the PETSc calls
    they make (tests/petsc_decl_mock: declarations, no definitions -- nothing links, nothing runs);
  * they fill every op the reference's constructors fill and compose the sample-callback setter;
  * every pmg_* function they call is declared in include/parmgmc_hip.h and exported by the library;
  * without -DPARMGMC_HIP_HAVE_PETSC they are empty translation units.
Behaviour against a real PETSc stays untested (stated in DESIGN.md)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ADAPTER = ROOT / "adapter"
FILES = [ADAPTER / "pc_hipgibbs.c", ADAPTER / "pc_hipgamgmc.c", ADAPTER / "pc_hipparsor.c", ADAPTER / "pc_hipwoodbury.c"]
GCC = shutil.which("gcc")
INC = ["-I", str(ROOT / "tests" / "petsc_decl_mock"), "-I", str(ROOT / "include"), "-I", "/opt/rocm/include", "-I", str(ADAPTER), "-D__HIP_PLATFORM_AMD__"]


@pytest.mark.parametrize("idx", ["32", "64"])
@pytest.mark.parametrize("src", FILES, ids=lambda p: p.name)
def test_adapter_is_valid_c_for_both_index_widths(src, idx):
    if not GCC:
        pytest.skip("no gcc")
    extra = ["-DPETSC_DECL_MOCK_64BIT_INDICES"] if idx == "64" else []
    r = subprocess.run([GCC, "-std=gnu11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DPARMGMC_HIP_HAVE_PETSC", *extra, *INC, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("src", FILES, ids=lambda p: p.name)
def test_adapter_is_empty_without_petsc(src, tmp_path):
    if not GCC:
        pytest.skip("no gcc")
    obj = tmp_path / "a.o"
    subprocess.check_call([GCC, "-std=gnu11", "-c", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(src), "-o", str(obj)])
    syms = subprocess.run(["nm", str(obj)], capture_output=True, text=True).stdout
    assert " T " not in syms


def test_constructors_fill_the_ops_of_the_reference_constructors():
    """reference src/pc_sorgibbs.c:306-324, src/pc_mcgibbs.c:305-327, src/pc_gamgmc.c:381-404"""
    gibbs = (ADAPTER / "pc_hipgibbs.c").read_text()
    common = gibbs[gibbs.index("static PetscErrorCode PCCreate_HipGibbsCommon"):gibbs.index("PetscErrorCode PCCreate_HipSORGibbs")]
    for op in ("setup", "applyrichardson", "destroy", "reset", "setfromoptions", "view", "apply"):
        assert f"pc->ops->{op}" in common, op
    assert "PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipGibbs)" in common
    mg = (ADAPTER / "pc_hipgamgmc.c").read_text()
    ctor = mg[mg.index("PetscErrorCode PCCreate_HipGAMGMC"):mg.index("PetscErrorCode ParMGMCHipRegisterPCAll")]
    for op in ("setup", "reset", "applyrichardson", "view", "destroy", "setfromoptions"):
        assert f"pc->ops->{op}" in ctor, op
    assert "PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipGAMGMC)" in ctor and '"PCMGGetLevels_C"' in ctor
    reg = mg[mg.index("PetscErrorCode ParMGMCHipRegisterPCAll"):]
    for name in ("PCSORGIBBS", "PCMCGIBBS", "PCGAMGMC", "PCCHOLSAMPLER", "PCPARSOR", "PCWOODBURY"):  # include/parmgmc/parmgmc.h:26-31
        assert f"PCRegister({name}," in reg
    # the two constructors the round-2 adapter lacked: ops of reference src/pc_parsor.c:1021-1039 and src/woodbury.c:291-302
    par = (ADAPTER / "pc_hipparsor.c").read_text()
    ctor = par[par.index("PetscErrorCode PCCreate_HipPARSOR"):]
    for op in ("apply", "destroy", "reset", "setup", "setfromoptions", "view"):
        assert f"pc->ops->{op}" in ctor, op
    assert "-pc_parsor_omega" in par and "-pc_parsor_its" in par and "pmg_mcsor_set_idiag_by_division" in par
    wo = (ADAPTER / "pc_hipwoodbury.c").read_text()
    ctor = wo[wo.index("PetscErrorCode PCCreate_HipWoodbury"):]
    for op in ("setup", "reset", "destroy", "setfromoptions", "applyrichardson"):
        assert f"pc->ops->{op}" in ctor, op
    assert "PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipWoodbury)" in ctor
    assert '"pc_woodbury_solver_"' in wo and '"pc_woodbury_sampler")' in wo and "-pc_woodbury_solver" in wo and "-pc_woodbury_sampler" in wo
    # every constructor has its more-than-one-rank branch: no PETSC_ERR_SUP on size > 1 any more
    assert "pmg_rowblock_sampler_create" in gibbs and "pmg_distmcsor_sample" in gibbs and "MatMPIAIJGetSeqAIJ" in (ADAPTER / "hip_petsc_common.h").read_text()
    assert "pmg_mgmc_create_dmda_slab" in mg and "pmg_rbh_create_mgmc" in mg and "pmg_rbh_build" in mg
    assert "pmg_distmcsor_apply" in par and "pmg_woodbury_correct" in wo and "pmg_dist_create_comm" in (ADAPTER / "hip_petsc_common.h").read_text()
    # one noise stream per PC instance (reference: one advancing PetscRandom shared by all PCs, src/parmgmc.c:56-68)
    for txt in (gibbs, mg, wo):
        assert "ParMGMCHipNextStreamId()" in txt and "HipNoiseSeed(" in txt
    # option names of the reference
    for opt in ("-pc_mcgibbs_omega", "-pc_mcgibbs_forward", "-pc_mcgibbs_backward", "-pc_mcgibbs_symmetric", "-pc_sorgibbs_forward"):
        assert opt in gibbs
    for opt in ("-pc_gamgmc_mg_type", "-mg_levels_ksp_max_it", "-mg_coarse_pc_type", "-mg_levels_pc_type", "-pc_mg_galerkin"):
        assert opt in mg


def test_adapter_calls_only_exported_c_abi_functions():
    from parmgmc_amd import capi

    declared = set(capi.declared_symbols())
    used = set()
    for f in list(FILES) + [ADAPTER / "hip_petsc_common.h"]:
        used |= set(re.findall(r"\b(pmg_[a-z0-9_]+)\s*\(", f.read_text()))
    assert used and used <= declared, sorted(used - declared)
    assert all(hasattr(capi.lib, s) for s in used)
    # the index arrays cross the boundary with PetscInt's own width
    assert "pmg_mcsor_create_csr_idx" in used and "pmg_mgmc_set_level_operator_idx" in used and "pmg_chol_create_csr_idx" in used
    assert "pmg_rowblock_merge_mpiaij" in used  # MatMPIAIJGetSeqAIJ's blocks cross with PetscInt's width too
