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
FILES = [ADAPTER / "pc_hipgibbs.c", ADAPTER / "pc_hipgamgmc.c", ADAPTER / "pc_hipparsor.c", ADAPTER / "pc_hipwoodbury.c", ADAPTER / "mc_sor_hip.c"]
REFERENCE = Path("/root/reference")  # present in the build container only; never on the GPU box
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


def test_mcsor_binding_and_the_reference_option_keys():
    """round 4: the ten functions of reference include/parmgmc/mc_sor.h:21-30 with their PETSc types (adapter/mc_sor_hip.c),
    -mc_sor_omega (src/mc_sor.c:638), the cholsampler's option keys (src/pc_chols.c:415-417), and an ERROR for the one
    option that has no device form (-pc_sorgibbs_local_forward, src/pc_sorgibbs.c:274)"""
    mc = (ADAPTER / "mc_sor_hip.c").read_text()
    for fn in ("MCSORCreate(Mat A, MCSOR *m)", "MCSORSetUp(MCSOR m)", "MCSORDestroy(MCSOR *m)", "MCSORApply(MCSOR m, Vec b, Vec y)", "MCSORSetOmega(MCSOR m, PetscReal omega)", "MCSORSetSweepType(MCSOR m, MatSORType type)", "MCSORGetSweepType(MCSOR m, MatSORType *type)",
               "MCSORGetISColoring(MCSOR m, ISColoring *isc)", "MCSORGetNumColors(MCSOR m, PetscInt *colors)", "MCSORBuildLRCCorrection(PetscErrorCode (*det_sor)(void *, Vec, Vec), void *ctx, Mat Asor, Mat B, Vec S, Mat *Bb)"):
        assert "PetscErrorCode " + fn in mc, fn
    assert '"-mc_sor_omega"' in mc and "pmg_mcsor_apply(" in mc and "pmg_distmcsor_apply(" in mc and "pmg_rowblock_sampler_create(" in mc
    assert "pmg_mcsor_set_lowrank(" in mc and "pmg_distmcsor_set_lowrank(" in mc and "pmg_mcsor_get_coloring(" in mc and "ISColoringCreate(" in mc
    gibbs = (ADAPTER / "pc_hipgibbs.c").read_text()
    chol = gibbs[gibbs.index("static PetscErrorCode PCSetFromOptions_HipChol"):]
    assert '"-pc_cholsampler_dense_threshold"' in chol and '"-pc_cholsampler_coarse_gamg"' in chol and "pc->ops->setfromoptions  = PCSetFromOptions_HipChol" in chol
    lf = gibbs[gibbs.index('"-pc_sorgibbs_local_forward"'):]
    assert "PETSC_ERR_SUP" in lf[:900]


@pytest.mark.parametrize("idx", ["32", "64"])
def test_mcsor_binding_matches_the_reference_header(idx):
    """adapter/mc_sor_hip.c compiled with the REFERENCE's own <parmgmc/mc_sor.h> and <parmgmc/parmgmc.h> in front of the
    declaration mock: a definition that disagrees with the reference's prototype is a compile error (read in place under
    /root/reference; nothing of it is in this repository)"""
    if not GCC or not (REFERENCE / "include" / "parmgmc" / "mc_sor.h").exists():
        pytest.skip("needs gcc and /root/reference (build container only)")
    extra = ["-DPETSC_DECL_MOCK_64BIT_INDICES"] if idx == "64" else []
    r = subprocess.run([GCC, "-std=gnu11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DPARMGMC_HIP_HAVE_PETSC", *extra, "-I", str(REFERENCE / "include"), *INC, str(ADAPTER / "mc_sor_hip.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("example", ["ex3.c", "ex5.c"])
def test_reference_examples_compile_unchanged_against_the_binding(example):
    """examples/ex3.c (MCSOR inside a PCSHELL: the route BASELINE's north star names) and examples/ex5.c (symmetric =
    forward o backward) of the reference, UNCHANGED, read in place: valid C against the reference's own parmgmc headers +
    the PETSc declaration mock, i.e. every MCSOR* / PETSc call they make has the signature the binding and the mock declare"""
    src = REFERENCE / "examples" / example
    if not GCC or not src.exists():
        pytest.skip("needs gcc and /root/reference (build container only)")
    r = subprocess.run([GCC, "-std=gnu11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", str(REFERENCE / "include"), "-I", str(ROOT / "tests" / "petsc_decl_mock"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    calls = set(re.findall(r"\b(MCSOR[A-Za-z]+)\(", src.read_text()))
    binding = (ADAPTER / "mc_sor_hip.c").read_text()
    assert calls and all(f"PetscErrorCode {c}(" in binding for c in calls), calls


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
