/*
 * pc_hipgamgmc.c -- PCCreate_HipGAMGMC: the Multigrid Monte Carlo sampler of ParMGMC with the V-cycle on an MI355X.
 *
 * Replaces PCCreate_GAMGMC (reference src/pc_gamgmc.c:381-404).  Like the reference it keeps an inner PETSc PC
 * (PCGAMG by default, PCMG with -pc_gamgmc_mg_type mg) whose ONLY job is to build the hierarchy -- aggregation or
 * DMDA coarsening, interpolations, Galerkin products -- with the same options the reference injects (src/pc_gamgmc.c:
 * 299-350).  What the reference then runs per sample through PCApply(pg->mg, ...) (src/pc_gamgmc.c:246,255: smoother
 * sweeps, residuals, MatRestrict / MatInterpolateAdd, the coarse Cholesky sample) runs here inside libparmgmc_hip:
 * after PCSetUp(mg) every level operator and interpolation is handed over exactly where PCGAMGMC_SetUpHierarchy reads
 * them (src/pc_gamgmc.c:165-176), and PCApplyRichardson forwards to pmg_mgmc_sample (src/pc_gamgmc.c:227-264).
 *
 * Built only inside a ParMGMC + PETSc tree with -DPARMGMC_HIP_HAVE_PETSC; empty otherwise.
 */
#ifdef PARMGMC_HIP_HAVE_PETSC
#include "hip_petsc_common.h"
#include <string.h>

typedef struct {
  PC          mg; /* PETSc's hierarchy builder; never applied */
  char        mgtype[64];
  pmg_mgmc    h;
  uint64_t    seed, counter;
  HipStageBuf bbuf, ybuf;

  void *cbctx;
  PetscErrorCode (*scb)(PetscInt, Vec, void *);
  PetscErrorCode (*del_scb)(void *);
} PC_HipGAMGMC;

/* context of the library's per-sample callback: gives the sample to the user's PETSc callback as a Vec */
typedef struct {
  PC_HipGAMGMC  *pg;
  HipVecAccess  *y;
  PetscErrorCode ierr;
} HipTrampoline;

static int HipSampleTrampoline(int32_t it, const double *y_nat_dev, int32_t n, void *ctx)
{
  HipTrampoline *t = (HipTrampoline *)ctx;
  (void)y_nat_dev; /* == t->y->dev: pmg_mgmc_sample writes the sample into the caller's y before it calls back */
  (void)n;
  t->ierr = HipCallSampleCallback(t->pg->scb, t->pg->cbctx, (PetscInt)it, t->y, NULL); /* pg->scb(it, y, ctx), src/pc_gamgmc.c:258 */
  return t->ierr ? (int)t->ierr : 0;
}

static PetscErrorCode HipGAMGMCRelease(PC_HipGAMGMC *pg)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_mgmc_destroy(&pg->h));
  PetscCall(HipStageBufFree(&pg->bbuf));
  PetscCall(HipStageBufFree(&pg->ybuf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCReset_HipGAMGMC(PC pc) /* src/pc_gamgmc.c:98-114 */
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(HipGAMGMCRelease(pg));
  PetscCall(PCReset(pg->mg));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_HipGAMGMC(PC pc) /* src/pc_gamgmc.c:78-96 */
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(HipGAMGMCRelease(pg));
  if (pg->scb && pg->del_scb) PetscCall(pg->del_scb(pg->cbctx));
  PetscCall(PCDestroy(&pg->mg));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCSetSampleCallback_C", NULL));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCMGGetLevels_C", NULL));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* "-<prefix><name>" is set to `value` unless the user gave it: the way the reference injects its defaults
   (src/pc_gamgmc.c:305-349; PetscOptionsSetValue ignores the prefix stack, hence the full name) */
static PetscErrorCode HipDefaultOption(const char *prefix, const char *name, const char *value)
{
  PetscBool flag;
  char      opt[512];

  PetscFunctionBeginUser;
  PetscCall(PetscOptionsHasName(NULL, prefix, name, &flag));
  if (!flag) {
    PetscCall(PetscSNPrintf(opt, sizeof(opt), "-%s%s", prefix ? prefix : "", name + 1));
    PetscCall(PetscOptionsSetValue(NULL, opt, value));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* CSR of a sequential AIJ matrix for the library (borrowed until pmg_mgmc_setup) */
static PetscErrorCode HipSeqAIJArrays(Mat A, const char *what, PetscInt l, const PetscInt **ia, const PetscInt **ja, PetscScalar **aa)
{
  PetscBool isseq;

  PetscFunctionBeginUser;
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATSEQAIJ, &isseq));
  PetscCheck(isseq, PetscObjectComm((PetscObject)A), PETSC_ERR_SUP, "%s of level %" PetscInt_FMT " is not MATSEQAIJ: one rank per device hands over sequential matrices (a DMDA split in z uses pmg_mgmc_create_dmda_slab, INTEGRATION.md)", what, l);
  PetscCall(MatSeqAIJGetCSRAndMemType(A, ia, ja, aa, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCSetUp_GAMGMC + PCGAMGMC_SetUpHierarchy (src/pc_gamgmc.c:275-356, :145-225) */
static PetscErrorCode PCSetUp_HipGAMGMC(PC pc)
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;
  Mat           P;
  PetscBool     islrc, flag;
  const char   *prefix;
  PetscInt      levels, nu = 1, coarse_its = 1;
  char          lvl_type[64] = PCSORGIBBS, coarse_type[64] = PCCHOLSAMPLER;
  PetscReal     omega = 1.0;
  MatSORType    sweep = SOR_FORWARD_SWEEP;

  PetscFunctionBeginUser;
  PMGCall(pmg_mgmc_destroy(&pg->h));
  PetscCall(PCSetType(pg->mg, pg->mgtype));
  PetscCall(PCGetOptionsPrefix(pc, &prefix));
  PetscCall(PCSetOptionsPrefix(pg->mg, prefix));
  PetscCall(PCAppendOptionsPrefix(pg->mg, "gamgmc_"));
  PetscCall(PetscObjectTypeCompare((PetscObject)pc->pmat, MATLRC, &islrc));
  if (islrc) PetscCall(MatLRCGetMats(pc->pmat, &P, NULL, NULL, NULL)); /* the hierarchy is built from the base matrix, src/pc_gamgmc.c:282-286 */
  else P = pc->pmat;
  PetscCall(PCSetOperators(pg->mg, P, P));
  if (strcmp(pg->mgtype, PCMG) == 0) PetscCall(PCSetDM(pg->mg, pc->dm));

  /* the defaults PCGAMGMC injects, under the same names, so that -help / -ksp_view / user overrides behave alike */
  PetscCall(PCGetOptionsPrefix(pg->mg, &prefix));
  PetscCall(HipDefaultOption(prefix, "-mg_levels_ksp_type", KSPRICHARDSON));
  PetscCall(HipDefaultOption(prefix, "-mg_coarse_ksp_type", KSPRICHARDSON));
  PetscCall(HipDefaultOption(prefix, "-mg_levels_ksp_max_it", "1"));
  PetscCall(HipDefaultOption(prefix, "-mg_coarse_ksp_max_it", "1"));
  PetscCall(HipDefaultOption(prefix, "-mg_levels_pc_type", PCSORGIBBS));
  PetscCall(HipDefaultOption(prefix, "-mg_coarse_pc_type", PCCHOLSAMPLER));
  PetscCall(HipDefaultOption(prefix, "-pc_mg_galerkin", "both")); /* MGMC needs Galerkin coarse operators, src/pc_gamgmc.c:344-349 */
  PetscCall(PCSetFromOptions(pg->mg));
  PetscCall(PCSetUp(pg->mg));

  /* what the level / coarse options select: read back from the database the inner PC was configured from */
  PetscCall(PetscOptionsGetString(NULL, prefix, "-mg_levels_pc_type", lvl_type, sizeof(lvl_type), NULL));
  PetscCall(PetscOptionsGetString(NULL, prefix, "-mg_coarse_pc_type", coarse_type, sizeof(coarse_type), NULL));
  PetscCall(PetscOptionsGetInt(NULL, prefix, "-mg_levels_ksp_max_it", &nu, NULL));
  PetscCall(PetscOptionsGetInt(NULL, prefix, "-mg_coarse_ksp_max_it", &coarse_its, NULL));
  PetscCall(PetscOptionsGetReal(NULL, prefix, "-mg_levels_pc_mcgibbs_omega", &omega, NULL));
  flag = PETSC_FALSE;
  PetscCall(PetscOptionsGetBool(NULL, prefix, "-mg_levels_pc_mcgibbs_backward", &flag, NULL));
  if (flag) sweep = SOR_BACKWARD_SWEEP;
  flag = PETSC_FALSE;
  PetscCall(PetscOptionsGetBool(NULL, prefix, "-mg_levels_pc_mcgibbs_symmetric", &flag, NULL));
  if (flag) sweep = SOR_SYMMETRIC_SWEEP;
  const PetscBool lvl_mc = (PetscBool)(strcmp(lvl_type, PCMCGIBBS) == 0), lvl_sor = (PetscBool)(strcmp(lvl_type, PCSORGIBBS) == 0);
  const PetscBool c_chol = (PetscBool)(strcmp(coarse_type, PCCHOLSAMPLER) == 0), c_gibbs = (PetscBool)(strcmp(coarse_type, PCSORGIBBS) == 0 || strcmp(coarse_type, PCMCGIBBS) == 0);
  PetscCheck(lvl_mc || lvl_sor, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "level sampler '%s': the device V-cycle runs sorgibbs or mcgibbs on the levels", lvl_type);
  PetscCheck(c_chol || c_gibbs, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "coarse sampler '%s': the device V-cycle runs cholsampler, sorgibbs or mcgibbs on the coarsest level", coarse_type);

  /* hand the hierarchy over, level 0 = coarsest as in PCMG (src/pc_gamgmc.c:165-176) */
  PetscCall(PCMGGetLevels(pg->mg, &levels));
  PMGCall(pmg_mgmc_create_hierarchy((int32_t)levels, &pg->h));
  for (PetscInt l = 0; l < levels; ++l) {
    KSP             ksp;
    PC              pcl;
    Mat             A, Ip;
    const PetscInt *ia, *ja;
    PetscScalar    *aa;
    PetscInt        n, m;

    PetscCall(PCMGGetSmoother(pg->mg, l, &ksp));
    PetscCall(KSPGetPC(ksp, &pcl));
    PetscCall(PCGetOperators(pcl, NULL, &A));
    PetscCall(MatGetSize(A, &n, NULL));
    PetscCall(HipSeqAIJArrays(A, "operator", l, &ia, &ja, &aa));
    PMGCall(pmg_mgmc_set_level_operator_idx(pg->h, (int32_t)l, (int64_t)n, ia, ja, aa, PMG_IDX_WIDTH));
    if (l > 0) {
      PetscCall(PCMGGetInterpolation(pg->mg, l, &Ip));
      PetscCall(MatGetSize(Ip, &n, &m));
      PetscCall(HipSeqAIJArrays(Ip, "interpolation", l, &ia, &ja, &aa));
      PMGCall(pmg_mgmc_set_level_interpolation_idx(pg->h, (int32_t)l, (int64_t)n, (int64_t)m, ia, ja, aa, PMG_IDX_WIDTH));
    }
    /* the level sampler PETSc created from -mg_levels_pc_type holds its own copy of A_l (on the device too if it is one
       of the constructors of pc_hipgibbs.c) and is never applied: release it */
    PetscCall(PCReset(pcl));
  }
  PMGCall(pmg_mgmc_set_smoother(pg->h, (int)lvl_mc, lvl_mc ? omega : 1.0, (int)sweep, (int32_t)nu));
  PMGCall(pmg_mgmc_set_coarse(pg->h, c_chol ? 0 : 1, (int32_t)coarse_its));
  if (islrc) { /* every level gets A_l + B_l S B_l^T with B_{l-1} = P_l^T B_l, src/pc_gamgmc.c:157-196 */
    Mat                Abase, Bmat;
    Vec                S;
    PetscInt           k;
    const PetscScalar *B, *Sarr;
    PetscScalar       *Bcopy;

    PetscCall(HipGetLRC(pc->pmat, &Abase, &k, &B, &Bcopy, &Bmat, &S));
    PetscCall(VecGetArrayRead(S, &Sarr));
    PMGCall(pmg_mgmc_set_lowrank(pg->h, (int32_t)k, Bcopy ? Bcopy : B, Sarr));
    PetscCall(VecRestoreArrayRead(S, &Sarr));
    PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
    PetscCall(PetscFree(Bcopy));
  }
  PMGCall(pmg_mgmc_setup(pg->h)); /* colours and uploads every level; the PETSc matrices are no longer read afterwards */
  PetscCall(HipNoiseSeed(&pg->seed));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCApplyRichardson_GAMGMC (src/pc_gamgmc.c:227-264): y = MG(b) from a zero guess, then y += MG(b - A y) */
static PetscErrorCode PCApplyRichardson_HipGAMGMC(PC pc, Vec b, Vec y, Vec w, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt its, PetscBool guesszero, PetscInt *outits, PCRichardsonConvergedReason *reason)
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;
  HipVecAccess  ab, ay;
  HipTrampoline tr;
  (void)w;
  (void)rtol;
  (void)abstol;
  (void)dtol;

  PetscFunctionBeginUser;
  PetscCall(HipVecGet(b, PETSC_FALSE, &pg->bbuf, &ab));
  PetscCall(HipVecGet(y, PETSC_TRUE, &pg->ybuf, &ay));
  tr.pg   = pg;
  tr.y    = &ay;
  tr.ierr = PETSC_SUCCESS;
  {
    const int rc = pmg_mgmc_sample(pg->h, ab.dev, ay.dev, (int32_t)its, (int)guesszero, pg->seed, pg->counter, &pg->counter, pg->scb ? HipSampleTrampoline : NULL, &tr, NULL);
    PetscCall(tr.ierr); /* an error raised inside the user's callback keeps its own stack */
    PMGCall(rc);
  }
  PetscCall(HipVecRestore(&ay, NULL));
  PetscCall(HipVecRestore(&ab, NULL));
  *outits = its;
  *reason = PCRICHARDSON_CONVERGED_ITS;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCView_HipGAMGMC(PC pc, PetscViewer v) /* src/pc_gamgmc.c:266-273 */
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(PetscViewerASCIIPrintf(v, "V-cycle on the device: libparmgmc_hip %s (%s); hierarchy built by the PC below, which is not applied\n", pmg_version(), pmg_gpu_arch()));
  PetscCall(PCView(pg->mg, v));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetFromOptions_HipGAMGMC(PC pc, PetscOptionItems_ARG PetscOptionsObject) /* src/pc_gamgmc.c:358-367 */
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  PetscOptionsHeadBegin(PetscOptionsObject, "PCGAMGMC options");
  PetscCall(PetscOptionsString("-pc_gamgmc_mg_type", "The type of the inner multigrid method", NULL, pg->mgtype, pg->mgtype, sizeof(pg->mgtype), NULL));
  PetscOptionsHeadEnd();
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetSampleCallback_HipGAMGMC(PC pc, PetscErrorCode (*cb)(PetscInt, Vec, void *), void *ctx, PetscErrorCode (*deleter)(void *)) /* src/pc_gamgmc.c:369-379 */
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  if (pg->scb && pg->del_scb) PetscCall(pg->del_scb(pg->cbctx));
  PetscCheck(cb, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "Must pass callback function");
  pg->scb = cb;
  if (ctx) pg->cbctx = ctx;
  if (deleter) pg->del_scb = deleter;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCHipGAMGMCGetLevels(PC pc, PetscInt *levels)
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(PCMGGetLevels(pg->mg, levels));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCGAMGMCGetInternalPC / SetLevels (src/pc_gamgmc.c:116-143) for this constructor's data */
PetscErrorCode PCHipGAMGMCGetInternalPC(PC pc, PC *mg)
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  if (mg) *mg = pg->mg;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipGAMGMCSetLevels(PC pc, PetscInt levels)
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(PCMGSetLevels(pg->mg, levels, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_HipGAMGMC(PC pc)
{
  PC_HipGAMGMC *pg;

  PetscFunctionBeginUser;
  PetscCall(PetscNew(&pg));
  PetscCall(PCCreate(PetscObjectComm((PetscObject)pc), &pg->mg));
  PetscCall(PetscStrncpy(pg->mgtype, PCGAMG, sizeof(pg->mgtype))); /* the reference's default, src/pc_gamgmc.c:388 */

  pc->data                 = pg;
  pc->ops->setup           = PCSetUp_HipGAMGMC;
  pc->ops->reset           = PCReset_HipGAMGMC;
  pc->ops->applyrichardson = PCApplyRichardson_HipGAMGMC;
  pc->ops->view            = PCView_HipGAMGMC;
  pc->ops->destroy         = PCDestroy_HipGAMGMC;
  pc->ops->setfromoptions  = PCSetFromOptions_HipGAMGMC;
  PetscCall(PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipGAMGMC));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCMGGetLevels_C", PCHipGAMGMCGetLevels));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- registration: the body of ParMGMCRegisterPCAll (reference src/parmgmc.c:44-54) for the four device samplers ---- */
PetscErrorCode ParMGMCHipRegisterPCAll(void)
{
  PetscFunctionBeginUser;
  PetscCall(PCRegister(PCSORGIBBS, PCCreate_HipSORGibbs));
  PetscCall(PCRegister(PCMCGIBBS, PCCreate_HipMulticolorGibbs));
  PetscCall(PCRegister(PCGAMGMC, PCCreate_HipGAMGMC));
  PetscCall(PCRegister(PCCHOLSAMPLER, PCCreate_HipCholSampler));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* PARMGMC_HIP_HAVE_PETSC */
typedef int parmgmc_hip_pc_hipgamgmc_translation_unit_not_empty;
