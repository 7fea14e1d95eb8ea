/*
 * pc_hipgibbs.c -- PETSc constructors that run ParMGMC's two stand-alone Gibbs samplers on an MI355X.
 *
 *   PCCreate_HipSORGibbs         replaces PCCreate_SORGibbs        (reference src/pc_sorgibbs.c:306-324)
 *   PCCreate_HipMulticolorGibbs  replaces PCCreate_MulticolorGibbs (reference src/pc_mcgibbs.c:305-327)
 *
 * Both fill pc->ops exactly as the reference constructors do (setup, apply [sorgibbs], applyrichardson, destroy,
 * reset, setfromoptions, view, and the composed "PCSetSampleCallback_C"), keep the reference's option names and
 * callback / deleter semantics, and forward the arithmetic to libparmgmc_hip through its C-ABI:
 *   PCSetUp            -> MATSEQAIJ: pmg_mcsor_create_csr_idx on the arrays of MatSeqAIJGetCSRAndMemType (src/mc_sor.c:250),
 *                         + pmg_mcsor_set_lowrank for a MATLRC operator (src/mc_sor.c:572-595);
 *                         MATMPIAIJ on more than one rank (one rank = one device; MCSORSetUp's MPIAIJ branch, src/mc_sor.c:
 *                         152-214,553-605): MatMPIAIJGetSeqAIJ's blocks merged (pmg_rowblock_merge_mpiaij), the ipc / RCCL
 *                         transport bootstrapped through MPI_Allgather (pmg_dist_create_comm), colouring + ghost plan +
 *                         local operator + C sample loop from pmg_rowblock_sampler_create, MATLRC via pmg_distmcsor_set_lowrank
 *   PCApplyRichardson  -> pmg_mcsor_sample / pmg_distmcsor_sample (src/pc_mcgibbs.c:155-188 / src/pc_sorgibbs.c:115-134;
 *                         MCSORApply_MPIAIJ src/mc_sor.c:298-381: one ghost update per colour)
 *   PCApply (sorgibbs) -> y = 0, one sample (src/pc_sorgibbs.c:105-113)
 * Noise: counter-based (seed, sample counter) instead of the sequential PetscRandom stream; the counter lives in the
 * PC, so consecutive KSPSolve calls continue one chain.
 *
 * Built only inside a ParMGMC + PETSc tree with -DPARMGMC_HIP_HAVE_PETSC; empty otherwise.
 */
#ifdef PARMGMC_HIP_HAVE_PETSC
#include "hip_petsc_common.h"

typedef struct {
  pmg_mcsor   mc;
  /* more than one rank: row block of a MATMPIAIJ */
  pmg_distmcsor dm;
  pmg_dist      transport;
  pmg_host_comm hc;
  MPI_Comm      hc_comm;
  PetscInt      nowned;
  uint64_t      stream_id; /* this PC's noise stream (HipNoiseSeed) */
  PetscBool   scaled;   /* PETSC_TRUE: mcgibbs (noise scaled by sqrt((2-omega)/omega)); PETSC_FALSE: sorgibbs (omega = 1) */
  PetscReal   omega;
  MatSORType  type;
  PetscBool   lexicographic; /* colouring = dependency levels of the natural order: PETSc MatSOR's result, update for update */
  PetscBool   iterated;      /* one rank: first-fit + one round of iterated greedy (PMG_COLORING_ITERATED): a class fewer on P1 meshes */
  uint64_t    seed, counter;
  HipStageBuf bbuf, ybuf;
  PetscInt    ncolors;

  void *cbctx;
  PetscErrorCode (*scb)(PetscInt, Vec, void *);
  PetscErrorCode (*del_scb)(void *);
} PC_HipGibbs;

static PetscErrorCode HipGibbsRelease(PC_HipGibbs *hg)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_distmcsor_destroy(&hg->dm));
  PMGCall(pmg_mcsor_destroy(&hg->mc));
  if (hg->transport) PMGCall(pmg_dist_destroy_comm(&hg->hc, &hg->transport)); /* collective, like PCReset / PCDestroy themselves */
  PetscCall(HipStageBufFree(&hg->bbuf));
  PetscCall(HipStageBufFree(&hg->ybuf));
  if (hg->del_scb) { /* reference src/pc_sorgibbs.c:153-156,173-176 */
    PetscCall(hg->del_scb(hg->cbctx));
    hg->del_scb = NULL;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCReset_HipGibbs(PC pc)
{
  PetscFunctionBeginUser;
  PetscCall(HipGibbsRelease((PC_HipGibbs *)pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_HipGibbs(PC pc)
{
  PetscFunctionBeginUser;
  PetscCall(HipGibbsRelease((PC_HipGibbs *)pc->data));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCSetSampleCallback_C", NULL));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MCSORSetUp on a MATMPIAIJ (reference src/mc_sor.c:553-605 with MatCreateScatters :152-214 and the parallel colouring
   :383-395): everything behind the C-ABI, the only collective it needs from here is MPI_Allgather */
static PetscErrorCode HipGibbsSetUpMPIAIJ(PC pc, Mat A, PetscBool islrc)
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;
  int64_t     *rp, *ci, *starts;
  double      *v;
  PetscBool    ismpi;

  PetscFunctionBeginUser;
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATMPIAIJ, &ismpi));
  PetscCheck(ismpi, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "Matrix type not supported (MATSEQAIJ, MATMPIAIJ, or MATLRC over one of them)"); /* src/mc_sor.c:568 */
  PetscCheck(!hg->lexicographic, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "-pc_hipgibbs_lexicographic is a one-rank option: on several ranks the sweep is multicoloured, as in the reference");
  PetscCall(HipHostComm(PetscObjectComm((PetscObject)A), &hg->hc_comm, &hg->hc));
  PetscCall(HipMPIAIJRows(A, &rp, &ci, &v, &starts, &hg->nowned));
  PetscCall(HipCreateTransport(&hg->hc, NULL, &hg->transport));
  /* colouring: the library's first-fit rule on the global matrix, computed rank after rank (the reference takes PETSc's
     randomised JP colouring: any valid distance-1 colouring gives a valid Gibbs sampler) */
  PMGCall(pmg_rowblock_sampler_create(&hg->hc, hg->transport, starts, rp, ci, v, 64, 0, NULL, hg->scaled ? hg->omega : 1.0, &hg->mc, &hg->dm));
  PetscCall(PetscFree(rp));
  PetscCall(PetscFree(ci));
  PetscCall(PetscFree(v));
  PetscCall(PetscFree(starts));
  if (islrc) { /* B's rows are distributed like A's (a dense MPI matrix): hand over the local block */
    Mat                Abase, Bmat;
    Vec                S;
    PetscInt           k, lda, mloc;
    const PetscScalar *B, *Sarr;
    double            *Bl;
    int32_t            nlocal;

    PetscCall(MatLRCGetMats(pc->pmat, &Abase, &Bmat, &S, NULL));
    PetscCall(MatGetSize(Bmat, NULL, &k));
    PetscCall(MatGetLocalSize(Bmat, &mloc, NULL));
    PetscCheck(mloc == hg->nowned, PetscObjectComm((PetscObject)pc), PETSC_ERR_ARG_SIZ, "the rows of B must be distributed like the rows of A");
    PetscCall(MatDenseGetLDA(Bmat, &lda));
    PetscCall(MatDenseGetArrayRead(Bmat, &B));
    PMGCall(pmg_mcsor_get_size(hg->mc, &nlocal)); /* owned rows + ghost rows */
    PetscCall(PetscCalloc1((size_t)nlocal * (size_t)k, &Bl));
    for (PetscInt c = 0; c < k; ++c) PetscCall(PetscArraycpy(Bl + (size_t)nlocal * c, B + (size_t)lda * c, mloc));
    PetscCall(VecGetArrayRead(S, &Sarr)); /* S is replicated on every rank in the reference (src/woodbury.c:56-63 scatters all of it) */
    PMGCall(pmg_distmcsor_set_lowrank(hg->dm, (int32_t)k, nlocal, (int32_t)hg->nowned, Bl, Sarr));
    PetscCall(VecRestoreArrayRead(S, &Sarr));
    PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
    PetscCall(PetscFree(Bl));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCSetUp_SORGibbs / PCSetUp_MulticolorGibbs (reference src/pc_sorgibbs.c:181-260, src/pc_mcgibbs.c:213-255) */
static PetscErrorCode PCSetUp_HipGibbs(PC pc)
{
  PC_HipGibbs    *hg = (PC_HipGibbs *)pc->data;
  Mat             A  = pc->pmat;
  PetscBool       islrc, isseq;
  const PetscInt *ia, *ja;
  PetscScalar    *aa;
  PetscInt        n;
  PetscMPIInt     size;

  PetscFunctionBeginUser;
  PMGCall(pmg_distmcsor_destroy(&hg->dm)); /* PCSetUp may run again on a new operator */
  PMGCall(pmg_mcsor_destroy(&hg->mc));
  if (hg->transport) PMGCall(pmg_dist_destroy_comm(&hg->hc, &hg->transport));
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATLRC, &islrc));
  if (islrc) PetscCall(MatLRCGetMats(pc->pmat, &A, NULL, NULL, NULL));
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)A), &size));
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATSEQAIJ, &isseq));
  PetscCall(HipNoiseSeed(hg->stream_id, &hg->seed));
  if (size > 1) { /* one rank = one device: row blocks with one ghost update per colour (reference src/mc_sor.c:298-381) */
    PetscCall(HipGibbsSetUpMPIAIJ(pc, A, islrc));
    {
      int32_t nc;
      PMGCall(pmg_mcsor_get_num_colors(hg->mc, &nc));
      hg->ncolors = nc - 1; /* the local operator keeps its ghost rows in one more colour that is never swept */
    }
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCheck(isseq, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "Matrix type not supported (MATSEQAIJ, MATMPIAIJ, or MATLRC over one of them)");
  PetscCall(MatGetSize(A, &n, NULL));
  PetscCall(MatSeqAIJGetCSRAndMemType(A, &ia, &ja, &aa, NULL)); /* borrowed host arrays, as src/mc_sor.c:250 */
  PMGCall(pmg_mcsor_create_csr_idx((int64_t)n, ia, ja, aa, PMG_IDX_WIDTH, &hg->mc));
  PMGCall(pmg_mcsor_set_coloring(hg->mc, hg->lexicographic ? PMG_COLORING_LEXLEVELS : (hg->iterated ? PMG_COLORING_ITERATED : PMG_COLORING_GREEDY), NULL));
  PMGCall(pmg_mcsor_set_omega(hg->mc, hg->scaled ? hg->omega : 1.0));
  PMGCall(pmg_mcsor_set_sweep_type(hg->mc, (int)hg->type));
  if (islrc) { /* A + B S B^T: PrepareRHS_LRC + MCSORPostSOR_LRC (src/pc_mcgibbs.c:130-140, src/mc_sor.c:101-112) */
    Mat                Abase, Bmat;
    Vec                S;
    PetscInt           k;
    const PetscScalar *B, *Sarr;
    PetscScalar       *Bcopy;

    PetscCall(HipGetLRC(pc->pmat, &Abase, &k, &B, &Bcopy, &Bmat, &S));
    PetscCall(VecGetArrayRead(S, &Sarr));
    PMGCall(pmg_mcsor_set_lowrank(hg->mc, (int32_t)k, Bcopy ? Bcopy : B, Sarr));
    PetscCall(VecRestoreArrayRead(S, &Sarr));
    PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
    PetscCall(PetscFree(Bcopy));
  }
  PMGCall(pmg_mcsor_setup(hg->mc)); /* colours, permutes, uploads: the CSR arrays are no longer needed afterwards */
  {
    int32_t nc;
    PMGCall(pmg_mcsor_get_num_colors(hg->mc, &nc));
    hg->ncolors = nc;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* `its` samples of the chain on (b, y): the loop of PCApplyRichardson_MulticolorGibbs (src/pc_mcgibbs.c:166-184) */
static PetscErrorCode HipGibbsSamples(PC pc, Vec b, Vec y, PetscInt its)
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;
  HipVecAccess ab, ay;

  PetscFunctionBeginUser;
  PetscCall(PetscLogEventBegin(MULTICOL_SOR, pc, b, y, 0)); /* the reference logs every MCSORApply under this event, src/mc_sor.c:221 */
  PetscCall(HipVecGet(b, PETSC_FALSE, &hg->bbuf, &ab));
  PetscCall(HipVecGet(y, PETSC_TRUE, &hg->ybuf, &ay));
  if (hg->dm) { /* row block of a MATMPIAIJ: b and y are the local parts of the Vecs (natural order of the owned rows) */
    const PetscInt chunk = hg->scb ? 1 : its;
    for (PetscInt it = 0; it < its; it += chunk) {
      PMGCall(pmg_distmcsor_sample(hg->dm, (int32_t)hg->nowned, ab.dev, ay.dev, (int32_t)chunk, (int)hg->scaled, (int)hg->type, hg->seed, hg->counter, &hg->counter, NULL));
      if (hg->scb) PetscCall(HipCallSampleCallback(hg->scb, hg->cbctx, it, &ay, NULL)); /* src/pc_mcgibbs.c:183 */
    }
  } else if (!hg->scb) {
    PMGCall(pmg_mcsor_sample(hg->mc, ab.dev, ay.dev, (int32_t)its, (int)hg->scaled, hg->seed, hg->counter, &hg->counter, NULL));
  } else {
    for (PetscInt it = 0; it < its; ++it) {
      PMGCall(pmg_mcsor_sample(hg->mc, ab.dev, ay.dev, 1, (int)hg->scaled, hg->seed, hg->counter, &hg->counter, NULL));
      PetscCall(HipCallSampleCallback(hg->scb, hg->cbctx, it, &ay, NULL)); /* src/pc_mcgibbs.c:183 */
    }
  }
  PetscCall(HipVecRestore(&ay, NULL));
  PetscCall(HipVecRestore(&ab, NULL));
  PetscCall(PetscLogEventEnd(MULTICOL_SOR, pc, b, y, 0));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApplyRichardson_HipGibbs(PC pc, Vec b, Vec y, Vec w, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt its, PetscBool guesszero, PetscInt *outits, PCRichardsonConvergedReason *reason)
{
  (void)w; /* the noisy right-hand side is formed in registers inside the sweep kernel: the work vector stays untouched */
  (void)rtol;
  (void)abstol;
  (void)dtol;
  (void)guesszero; /* ignored like the reference (src/pc_sorgibbs.c:117-120) */

  PetscFunctionBeginUser;
  PetscCall(HipGibbsSamples(pc, b, y, its));
  *outits = its;
  *reason = PCRICHARDSON_CONVERGED_ITS;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCApply_SORGibbs (src/pc_sorgibbs.c:105-113): y = 0, one sample, no callback */
static PetscErrorCode PCApply_HipSORGibbs(PC pc, Vec b, Vec y)
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;
  PetscErrorCode (*scb)(PetscInt, Vec, void *) = hg->scb;

  PetscFunctionBeginUser;
  PetscCall(VecZeroEntries(y));
  hg->scb = NULL;
  PetscCall(HipGibbsSamples(pc, b, y, 1));
  hg->scb = scb;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* option names of the reference: src/pc_mcgibbs.c:190-211, src/pc_sorgibbs.c:262-276 */
static PetscErrorCode PCSetFromOptions_HipGibbs(PC pc, PetscOptionItems_ARG PetscOptionsObject)
{
  PC_HipGibbs *hg   = (PC_HipGibbs *)pc->data;
  PetscBool    flag = PETSC_FALSE;

  PetscFunctionBeginUser;
  if (hg->scaled) {
    PetscOptionsHeadBegin(PetscOptionsObject, "MulticolorGibbs options");
    PetscCall(PetscOptionsRangeReal("-pc_mcgibbs_omega", "MulticolorGibbs SOR parameter", NULL, hg->omega, &hg->omega, &flag, 0.0, 2.0));
    flag = PETSC_FALSE;
    PetscCall(PetscOptionsBool("-pc_mcgibbs_forward", "MulticolorGibbs forward sweep", NULL, (PetscBool)(hg->type == SOR_FORWARD_SWEEP), &flag, NULL));
    if (flag) hg->type = SOR_FORWARD_SWEEP;
    flag = PETSC_FALSE;
    PetscCall(PetscOptionsBool("-pc_mcgibbs_backward", "MulticolorGibbs backward sweep", NULL, (PetscBool)(hg->type == SOR_BACKWARD_SWEEP), &flag, NULL));
    if (flag) hg->type = SOR_BACKWARD_SWEEP;
    flag = PETSC_FALSE;
    PetscCall(PetscOptionsBool("-pc_mcgibbs_symmetric", "MulticolorGibbs symmetric sweep", NULL, (PetscBool)(hg->type == SOR_SYMMETRIC_SWEEP), &flag, NULL));
    if (flag) hg->type = SOR_SYMMETRIC_SWEEP;
  } else {
    PetscOptionsHeadBegin(PetscOptionsObject, "SOR Gibbs options");
    PetscCall(PetscOptionsBool("-pc_sorgibbs_forward", "SOR Gibbs forward sweep", NULL, (PetscBool)(hg->type == SOR_FORWARD_SWEEP), &flag, NULL));
    if (flag) hg->type = SOR_FORWARD_SWEEP;
    /* -pc_sorgibbs_local_forward (the Hogwild variant: every rank sweeps its diagonal block with stale off-process values,
       src/pc_sorgibbs.c:272-275) has no device counterpart -- on one device the multicolour sweep IS the exact Gauss-Seidel
       sweep, on several ranks the row-block sweep exchanges ghost values per colour.  Asking for it is an error, not a
       silently different sampler */
    flag = PETSC_FALSE;
    PetscCall(PetscOptionsBool("-pc_sorgibbs_local_forward", "SOR Gibbs local forward sweep (Hogwild): not supported on the device", NULL, PETSC_FALSE, &flag, NULL));
    PetscCheck(!flag, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "-pc_sorgibbs_local_forward (Hogwild, src/pc_sorgibbs.c:272-275) is not supported by the device sampler: use -pc_sorgibbs_forward (exact sweep; on several ranks one ghost update per colour)");
  }
  /* not in the reference: the order of the device sweep.  Default: greedy multicolouring (few colours = few launches);
     lexicographic = the dependency levels of the natural order, i.e. PETSc MatSOR's result update for update */
  PetscCall(PetscOptionsBool("-pc_hipgibbs_lexicographic", "sweep in the dependency levels of the natural row order (MatSOR's order)", NULL, hg->lexicographic, &hg->lexicographic, NULL));
  PetscCall(PetscOptionsBool("-pc_hipgibbs_iterated_coloring", "one rank: first-fit colouring followed by one round of iterated greedy (never more colours, often one fewer)", NULL, hg->iterated, &hg->iterated, NULL));
  PetscOptionsHeadEnd();
  if (hg->mc) { /* options changed after set-up: the library applies them lazily like MCSORSetOmega (src/mc_sor.c:412-420) */
    PMGCall(pmg_mcsor_set_omega(hg->mc, hg->scaled ? hg->omega : 1.0));
    PMGCall(pmg_mcsor_set_sweep_type(hg->mc, (int)hg->type));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCView_HipGibbs(PC pc, PetscViewer viewer)
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;

  PetscFunctionBeginUser;
  if (hg->scaled) PetscCall(PetscViewerASCIIPrintf(viewer, "Number of colours: %" PetscInt_FMT "\n", hg->ncolors)); /* src/pc_mcgibbs.c:257-266 */
  else PetscCall(PetscViewerASCIIPrintf(viewer, "Sweep type: Forward\n"));                                           /* src/pc_sorgibbs.c:300 */
  PetscCall(PetscViewerASCIIPrintf(viewer, "Device sweep: libparmgmc_hip %s (%s), %s order, %" PetscInt_FMT " colour launches per sweep\n", pmg_version(), pmg_gpu_arch(), hg->lexicographic ? "lexicographic" : (hg->iterated ? "iterated-greedy multicolour" : "greedy multicolour"), hg->ncolors));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* src/pc_sorgibbs.c:278-293, src/pc_mcgibbs.c:290-303: a previous context is deleted before it is replaced */
static PetscErrorCode PCSetSampleCallback_HipGibbs(PC pc, PetscErrorCode (*cb)(PetscInt, Vec, void *), void *ctx, PetscErrorCode (*deleter)(void *))
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;

  PetscFunctionBeginUser;
  if (hg->del_scb) {
    PetscCall(hg->del_scb(hg->cbctx));
    hg->del_scb = NULL;
  }
  hg->scb     = cb;
  hg->cbctx   = ctx;
  hg->del_scb = deleter;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCMulticolorGibbsSetOmega / SetSweepType (src/pc_mcgibbs.c:268-288) for callers that use the typed setters */
PetscErrorCode PCHipGibbsSetOmega(PC pc, PetscReal omega)
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;

  PetscFunctionBeginUser;
  hg->omega = omega;
  if (hg->mc) PMGCall(pmg_mcsor_set_omega(hg->mc, omega));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipGibbsSetSweepType(PC pc, MatSORType type)
{
  PC_HipGibbs *hg = (PC_HipGibbs *)pc->data;

  PetscFunctionBeginUser;
  PetscCheck(type == SOR_FORWARD_SWEEP || type == SOR_BACKWARD_SWEEP || type == SOR_SYMMETRIC_SWEEP, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "Only forward, backward and symmetric sweep supported"); /* src/mc_sor.c:427 */
  hg->type = type;
  if (hg->mc) PMGCall(pmg_mcsor_set_sweep_type(hg->mc, (int)type));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCCreate_HipGibbsCommon(PC pc, PetscBool scaled)
{
  PC_HipGibbs *hg;

  PetscFunctionBeginUser;
  PetscCall(PetscNew(&hg));
  hg->scaled    = scaled;
  hg->omega     = 1;
  hg->type      = SOR_FORWARD_SWEEP;
  hg->stream_id = ParMGMCHipNextStreamId(); /* PCs are created collectively, in the same order on every rank */

  pc->data                 = hg;
  pc->ops->setup           = PCSetUp_HipGibbs;
  pc->ops->applyrichardson = PCApplyRichardson_HipGibbs;
  pc->ops->destroy         = PCDestroy_HipGibbs;
  pc->ops->reset           = PCReset_HipGibbs;
  pc->ops->setfromoptions  = PCSetFromOptions_HipGibbs;
  pc->ops->view            = PCView_HipGibbs;
  if (!scaled) pc->ops->apply = PCApply_HipSORGibbs; /* only sorgibbs has PCApply in the reference (src/pc_sorgibbs.c:315) */
  PetscCall(PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipGibbs));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_HipSORGibbs(PC pc)
{
  PetscFunctionBeginUser;
  PetscCall(PCCreate_HipGibbsCommon(pc, PETSC_FALSE));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_HipMulticolorGibbs(PC pc)
{
  PetscFunctionBeginUser;
  PetscCall(PCCreate_HipGibbsCommon(pc, PETSC_TRUE));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- exact coarse sampler: PCCreate_CholSampler, dense path (reference src/pc_chols.c:174-194,262-342) ------------- */
typedef struct {
  pmg_chol    ch;
  PetscInt    dense_threshold; /* -pc_cholsampler_dense_threshold (src/pc_chols.c:416): kept and shown; the device factorisation is dense at every size */
  PetscBool   is_gamg_coarse;  /* -pc_cholsampler_coarse_gamg (src/pc_chols.c:415) */
  uint64_t    seed, counter, stream_id;
  HipStageBuf bbuf, ybuf;
  void *cbctx;
  PetscErrorCode (*scb)(PetscInt, Vec, void *);
  PetscErrorCode (*del_scb)(void *);
} PC_HipChol;

static PetscErrorCode PCSetUp_HipChol(PC pc)
{
  PC_HipChol        *hc = (PC_HipChol *)pc->data;
  Mat                A  = pc->pmat, Bmat = NULL;
  Vec                S  = NULL;
  PetscBool          islrc;
  const PetscInt    *ia, *ja;
  PetscScalar       *aa, *Bcopy = NULL;
  const PetscScalar *B = NULL, *Sarr = NULL;
  PetscInt           n, k = 0;

  PetscFunctionBeginUser;
  PMGCall(pmg_chol_destroy(&hc->ch));
  {
    PetscMPIInt size;
    PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)A), &size));
    PetscCheck(size == 1, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "the device Cholesky sampler factors an operator that lives on ONE rank (as GAMG's coarse grid does, src/pc_chols.c:38-47); this one is spread over %d ranks", (int)size);
  }
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATLRC, &islrc));
  if (islrc) { /* factor the explicit sum A + B S B^T, src/pc_chols.c:119-153 */
    PetscCall(HipGetLRC(pc->pmat, &A, &k, &B, &Bcopy, &Bmat, &S));
    PetscCall(VecGetArrayRead(S, &Sarr));
  }
  PetscCall(MatGetSize(A, &n, NULL));
  PetscCall(MatSeqAIJGetCSRAndMemType(A, &ia, &ja, &aa, NULL));
  PMGCall(pmg_chol_create_csr_idx((int64_t)n, ia, ja, aa, PMG_IDX_WIDTH, (int32_t)k, Bcopy ? Bcopy : B, Sarr, &hc->ch));
  if (islrc) {
    PetscCall(VecRestoreArrayRead(S, &Sarr));
    PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
    PetscCall(PetscFree(Bcopy));
  }
  PetscCall(HipNoiseSeed(hc->stream_id, &hc->seed));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* y = L^-T (L^-1 b + xi), src/pc_chols.c:262-291 */
static PetscErrorCode PCApply_HipChol(PC pc, Vec b, Vec y)
{
  PC_HipChol  *hc = (PC_HipChol *)pc->data;
  HipVecAccess ab, ay;

  PetscFunctionBeginUser;
  PetscCall(HipVecGet(b, PETSC_FALSE, &hc->bbuf, &ab));
  PetscCall(HipVecGet(y, PETSC_TRUE, &hc->ybuf, &ay));
  PMGCall(pmg_chol_sample(hc->ch, ab.dev, ay.dev, 1, hc->seed, hc->counter++, NULL));
  PetscCall(HipVecRestore(&ay, NULL));
  PetscCall(HipVecRestore(&ab, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* independent exact samples, src/pc_chols.c:293-342 */
static PetscErrorCode PCApplyRichardson_HipChol(PC pc, Vec b, Vec y, Vec w, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt its, PetscBool guesszero, PetscInt *outits, PCRichardsonConvergedReason *reason)
{
  PC_HipChol *hc = (PC_HipChol *)pc->data;
  (void)w;
  (void)rtol;
  (void)abstol;
  (void)dtol;
  (void)guesszero;

  PetscFunctionBeginUser;
  for (PetscInt it = 0; it < its; ++it) {
    PetscCall(PCApply_HipChol(pc, b, y));
    if (hc->scb) PetscCall(hc->scb(it, y, hc->cbctx));
  }
  *outits = its;
  *reason = PCRICHARDSON_CONVERGED_ITS;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode HipCholRelease(PC_HipChol *hc)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_chol_destroy(&hc->ch));
  PetscCall(HipStageBufFree(&hc->bbuf));
  PetscCall(HipStageBufFree(&hc->ybuf));
  if (hc->del_scb) {
    PetscCall(hc->del_scb(hc->cbctx));
    hc->del_scb = NULL;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCReset_HipChol(PC pc)
{
  PetscFunctionBeginUser;
  PetscCall(HipCholRelease((PC_HipChol *)pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_HipChol(PC pc)
{
  PetscFunctionBeginUser;
  PetscCall(HipCholRelease((PC_HipChol *)pc->data));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCSetSampleCallback_C", NULL));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCSetFromOptions_CholSampler (src/pc_chols.c:406-419), same keys.  What the device path does with them:
     -pc_cholsampler_dense_threshold N : the reference factors sequential blocks of at most N rows densely (LAPACK) and larger
        ones with a sparse direct solver; the device factorisation is dense (blocked MFMA Cholesky) at EVERY size, so the value
        only decides nothing -- it is stored and shown by -ksp_view so that an option file of the reference is accepted;
     -pc_cholsampler_coarse_gamg : the reference then factors on rank 0 only and scatters (GAMG leaves the coarse grid on one
        rank, src/pc_chols.c:38-47,272-282).  Here the coarse operator must already live on ONE rank (PCSetUp checks it): a
        coarse matrix spread over several ranks is an error with or without the flag, not a silent gather. */
static PetscErrorCode PCSetFromOptions_HipChol(PC pc, PetscOptionItems_ARG PetscOptionsObject)
{
  PC_HipChol *hc   = (PC_HipChol *)pc->data;
  PetscBool   flag = PETSC_FALSE;

  PetscFunctionBeginUser;
  PetscOptionsHeadBegin(PetscOptionsObject, "Cholesky options");
  PetscCall(PetscOptionsBool("-pc_cholsampler_coarse_gamg", "Sampler is coarse GAMGMC sampler", NULL, flag, &flag, NULL));
  if (flag) hc->is_gamg_coarse = PETSC_TRUE;
  PetscCall(PetscOptionsInt("-pc_cholsampler_dense_threshold", "Sequential blocks of size <= this are factored and solved densely (the device path is dense at every size)", NULL, hc->dense_threshold, &hc->dense_threshold, NULL));
  PetscOptionsHeadEnd();
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCView_HipChol(PC pc, PetscViewer viewer)
{
  PC_HipChol *hc = (PC_HipChol *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(PetscViewerASCIIPrintf(viewer, "Dense Cholesky sampler on the device (libparmgmc_hip %s, %s; MFMA f64 trailing updates)\n", pmg_version(), pmg_gpu_arch()));
  PetscCall(PetscViewerASCIIPrintf(viewer, "dense_threshold %" PetscInt_FMT " (not used: dense at every size), coarse_gamg %s\n", hc->dense_threshold, hc->is_gamg_coarse ? "true" : "false"));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetSampleCallback_HipChol(PC pc, PetscErrorCode (*cb)(PetscInt, Vec, void *), void *ctx, PetscErrorCode (*deleter)(void *))
{
  PC_HipChol *hc = (PC_HipChol *)pc->data;

  PetscFunctionBeginUser;
  if (hc->del_scb) {
    PetscCall(hc->del_scb(hc->cbctx));
    hc->del_scb = NULL;
  }
  hc->scb     = cb;
  hc->cbctx   = ctx;
  hc->del_scb = deleter;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_HipCholSampler(PC pc)
{
  PC_HipChol *hc;

  PetscFunctionBeginUser;
  PetscCall(PetscNew(&hc));
  hc->stream_id            = ParMGMCHipNextStreamId();
  hc->dense_threshold      = 64; /* src/pc_chols.c:432 */
  pc->data                 = hc;
  pc->ops->setup           = PCSetUp_HipChol;
  pc->ops->apply           = PCApply_HipChol;
  pc->ops->applyrichardson = PCApplyRichardson_HipChol;
  pc->ops->destroy         = PCDestroy_HipChol;
  pc->ops->reset           = PCReset_HipChol;
  pc->ops->setfromoptions  = PCSetFromOptions_HipChol;
  pc->ops->view            = PCView_HipChol;
  PetscCall(PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipChol));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* PARMGMC_HIP_HAVE_PETSC */
typedef int parmgmc_hip_pc_hipgibbs_translation_unit_not_empty;
