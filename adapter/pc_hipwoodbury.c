/*
 * pc_hipwoodbury.c -- PCCreate_HipWoodbury: ParMGMC's Woodbury posterior sampler (reference src/woodbury.c) with its dense
 * products on MI355X devices.
 *
 * Replaces PCCreate_Woodbury (reference src/woodbury.c:291-302).  Like the reference it is a composite over two inner PETSc
 * PCs -- a SOLVER (any PC: -pc_woodbury_solver cholesky | gamg | ...) used once at set-up and a SAMPLER of the prior
 * precision A (-pc_woodbury_sampler gamgmc | mcgibbs | cholsampler | ..., normally one of this directory's constructors) --
 * with the reference's option names and inner prefixes ("pc_woodbury_solver_" and, sic, "pc_woodbury_sampler", :185-213),
 * the same ops and the same callback / deleter semantics.  What moves to the device is the rank-k algebra, through the
 * library's pmg_woodbury object:
 *   set-up (:21-91)   C = solver(B) column by column from a zero guess, T = S^-1 + B^T C, G = C T^-1
 *   sample (:263-289) w = b + B (sqrt|S| o xi);  y <- one sample of the A-sampler on w;  y -= G (B^T y)
 * On N ranks B's rows are distributed like A's (a dense MPI matrix, as in the reference) and B^T C / B^T y are summed over
 * the ranks in rank order through the ipc / RCCL transport (pmg_dist_create_comm over MPI_Allgather); S, the k x k inverse
 * and the noise k-vector are replicated.
 *
 * Built only inside a ParMGMC + PETSc tree with -DPARMGMC_HIP_HAVE_PETSC; empty otherwise.
 */
#ifdef PARMGMC_HIP_HAVE_PETSC
#include "hip_petsc_common.h"

typedef struct {
  PC            solver, sampler;
  pmg_woodbury  wb;
  pmg_dist      transport;
  pmg_host_comm hc;
  MPI_Comm      hc_comm;
  Vec           swork; /* the sampler's work vector (src/woodbury.c:278) */
  uint64_t      seed, counter, stream_id;
  HipStageBuf   bbuf, wbuf, ybuf;

  void *cbctx;
  PetscErrorCode (*scb)(PetscInt, Vec, void *);
  PetscErrorCode (*del_scb)(void *);
} PC_HipWoodbury;

static PetscErrorCode HipWoodburyFreeSetup(PC_HipWoodbury *wb)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_woodbury_destroy(&wb->wb));
  if (wb->transport) PMGCall(pmg_dist_destroy_comm(&wb->hc, &wb->transport));
  PetscCall(VecDestroy(&wb->swork));
  PetscCall(HipStageBufFree(&wb->bbuf));
  PetscCall(HipStageBufFree(&wb->wbuf));
  PetscCall(HipStageBufFree(&wb->ybuf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCReset_HipWoodbury(PC pc) /* src/woodbury.c:93-109 */
{
  PC_HipWoodbury *wb = (PC_HipWoodbury *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(HipWoodburyFreeSetup(wb));
  if (wb->solver) PetscCall(PCReset(wb->solver));
  if (wb->sampler) PetscCall(PCReset(wb->sampler));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_HipWoodbury(PC pc) /* :111-125 */
{
  PC_HipWoodbury *wb = (PC_HipWoodbury *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(HipWoodburyFreeSetup(wb));
  PetscCall(PCDestroy(&wb->solver));
  PetscCall(PCDestroy(&wb->sampler));
  if (wb->del_scb) {
    PetscCall(wb->del_scb(wb->cbctx));
    wb->del_scb = NULL;
  }
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCSetSampleCallback_C", NULL));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetSampleCallback_HipWoodbury(PC pc, PetscErrorCode (*cb)(PetscInt, Vec, void *), void *ctx, PetscErrorCode (*deleter)(void *)) /* :127-140 */
{
  PC_HipWoodbury *wb = (PC_HipWoodbury *)pc->data;

  PetscFunctionBeginUser;
  if (wb->del_scb) {
    PetscCall(wb->del_scb(wb->cbctx));
    wb->del_scb = NULL;
  }
  wb->scb     = cb;
  wb->cbctx   = ctx;
  wb->del_scb = deleter;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCSetUp_Woodbury + PCWoodburyBuildLRCCorrection (src/woodbury.c:142-183, :21-91) */
static PetscErrorCode PCSetUp_HipWoodbury(PC pc)
{
  PC_HipWoodbury    *wb = (PC_HipWoodbury *)pc->data;
  PetscBool          islrc;
  Mat                A, Bmat;
  Vec                S, x, bcol;
  PetscInt           k, mloc, lda;
  const PetscScalar *B, *Sarr;
  PetscMPIInt        size;
  HipStageBuf        xbuf = {NULL, 0};

  PetscFunctionBeginUser;
  PetscCheck(wb->solver && wb->sampler, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "Must provide sampler and solver"); /* :151 */
  PetscCall(HipWoodburyFreeSetup(wb));
  PetscCall(PetscObjectTypeCompare((PetscObject)pc->pmat, MATLRC, &islrc));
  PetscCheck(islrc, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "PCWoodbury only supports matrices of type LRC"); /* :161 */
  PetscCall(MatLRCGetMats(pc->pmat, &A, &Bmat, &S, NULL));
  PetscCall(MatCreateVecs(A, &wb->swork, NULL));
  PetscCall(PCSetOperators(wb->solver, A, A)); /* :178-181 */
  PetscCall(PCSetOperators(wb->sampler, A, A));
  PetscCall(PCSetUp(wb->solver));
  PetscCall(PCSetUp(wb->sampler));
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)pc), &size));
  if (size > 1) {
    PetscCall(HipHostComm(PetscObjectComm((PetscObject)pc), &wb->hc_comm, &wb->hc));
    PetscCall(HipCreateTransport(&wb->hc, NULL, &wb->transport));
  }
  /* this rank's rows of B, all of S (the reference scatters the whole of S to every rank, :56-63) */
  PetscCall(MatGetSize(Bmat, NULL, &k));
  PetscCall(MatGetLocalSize(Bmat, &mloc, NULL));
  PetscCall(MatDenseGetLDA(Bmat, &lda));
  PetscCall(MatDenseGetArrayRead(Bmat, &B));
  PetscCall(VecGetArrayRead(S, &Sarr));
  PMGCall(pmg_woodbury_create((int64_t)mloc, (int32_t)k, B, (int64_t)lda, Sarr, wb->transport, &wb->wb));
  PetscCall(VecRestoreArrayRead(S, &Sarr));
  PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
  /* C = solver(B), column by column from a zero guess (:35-50): the solver is a PETSc PC and works on Vecs */
  PetscCall(MatCreateVecs(A, &x, NULL));
  for (PetscInt c = 0; c < k; ++c) {
    HipVecAccess ax;
    PetscCall(VecZeroEntries(x));
    PetscCall(MatDenseGetColumnVecRead(Bmat, c, &bcol));
    PetscCall(PCApply(wb->solver, bcol, x));
    PetscCall(MatDenseRestoreColumnVecRead(Bmat, c, &bcol));
    PetscCall(HipVecGet(x, PETSC_FALSE, &xbuf, &ax));
    PMGCall(pmg_woodbury_set_c_column(wb->wb, (int32_t)c, ax.dev, NULL)); /* VecCopy(x, c), :46-48 */
    PMGHip(hipStreamSynchronize(NULL));
    PetscCall(HipVecRestore(&ax, NULL));
  }
  PetscCall(HipStageBufFree(&xbuf));
  PetscCall(VecDestroy(&x));
  PMGCall(pmg_woodbury_finish(wb->wb)); /* G = C (S^-1 + B^T C)^-1, :52-81; collective */
  PetscCall(PCDestroy(&wb->solver));    /* :182 */
  PetscCall(HipNoiseSeed(wb->stream_id, &wb->seed));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipWoodburySetSolver(PC pc, PC solver) /* PCWoodburySetSolver, :185-198 */
{
  PC_HipWoodbury *wb = (PC_HipWoodbury *)pc->data;
  const char     *prefix;

  PetscFunctionBeginUser;
  PetscCall(PCGetOptionsPrefix(pc, &prefix));
  PetscCall(PCSetOptionsPrefix(solver, prefix));
  PetscCall(PCAppendOptionsPrefix(solver, "pc_woodbury_solver_"));
  PetscCall(PetscObjectReference((PetscObject)solver));
  PetscCall(PCDestroy(&wb->solver));
  wb->solver = solver;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipWoodburySetSampler(PC pc, PC sampler) /* PCWoodburySetSampler, :200-213 */
{
  PC_HipWoodbury *wb = (PC_HipWoodbury *)pc->data;
  const char     *prefix;

  PetscFunctionBeginUser;
  PetscCall(PCGetOptionsPrefix(pc, &prefix));
  PetscCall(PCSetOptionsPrefix(sampler, prefix));
  PetscCall(PCAppendOptionsPrefix(sampler, "pc_woodbury_sampler")); /* no trailing underscore in the reference (:208): kept, the option keys are the reference's */
  PetscCall(PetscObjectReference((PetscObject)sampler));
  PetscCall(PCDestroy(&wb->sampler));
  wb->sampler = sampler;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode HipWoodburySetInnerType(PC pc, PCType type, PetscBool is_solver) /* :215-243 */
{
  PC inner;

  PetscFunctionBeginUser;
  PetscCall(PCCreate(PetscObjectComm((PetscObject)pc), &inner));
  if (is_solver) PetscCall(PCHipWoodburySetSolver(pc, inner));
  else PetscCall(PCHipWoodburySetSampler(pc, inner));
  PetscCall(PCSetType(inner, type));
  PetscCall(PCDestroy(&inner)); /* the woodbury PC holds its own reference */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetFromOptions_HipWoodbury(PC pc, PetscOptionItems_ARG PetscOptionsObject) /* :245-261 */
{
  PC_HipWoodbury *wb = (PC_HipWoodbury *)pc->data;
  char            name[256];
  PetscBool       flg;

  PetscFunctionBeginUser;
  name[0] = 0;
  PetscOptionsHeadBegin(PetscOptionsObject, "Woodbury preconditioner/ sampler options");
  PetscCall(PetscOptionsString("-pc_woodbury_solver", "Solver for the Woodbury preconditioner", NULL, name, name, sizeof(name), &flg));
  if (flg) PetscCall(HipWoodburySetInnerType(pc, name, PETSC_TRUE));
  PetscCall(PetscOptionsString("-pc_woodbury_sampler", "Sampler for the Woodbury preconditioner", NULL, name, name, sizeof(name), &flg));
  if (flg) PetscCall(HipWoodburySetInnerType(pc, name, PETSC_FALSE));
  PetscOptionsHeadEnd();
  if (wb->solver) PetscCall(PCSetFromOptions(wb->solver));
  if (wb->sampler) PetscCall(PCSetFromOptions(wb->sampler));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCApplyRichardson_Woodbury (src/woodbury.c:263-289).  w is the caller's work vector, as in the reference. */
static PetscErrorCode PCApplyRichardson_HipWoodbury(PC pc, Vec b, Vec y, Vec w, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt its, PetscBool guesszero, PetscInt *outits, PCRichardsonConvergedReason *reason)
{
  PC_HipWoodbury             *wb = (PC_HipWoodbury *)pc->data;
  PetscInt                    sits;
  PCRichardsonConvergedReason sreason;
  (void)rtol;
  (void)abstol;
  (void)dtol;
  (void)guesszero;

  PetscFunctionBeginUser;
  for (PetscInt it = 0; it < its; ++it) {
    HipVecAccess ab, aw, ay;
    PetscCall(HipVecGet(b, PETSC_FALSE, &wb->bbuf, &ab));
    PetscCall(HipVecGet(w, PETSC_TRUE, &wb->wbuf, &aw));
    PMGCall(pmg_woodbury_noisy_rhs(wb->wb, ab.dev, aw.dev, wb->seed, wb->counter++, NULL)); /* w = b + B (sqrt|S| o xi), :275-277 */
    PetscCall(HipVecRestore(&aw, NULL));
    PetscCall(HipVecRestore(&ab, NULL));
    PetscCall(PCApplyRichardson(wb->sampler, w, y, wb->swork, 0., 0., 0., 1, PETSC_FALSE, &sits, &sreason)); /* :278 */
    PetscCall(HipVecGet(y, PETSC_TRUE, &wb->ybuf, &ay));
    PMGCall(pmg_woodbury_correct(wb->wb, ay.dev, NULL)); /* y -= G (B^T y), :280-282; collective */
    PetscCall(HipVecRestore(&ay, NULL));
    if (wb->scb) PetscCall(wb->scb(it, y, wb->cbctx)); /* :284 */
  }
  *outits = its;
  *reason = PCRICHARDSON_CONVERGED_ITS;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_HipWoodbury(PC pc) /* :291-302 */
{
  PC_HipWoodbury *wb;

  PetscFunctionBeginUser;
  PetscCall(PetscNew(&wb));
  wb->stream_id            = ParMGMCHipNextStreamId();
  pc->data                 = wb;
  pc->ops->setup           = PCSetUp_HipWoodbury;
  pc->ops->reset           = PCReset_HipWoodbury;
  pc->ops->destroy         = PCDestroy_HipWoodbury;
  pc->ops->setfromoptions  = PCSetFromOptions_HipWoodbury;
  pc->ops->applyrichardson = PCApplyRichardson_HipWoodbury;
  PetscCall(PCRegisterSetSampleCallback(pc, PCSetSampleCallback_HipWoodbury));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* PARMGMC_HIP_HAVE_PETSC */
typedef int parmgmc_hip_pc_hipwoodbury_translation_unit_not_empty;
