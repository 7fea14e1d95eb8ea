/*
 * pc_hipparsor.c -- PCCreate_HipPARSOR: ParMGMC's parallel SOR preconditioner (reference src/pc_parsor.c) on MI355X devices.
 *
 * Replaces PCCreate_PARSOR (reference src/pc_parsor.c:1021-1039): same ops (setup, apply, reset, destroy, setfromoptions,
 * view), same options (-pc_parsor_omega, -pc_parsor_its), same typed setters, same row scaling omega / d in one rounding
 * (LocalMatInvertDiagonalForSOR, :53-86).  PCApply = `its` forward SOR sweeps from a zero initial guess (:880-891);
 * PCHipPARSORApplySOR = PCPARSORApplySOR (:893-906).
 *
 *   one rank   (the reference falls back to PETSc's PCSOR there, :941-951): the dependency levels of the natural row order
 *              swept in order (PMG_COLORING_LEXLEVELS) -- the lexicographic forward sweep of MatSOR, update for update;
 *   N ranks    MATMPIAIJ by row blocks, one rank = one device: the multicolour sweep with one ghost update per colour
 *              (pmg_rowblock_sampler_create + pmg_distmcsor_apply).  DELIBERATE DIFFERENCE: the reference orders
 *              the rows TOP / INT1 / MID / INT2 / BOT per rank with a message pipeline for the MID rows (:703-878) so that the
 *              result is a lexicographic sweep by (rank colour, local row); on devices the colours are the parallelism and
 *              the result is the coloured Gauss-Seidel sweep -- another valid SOR splitting with the same fixed point.
 *              (The library reproduces the reference's multi-rank ORDER on one device for any emulated partition:
 *              pmg_pc_parsor_set_partition, tests/test_parsor_partition.py.)
 *
 * Built only inside a ParMGMC + PETSc tree with -DPARMGMC_HIP_HAVE_PETSC; empty otherwise.
 */
#ifdef PARMGMC_HIP_HAVE_PETSC
#include "hip_petsc_common.h"

typedef struct {
  PetscReal     omega;
  PetscInt      its;
  pmg_mcsor     mc;
  pmg_distmcsor dm;
  pmg_dist      transport;
  pmg_host_comm hc;
  MPI_Comm      hc_comm;
  PetscInt      nowned, ncolors;
  HipStageBuf   bbuf, xbuf;
} PC_HipPARSOR;

static PetscErrorCode HipPARSORRelease(PC_HipPARSOR *ps)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_distmcsor_destroy(&ps->dm));
  PMGCall(pmg_mcsor_destroy(&ps->mc));
  if (ps->transport) PMGCall(pmg_dist_destroy_comm(&ps->hc, &ps->transport));
  PetscCall(HipStageBufFree(&ps->bbuf));
  PetscCall(HipStageBufFree(&ps->xbuf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCReset_HipPARSOR(PC pc) /* src/pc_parsor.c:908-920 */
{
  PetscFunctionBeginUser;
  PetscCall(HipPARSORRelease((PC_HipPARSOR *)pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_HipPARSOR(PC pc) /* :922-932 */
{
  PetscFunctionBeginUser;
  PetscCall(HipPARSORRelease((PC_HipPARSOR *)pc->data));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetUp_HipPARSOR(PC pc) /* :934-968 */
{
  PC_HipPARSOR *ps = (PC_HipPARSOR *)pc->data;
  PetscBool     ismpi, isseq;
  PetscMPIInt   size;
  int32_t       nc;

  PetscFunctionBeginUser;
  PetscCall(HipPARSORRelease(ps));
  PetscCall(PetscObjectTypeCompare((PetscObject)pc->pmat, MATMPIAIJ, &ismpi));
  PetscCall(PetscObjectTypeCompare((PetscObject)pc->pmat, MATSEQAIJ, &isseq));
  PetscCheck(ismpi || isseq, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "PCPARSOR only supports MATMPIAIJ and MATSEQAIJ matrices"); /* :954 */
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)pc->pmat), &size));
  if (isseq || size == 1) {
    const PetscInt *ia, *ja;
    PetscScalar    *aa;
    PetscInt        n;
    Mat             A = pc->pmat;

    if (ismpi) PetscCall(MatMPIAIJGetSeqAIJ(pc->pmat, &A, NULL, NULL)); /* one rank: the diagonal block is the matrix */
    PetscCall(MatGetSize(A, &n, NULL));
    PetscCall(MatSeqAIJGetCSRAndMemType(A, &ia, &ja, &aa, NULL));
    PMGCall(pmg_mcsor_create_csr_idx((int64_t)n, ia, ja, aa, PMG_IDX_WIDTH, &ps->mc));
    PMGCall(pmg_mcsor_set_coloring(ps->mc, PMG_COLORING_LEXLEVELS, NULL)); /* MatSOR's forward sweep, update for update */
    PMGCall(pmg_mcsor_set_idiag_by_division(ps->mc, 1));                   /* omega / d in one rounding, :69-81 */
    PMGCall(pmg_mcsor_set_omega(ps->mc, ps->omega));
    PMGCall(pmg_mcsor_set_sweep_type(ps->mc, (int)SOR_FORWARD_SWEEP));
    PMGCall(pmg_mcsor_setup(ps->mc));
    ps->nowned = n;
  } else {
    int64_t *rp, *ci, *starts;
    double  *v;

    PetscCall(HipHostComm(PetscObjectComm((PetscObject)pc->pmat), &ps->hc_comm, &ps->hc));
    PetscCall(HipMPIAIJRows(pc->pmat, &rp, &ci, &v, &starts, &ps->nowned));
    PetscCall(HipCreateTransport(&ps->hc, NULL, &ps->transport));
    PMGCall(pmg_rowblock_sampler_create(&ps->hc, ps->transport, starts, rp, ci, v, 64, 0, NULL, ps->omega, &ps->mc, &ps->dm));
    PetscCall(PetscFree(rp));
    PetscCall(PetscFree(ci));
    PetscCall(PetscFree(v));
    PetscCall(PetscFree(starts));
  }
  PMGCall(pmg_mcsor_get_num_colors(ps->mc, &nc));
  ps->ncolors = ps->dm ? nc - 1 : nc;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ParallelSORApply(..., its, zero_initial_guess, x) (src/pc_parsor.c:703-878 as called from :886 and :902) */
static PetscErrorCode HipPARSORSweeps(PC pc, Vec b, PetscInt its, PetscBool zero_initial_guess, Vec x)
{
  PC_HipPARSOR *ps = (PC_HipPARSOR *)pc->data;
  HipVecAccess  ab, ax;

  PetscFunctionBeginUser;
  if (zero_initial_guess) PetscCall(VecZeroEntries(x));
  PetscCall(HipVecGet(b, PETSC_FALSE, &ps->bbuf, &ab));
  PetscCall(HipVecGet(x, PETSC_TRUE, &ps->xbuf, &ax));
  for (PetscInt it = 0; it < its; ++it) {
    if (ps->dm) PMGCall(pmg_distmcsor_apply(ps->dm, (int32_t)ps->nowned, ab.dev, ax.dev, (int)SOR_FORWARD_SWEEP, NULL));
    else PMGCall(pmg_mcsor_apply(ps->mc, ab.dev, ax.dev, NULL));
  }
  PetscCall(HipVecRestore(&ax, NULL));
  PetscCall(HipVecRestore(&ab, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApply_HipPARSOR(PC pc, Vec b, Vec x) /* :880-891 */
{
  PetscFunctionBeginUser;
  PetscCall(HipPARSORSweeps(pc, b, ((PC_HipPARSOR *)pc->data)->its, PETSC_TRUE, x));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipPARSORApplySOR(PC pc, Vec b, PetscInt its, PetscBool zero_initial_guess, Vec x) /* PCPARSORApplySOR, :893-906 */
{
  PetscFunctionBeginUser;
  PetscCall(HipPARSORSweeps(pc, b, its, zero_initial_guess, x));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetFromOptions_HipPARSOR(PC pc, PetscOptionItems_ARG PetscOptionsObject) /* :970-981 */
{
  PC_HipPARSOR *ps = (PC_HipPARSOR *)pc->data;

  PetscFunctionBeginUser;
  PetscOptionsHeadBegin(PetscOptionsObject, "Parallel SOR options");
  PetscCall(PetscOptionsReal("-pc_parsor_omega", "Relaxation factor", "PCPARSORSetOmega", ps->omega, &ps->omega, NULL));
  PetscCall(PetscOptionsInt("-pc_parsor_its", "Number of SOR iterations", "PCPARSORSetIterations", ps->its, &ps->its, NULL));
  PetscOptionsHeadEnd();
  if (ps->mc) PMGCall(pmg_mcsor_set_omega(ps->mc, ps->omega));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCView_HipPARSOR(PC pc, PetscViewer viewer) /* :983-997 */
{
  PC_HipPARSOR *ps = (PC_HipPARSOR *)pc->data;

  PetscFunctionBeginUser;
  PetscCall(PetscViewerASCIIPrintf(viewer, "  Omega: %g\n", (double)ps->omega));
  PetscCall(PetscViewerASCIIPrintf(viewer, "  Iterations: %" PetscInt_FMT "\n", ps->its));
  PetscCall(PetscViewerASCIIPrintf(viewer, "  Sweep type: Forward\n"));
  PetscCall(PetscViewerASCIIPrintf(viewer, "  Device sweep: libparmgmc_hip %s (%s), %s, %" PetscInt_FMT " launches per sweep\n", pmg_version(), pmg_gpu_arch(), ps->dm ? "multicolour by row blocks" : "lexicographic dependency levels", ps->ncolors));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipPARSORSetOmega(PC pc, PetscReal omega) /* PCPARSORSetOmega, :999-1009 */
{
  PC_HipPARSOR *ps = (PC_HipPARSOR *)pc->data;

  PetscFunctionBeginUser;
  ps->omega = omega;
  if (ps->mc) PMGCall(pmg_mcsor_set_omega(ps->mc, omega));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCHipPARSORSetIterations(PC pc, PetscInt its) /* PCPARSORSetIterations, :1011-1019 */
{
  PetscFunctionBeginUser;
  ((PC_HipPARSOR *)pc->data)->its = its;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_HipPARSOR(PC pc) /* :1021-1039 */
{
  PC_HipPARSOR *ps;

  PetscFunctionBeginUser;
  PetscCall(PetscNew(&ps));
  pc->data  = ps;
  ps->omega = 1.0;
  ps->its   = 1;

  pc->ops->apply          = PCApply_HipPARSOR;
  pc->ops->destroy        = PCDestroy_HipPARSOR;
  pc->ops->reset          = PCReset_HipPARSOR;
  pc->ops->setup          = PCSetUp_HipPARSOR;
  pc->ops->setfromoptions = PCSetFromOptions_HipPARSOR;
  pc->ops->view           = PCView_HipPARSOR;
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* PARMGMC_HIP_HAVE_PETSC */
typedef int parmgmc_hip_pc_hipparsor_translation_unit_not_empty;
