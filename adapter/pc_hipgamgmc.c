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
  uint64_t    seed, counter, stream_id;
  HipStageBuf bbuf, ybuf;
  /* more than one rank (one rank = one device): row blocks of MATMPIAIJ levels, or z-slabs of a DMDA */
  pmg_dist      transport;
  pmg_grid      slab;
  pmg_host_comm hc;
  MPI_Comm      hc_comm;
  pmg_rbh       rbh;            /* row-block builder, alive from PCGAMGMC_SetUpHierarchy to pmg_mgmc_setup */
  PetscInt      nowned, nlocal; /* row blocks: the library's fine-level vectors carry one entry per LOCAL row (owned, then ghosts) */
  HipStageBuf   bpad, ypad;

  void *cbctx;
  PetscErrorCode (*scb)(PetscInt, Vec, void *);
  PetscErrorCode (*del_scb)(void *);
} PC_HipGAMGMC;

/* context of the library's per-sample callback: gives the sample to the user's PETSc callback as a Vec */
typedef struct {
  PC_HipGAMGMC  *pg;
  HipVecAccess  *y;
  double        *ypad; /* row blocks: the library's vector (owned rows, then ghost rows); NULL: it works on y itself */
  PetscErrorCode ierr;
} HipTrampoline;

static int HipSampleTrampoline(int32_t it, const double *y_nat_dev, int32_t n, void *ctx)
{
  HipTrampoline *t = (HipTrampoline *)ctx;
  (void)y_nat_dev; /* == t->y->dev (or t->ypad): pmg_mgmc_sample writes the sample into the caller's y before it calls back */
  (void)n;
  if (t->ypad && hipMemcpy(t->y->dev, t->ypad, sizeof(double) * (size_t)t->pg->nowned, hipMemcpyDeviceToDevice) != hipSuccess) return PETSC_ERR_GPU;
  t->ierr = HipCallSampleCallback(t->pg->scb, t->pg->cbctx, (PetscInt)it, t->y, NULL); /* pg->scb(it, y, ctx), src/pc_gamgmc.c:258 */
  if (!t->ierr && t->ypad && hipMemcpy(t->ypad, t->y->dev, sizeof(double) * (size_t)t->pg->nowned, hipMemcpyDeviceToDevice) != hipSuccess) return PETSC_ERR_GPU; /* the callback may have changed y */
  return t->ierr ? (int)t->ierr : 0;
}

static PetscErrorCode HipGAMGMCRelease(PC_HipGAMGMC *pg)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_mgmc_destroy(&pg->h)); /* before the transport and the slab it borrows */
  pmg_rbh_destroy(&pg->rbh);
  if (pg->transport) PMGCall(pmg_dist_destroy_comm(&pg->hc, &pg->transport));
  PMGCall(pmg_grid_destroy(&pg->slab));
  PetscCall(HipStageBufFree(&pg->bbuf));
  PetscCall(HipStageBufFree(&pg->ybuf));
  PetscCall(HipStageBufFree(&pg->bpad));
  PetscCall(HipStageBufFree(&pg->ypad));
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
  PetscCheck(isseq, PetscObjectComm((PetscObject)A), PETSC_ERR_SUP, "%s of level %" PetscInt_FMT " is not MATSEQAIJ", what, l);
  PetscCall(MatSeqAIJGetCSRAndMemType(A, ia, ja, aa, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- more than one rank ------------------------------------------------------------------------------------------ */
/* Is pc->dm a 3-D DMDA split in z only, and `A` the operator of MatAssembleShiftedLaplaceFD on it (reference src/problems.c:
   14-75: off-diagonals -h2, diagonal kappa^2 + (#neighbours) h2, h2 = 1/(nx-1)^2)?  Then the hierarchy is the library's own
   DMDA hierarchy on z-slabs (matrix-free fine level, class-stencil Galerkin levels: pmg_mgmc_create_dmda_slab) and *kappa
   is read off the first local row.  Every rank checks ALL its rows; the verdict is agreed with an all-reduce. */
static PetscErrorCode HipDetectDMDASlab(PC pc, PC mg, PetscInt levels, Mat A, PetscBool *yes, PetscInt dims[3], PetscInt *zs, PetscInt *zm, PetscReal *kappa)
{
  PetscBool       isda = PETSC_FALSE;
  PetscInt        dim, M, N, P, m, n, p, dof, sw, xs, ys, xm, ym, rstart, rend;
  PetscMPIInt     ok = 1, all = 0;
  DMDAStencilType st;
  DMBoundaryType  bx, by, bz;

  PetscFunctionBeginUser;
  *yes = PETSC_FALSE;
  if (pc->dm) PetscCall(PetscObjectTypeCompare((PetscObject)pc->dm, DMDA, &isda));
  if (!isda) ok = 0;
  if (ok) {
    PetscCall(DMDAGetInfo(pc->dm, &dim, &M, &N, &P, &m, &n, &p, &dof, &sw, &bx, &by, &bz, &st));
    if (dim != 3 || dof != 1 || m != 1 || n != 1) ok = 0;
    /* The slab path REPLACES PETSc's PCMG hierarchy with the library's own (Q1 interpolation between vertex grids
       n -> (n-1)/2+1, R = P^T, Galerkin coarse operators), while the one-rank path hands over the level matrices PETSc
       actually built.  So that the same options mean the same sampler on 1 and on N ranks, the path is taken only where
       PETSc's hierarchy IS that one: star stencil of width 1 without periodic / ghosted boundaries, Q1 interpolation
       (DMDA's default; -da_interp_type q0 would be piecewise constant), Galerkin coarse operators, and extents that halve
       (levels - 1) times.  Anything else goes to the general row-block path (HipGAMGMCRowBlocks), which uses PETSc's own
       level matrices -- advisor finding, round 3. */
    if (ok) {
      DMDAInterpolationType itype;
      PCMGGalerkinType      gal;
      const PetscInt        div = (PetscInt)1 << (levels > 1 ? levels - 1 : 0);
      PetscCall(DMDAGetInterpolationType(pc->dm, &itype));
      PetscCall(PCMGGetGalerkin(mg, &gal));
      if (st != DMDA_STENCIL_STAR || sw != 1 || bx != DM_BOUNDARY_NONE || by != DM_BOUNDARY_NONE || bz != DM_BOUNDARY_NONE) ok = 0;
      if (itype != DMDA_Q1 || gal == PC_MG_GALERKIN_NONE) ok = 0;
      if (levels < 1 || ((M - 1) % div) || ((N - 1) % div) || ((P - 1) % div)) ok = 0;
    }
  }
  if (ok) {
    const PetscReal h2 = 1.0 / (PetscReal)((M - 1) * (M - 1));
    PetscReal       k2 = -1.0;
    PetscCall(DMDAGetCorners(pc->dm, &xs, &ys, zs, &xm, &ym, zm));
    PetscCall(MatGetOwnershipRange(A, &rstart, &rend));
    if (rend - rstart != M * N * (*zm)) ok = 0;
    for (PetscInt r = rstart; r < rend && ok; ++r) { /* z-slabs: PETSc's ordering of the owned points IS the natural order of the slab */
      const PetscInt     q = r - rstart, i = q % M, j = (q / M) % N, k = *zs + q / (M * N);
      const PetscInt     nnb = (i > 0) + (i < M - 1) + (j > 0) + (j < N - 1) + (k > 0) + (k < P - 1);
      PetscInt           ncols;
      const PetscInt    *cols;
      const PetscScalar *vals;
      PetscCall(MatGetRow(A, r, &ncols, &cols, &vals));
      if (ncols != nnb + 1) ok = 0;
      for (PetscInt c = 0; c < ncols && ok; ++c) {
        if (cols[c] == r) {
          const PetscReal kk = vals[c] - (PetscReal)nnb * h2;
          if (k2 < 0) k2 = kk;
          else if (PetscAbsReal(kk - k2) > 1e-12 * PetscAbsReal(vals[c])) ok = 0;
        } else if (vals[c] != -h2) ok = 0;
      }
      PetscCall(MatRestoreRow(A, r, &ncols, &cols, &vals));
    }
    if (k2 < 0) ok = 0;
    *kappa  = ok ? PetscSqrtReal(k2) : 0;
    dims[0] = M, dims[1] = N, dims[2] = P;
  }
  PetscCallMPI(MPI_Allreduce(&ok, &all, 1, MPI_INT, MPI_MIN, PetscObjectComm((PetscObject)pc)));
  *yes = (PetscBool)(all == 1);
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* PCGAMGMC_SetUpHierarchy on MATMPIAIJ levels (src/pc_gamgmc.c:157-223): every level operator and interpolation PCMG / PCGAMG
   built is handed to the library as this rank's rows with global columns; the library replicates the small levels, colours
   the others, builds the rows of P^T each rank owns and the ghost plans (pmg_rbh_*), all through MPI_Allgather */
static PetscErrorCode HipGAMGMCRowBlocks(PC pc, PetscInt levels)
{
  PC_HipGAMGMC *pg = (PC_HipGAMGMC *)pc->data;
  pmg_rbh       rbh = NULL;
  PetscInt      repl = 50000;
  const char   *prefix;

  PetscFunctionBeginUser;
  PetscCall(PCGetOptionsPrefix(pc, &prefix));
  PetscCall(PetscOptionsGetInt(NULL, prefix, "-pc_gamgmc_hip_replicate_below", &repl, NULL)); /* levels with at most that many rows run redundantly on every rank */
  PetscCall(HipCreateTransport(&pg->hc, NULL, &pg->transport));
  PMGCall(pmg_rbh_create(&pg->hc, (int32_t)levels, (int64_t)repl, &rbh));
  {
    PetscBool iterated = PETSC_FALSE; /* first-fit + one round of iterated greedy on every level: one colour (and one ghost update per sweep) fewer on P1 hierarchies */
    PetscCall(PetscOptionsGetBool(NULL, prefix, "-pc_gamgmc_hip_iterated_coloring", &iterated, NULL));
    if (iterated) PMGCall(pmg_rbh_set_coloring(rbh, PMG_COLORING_ITERATED));
  }
  for (PetscInt l = 0; l < levels; ++l) {
    KSP      ksp;
    PC       pcl;
    Mat      A, Ip;
    int64_t *rp, *ci;
    double  *v;
    PetscInt nloc, n, rstart;

    PetscCall(PCMGGetSmoother(pg->mg, l, &ksp));
    PetscCall(KSPGetPC(ksp, &pcl));
    PetscCall(PCGetOperators(pcl, NULL, &A));
    PetscCall(MatGetSize(A, &n, NULL));
    PetscCall(MatGetOwnershipRange(A, &rstart, NULL));
    PetscCall(HipMPIAIJRows(A, &rp, &ci, &v, NULL, &nloc));
    PMGCall(pmg_rbh_set_level_operator(rbh, (int32_t)l, (int64_t)n, (int64_t)rstart, (int64_t)nloc, rp, ci, v, 64)); /* copied */
    PetscCall(PetscFree(rp));
    PetscCall(PetscFree(ci));
    PetscCall(PetscFree(v));
    if (l > 0) {
      PetscCall(PCMGGetInterpolation(pg->mg, l, &Ip));
      PetscCall(HipMPIAIJRows(Ip, &rp, &ci, &v, NULL, &nloc));
      PMGCall(pmg_rbh_set_level_interpolation(rbh, (int32_t)l, (int64_t)nloc, rp, ci, v, 64));
      PetscCall(PetscFree(rp));
      PetscCall(PetscFree(ci));
      PetscCall(PetscFree(v));
    }
    if (l == levels - 1) pg->nowned = nloc;
    PetscCall(PCReset(pcl)); /* the level sampler PETSc created is never applied */
  }
  PMGCall(pmg_rbh_build(rbh));
  PMGCall(pmg_rbh_create_mgmc(rbh, pg->transport, &pg->h));
  {
    pmg_rbh_level_view view;
    PMGCall(pmg_rbh_get_level(rbh, (int32_t)levels - 1, &view));
    pg->nlocal = view.nlocal;
  }
  pg->rbh = rbh; /* its arrays stay borrowed until pmg_mgmc_setup: PCSetUp destroys it afterwards */
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
  PetscMPIInt   size;
  PetscBool     slab = PETSC_FALSE;

  PetscFunctionBeginUser;
  PMGCall(pmg_mgmc_destroy(&pg->h));
  pmg_rbh_destroy(&pg->rbh);
  if (pg->transport) PMGCall(pmg_dist_destroy_comm(&pg->hc, &pg->transport));
  PMGCall(pmg_grid_destroy(&pg->slab));
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)pc), &size));
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
  pg->nowned = pg->nlocal = 0;
  if (size > 1) { /* one rank = one device */
    PetscInt  dims[3], zs, zm;
    PetscReal kappa;

    PetscCall(HipHostComm(PetscObjectComm((PetscObject)pc), &pg->hc_comm, &pg->hc));
    if (strcmp(pg->mgtype, PCMG) == 0) PetscCall(HipDetectDMDASlab(pc, pg->mg, levels, P, &slab, dims, &zs, &zm, &kappa));
    if (slab) { /* DMDA split in z: the library's own hierarchy on z-slabs, halo exchange over xGMI */
      int32_t *cuts, z0 = (int32_t)zs;
      PetscCall(PetscMalloc1((size_t)size + 1, &cuts));
      PetscCallMPI(MPI_Allgather(&z0, 1, MPI_INT, cuts, 1, MPI_INT, PetscObjectComm((PetscObject)pc)));
      cuts[size] = (int32_t)dims[2];
      PMGCall(pmg_grid_create((int32_t)dims[0], (int32_t)dims[1], (int32_t)dims[2], (int32_t)zs, (int32_t)zm, kappa, &pg->slab));
      PetscCall(HipCreateTransport(&pg->hc, pg->slab, &pg->transport));
      PMGCall(pmg_mgmc_create_dmda_slab((int32_t)dims[0], (int32_t)dims[1], (int32_t)dims[2], kappa, (int32_t)levels, pg->slab, pg->transport, cuts, &pg->h));
      PetscCall(PetscFree(cuts));
      pg->nowned = pg->nlocal = dims[0] * dims[1] * zm;
    } else PetscCall(HipGAMGMCRowBlocks(pc, levels)); /* MATMPIAIJ levels by row blocks */
  } else {
    PetscBool   iterated = PETSC_FALSE;
    const char *pfx;
    PMGCall(pmg_mgmc_create_hierarchy((int32_t)levels, &pg->h));
    PetscCall(PCGetOptionsPrefix(pc, &pfx));
    PetscCall(PetscOptionsGetBool(NULL, pfx, "-pc_gamgmc_hip_iterated_coloring", &iterated, NULL));
    if (iterated) PMGCall(pmg_mgmc_set_coloring(pg->h, PMG_COLORING_ITERATED));
  }
  for (PetscInt l = 0; l < levels && size == 1; ++l) {
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

    if (size == 1) {
      PetscCall(HipGetLRC(pc->pmat, &Abase, &k, &B, &Bcopy, &Bmat, &S));
      PetscCall(VecGetArrayRead(S, &Sarr));
      PMGCall(pmg_mgmc_set_lowrank(pg->h, (int32_t)k, Bcopy ? Bcopy : B, Sarr));
    } else { /* this rank's rows of the dense MPI matrix B, one column of nlocal entries each (ghost entries zero) */
      PetscInt lda, mloc;
      PetscCall(MatLRCGetMats(pc->pmat, &Abase, &Bmat, &S, NULL));
      PetscCall(MatGetSize(Bmat, NULL, &k));
      PetscCall(MatGetLocalSize(Bmat, &mloc, NULL));
      PetscCheck(mloc == pg->nowned, PetscObjectComm((PetscObject)pc), PETSC_ERR_ARG_SIZ, "the rows of B must be distributed like the rows of A");
      PetscCall(MatDenseGetLDA(Bmat, &lda));
      PetscCall(MatDenseGetArrayRead(Bmat, &B));
      PetscCall(PetscCalloc1((size_t)pg->nlocal * (size_t)k, &Bcopy));
      for (PetscInt c = 0; c < k; ++c) PetscCall(PetscArraycpy(Bcopy + (size_t)pg->nlocal * c, B + (size_t)lda * c, mloc));
      PetscCall(VecGetArrayRead(S, &Sarr));
      PMGCall(pmg_mgmc_set_lowrank(pg->h, (int32_t)k, Bcopy, Sarr));
    }
    PetscCall(VecRestoreArrayRead(S, &Sarr));
    PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
    PetscCall(PetscFree(Bcopy));
  }
  PMGCall(pmg_mgmc_setup(pg->h)); /* colours and uploads every level; the PETSc matrices are no longer read afterwards */
  pmg_rbh_destroy(&pg->rbh);
  PetscCall(HipNoiseSeed(pg->stream_id, &pg->seed));
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
  tr.ypad = NULL;
  tr.ierr = PETSC_SUCCESS;
  {
    const double *bdev = ab.dev;
    double       *ydev = ay.dev;
    if (pg->nlocal > pg->nowned) { /* row blocks: one entry per local row of the finest level, the ghost entries are the library's */
      for (int q = 0; q < 2; ++q) {
        HipStageBuf *sb = q ? &pg->ypad : &pg->bpad;
        if (sb->cap < pg->nlocal) {
          PetscCall(HipStageBufFree(sb));
          PMGHip(hipMalloc((void **)&sb->buf, sizeof(double) * (size_t)pg->nlocal));
          PMGHip(hipMemset(sb->buf, 0, sizeof(double) * (size_t)pg->nlocal));
          sb->cap = pg->nlocal;
        }
      }
      PMGHip(hipMemcpy(pg->bpad.buf, ab.dev, sizeof(double) * (size_t)pg->nowned, hipMemcpyDeviceToDevice));
      PMGHip(hipMemcpy(pg->ypad.buf, ay.dev, sizeof(double) * (size_t)pg->nowned, hipMemcpyDeviceToDevice));
      bdev = pg->bpad.buf, ydev = tr.ypad = pg->ypad.buf;
    }
    const int rc = pmg_mgmc_sample(pg->h, bdev, ydev, (int32_t)its, (int)guesszero, pg->seed, pg->counter, &pg->counter, pg->scb ? HipSampleTrampoline : NULL, &tr, NULL);
    PetscCall(tr.ierr); /* an error raised inside the user's callback keeps its own stack */
    PMGCall(rc);
    if (tr.ypad) PMGHip(hipMemcpy(ay.dev, tr.ypad, sizeof(double) * (size_t)pg->nowned, hipMemcpyDeviceToDevice));
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
  pg->stream_id = ParMGMCHipNextStreamId();
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

/* ---- registration: the body of ParMGMCRegisterPCAll (reference src/parmgmc.c:44-54), all six type names ----------- */
static uint64_t parmgmc_hip_stream_ids; /* process-wide: one noise stream per PC instance (hip_petsc_common.h, HipNoiseSeed) */
uint64_t        ParMGMCHipNextStreamId(void) { return parmgmc_hip_stream_ids++; }

PetscErrorCode ParMGMCHipRegisterPCAll(void)
{
  PetscFunctionBeginUser;
  PetscCall(PCRegister(PCSORGIBBS, PCCreate_HipSORGibbs));
  PetscCall(PCRegister(PCMCGIBBS, PCCreate_HipMulticolorGibbs));
  PetscCall(PCRegister(PCGAMGMC, PCCreate_HipGAMGMC));
  PetscCall(PCRegister(PCCHOLSAMPLER, PCCreate_HipCholSampler));
  PetscCall(PCRegister(PCPARSOR, PCCreate_HipPARSOR));
  PetscCall(PCRegister(PCWOODBURY, PCCreate_HipWoodbury));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* PARMGMC_HIP_HAVE_PETSC */
typedef int parmgmc_hip_pc_hipgamgmc_translation_unit_not_empty;
