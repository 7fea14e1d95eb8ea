/*
 * mc_sor_hip.c -- the PETSc-typed MCSOR API of ParMGMC (reference include/parmgmc/mc_sor.h:17-30) on an MI355X.
 *
 * The ten functions the reference exports and its examples call directly -- examples/ex3.c:56-65,115-118,161 wraps an MCSOR
 * in a PCSHELL (the route BASELINE's north star names), examples/ex5.c:50-75 checks symmetric = forward o backward -- with
 * their PETSc signatures, forwarding to libparmgmc_hip's C-ABI:
 *
 *   MCSORCreate(Mat, MCSOR *)          src/mc_sor.c:618-642   opaque handle, omega = 1, forward sweep, -mc_sor_omega (:638)
 *   MCSORSetUp(MCSOR)                  src/mc_sor.c:553-605   MATSEQAIJ -> pmg_mcsor_create_csr_idx + set-up; MATMPIAIJ on
 *                                                             several ranks -> row blocks (pmg_rowblock_sampler_create, one
 *                                                             rank = one device); MATLRC over either -> + low-rank repair
 *   MCSORApply(MCSOR, Vec b, Vec y)    src/mc_sor.c:216-239   one deterministic sweep of the current type, y in/out
 *   MCSORSetOmega / SetSweepType / GetSweepType               src/mc_sor.c:412-439
 *   MCSORGetISColoring / GetNumColors  src/mc_sor.c:92-99, 607-616   the device sweep's colouring as an ISColoring
 *   MCSORDestroy(MCSOR *)              src/mc_sor.c:60-90
 *   MCSORBuildLRCCorrection(det_sor, ctx, Asor, B, S, &Bb)    src/mc_sor.c:480-544  Bb = C (S^-1 + B^T C)^-1, C = det_sor(B)
 *
 * One rank: the reference sweeps its rows in ONE colour, i.e. lexicographically (MatCreateISColoring_Seq, src/mc_sor.c:
 * 397-410).  A device sweep needs a valid colouring, so the default here is the library's LEXLEVELS rule -- the dependency
 * levels of the natural order, which reproduce the lexicographic result update for update (include/parmgmc_hip.h) --
 * and -mc_sor_hip_greedy selects the first-fit multicolouring (fewer launches, a different but equally valid sweep),
 * -mc_sor_hip_iterated first-fit followed by one round of iterated greedy (often one colour fewer still).
 * MCSORGetNumColors / MCSORGetISColoring report the colouring the device actually sweeps (the reference would say 1).
 * Several ranks: first-fit on the global matrix, rank after rank (the reference: PETSc's randomised JP, src/mc_sor.c:383-395).
 *
 * Built only inside a ParMGMC + PETSc tree with -DPARMGMC_HIP_HAVE_PETSC (adapter/README.md); empty otherwise.
 */
#ifdef PARMGMC_HIP_HAVE_PETSC
#include "hip_petsc_common.h"
#include <parmgmc/mc_sor.h>

typedef struct {
  Mat           A; /* borrowed, like the reference's ctx->A */
  PetscReal     omega;
  MatSORType    type;
  PetscBool     greedy;   /* -mc_sor_hip_greedy */
  PetscBool     iterated; /* -mc_sor_hip_iterated: first-fit + one round of iterated greedy (PMG_COLORING_ITERATED) */
  pmg_mcsor     mc;
  pmg_distmcsor dm; /* more than one rank */
  pmg_dist      transport;
  pmg_host_comm hc;
  MPI_Comm      hc_comm;
  PetscInt      nowned;
  HipStageBuf   bbuf, ybuf;
} *MCSORHip;

PetscErrorCode MCSORCreate(Mat A, MCSOR *m)
{
  MCSOR    mc;
  MCSORHip h;

  PetscFunctionBeginUser;
  PetscCall(PetscNew(&mc));
  PetscCall(PetscNew(&h));
  h->A     = A;
  h->omega = 1;
  h->type  = SOR_FORWARD_SWEEP;
  PetscCall(PetscOptionsGetReal(NULL, NULL, "-mc_sor_omega", &h->omega, NULL)); /* src/mc_sor.c:638 */
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-mc_sor_hip_greedy", &h->greedy, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-mc_sor_hip_iterated", &h->iterated, NULL));
  mc->ctx = h;
  *m      = mc;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MCSORHipRelease(MCSORHip h)
{
  PetscFunctionBeginUser;
  PMGCall(pmg_distmcsor_destroy(&h->dm));
  PMGCall(pmg_mcsor_destroy(&h->mc));
  if (h->transport) PMGCall(pmg_dist_destroy_comm(&h->hc, &h->transport)); /* collective, like MCSORDestroy on a MATMPIAIJ */
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MCSORDestroy(MCSOR *m)
{
  PetscFunctionBeginUser;
  if (!m || !*m) PetscFunctionReturn(PETSC_SUCCESS);
  MCSORHip h = (MCSORHip)(*m)->ctx;
  PetscCall(MCSORHipRelease(h));
  PetscCall(HipStageBufFree(&h->bbuf));
  PetscCall(HipStageBufFree(&h->ybuf));
  PetscCall(PetscFree(h));
  PetscCall(PetscFree(*m));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MCSORSetOmega(MCSOR m, PetscReal omega)
{
  MCSORHip h = (MCSORHip)m->ctx;

  PetscFunctionBeginUser;
  h->omega = omega;
  if (h->mc) PMGCall(pmg_mcsor_set_omega(h->mc, omega)); /* takes effect at the next sweep, as omega_changed does (src/mc_sor.c:222) */
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MCSORSetSweepType(MCSOR m, MatSORType type)
{
  MCSORHip h = (MCSORHip)m->ctx;

  PetscFunctionBeginUser;
  PetscCheck(type == SOR_FORWARD_SWEEP || type == SOR_BACKWARD_SWEEP || type == SOR_SYMMETRIC_SWEEP, PetscObjectComm((PetscObject)h->A), PETSC_ERR_SUP, "Only forward, backward and symmetric sweep supported"); /* src/mc_sor.c:427 */
  h->type = type;
  if (h->mc) PMGCall(pmg_mcsor_set_sweep_type(h->mc, (int)type));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MCSORGetSweepType(MCSOR m, MatSORType *type)
{
  PetscFunctionBeginUser;
  *type = ((MCSORHip)m->ctx)->type;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MCSORSetUp(MCSOR m)
{
  MCSORHip    h = (MCSORHip)m->ctx;
  Mat         A = h->A, Bmat = NULL;
  Vec         S = NULL;
  PetscBool   islrc, isseq, ismpi;
  PetscMPIInt size;

  PetscFunctionBeginUser;
  PetscCall(MCSORHipRelease(h)); /* MCSORSetUp may run again */
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATLRC, &islrc));
  if (islrc) PetscCall(MatLRCGetMats(h->A, &A, &Bmat, &S, NULL)); /* src/mc_sor.c:576 */
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATSEQAIJ, &isseq));
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATMPIAIJ, &ismpi));
  PetscCheck(isseq || ismpi, PetscObjectComm((PetscObject)h->A), PETSC_ERR_SUP, "Matrix type not supported"); /* src/mc_sor.c:568 */
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)A), &size));
  if (size == 1) {
    const PetscInt *ia, *ja;
    PetscScalar    *aa;
    PetscInt        n;
    Mat             Aseq = A, Ao;

    if (ismpi) PetscCall(MatMPIAIJGetSeqAIJ(A, &Aseq, &Ao, NULL)); /* one rank: the diagonal block is the matrix */
    PetscCall(MatGetSize(Aseq, &n, NULL));
    h->nowned = n;
    PetscCall(MatSeqAIJGetCSRAndMemType(Aseq, &ia, &ja, &aa, NULL)); /* borrowed host arrays, src/mc_sor.c:250 */
    PMGCall(pmg_mcsor_create_csr_idx((int64_t)n, ia, ja, aa, PMG_IDX_WIDTH, &h->mc));
    PMGCall(pmg_mcsor_set_coloring(h->mc, h->iterated ? PMG_COLORING_ITERATED : (h->greedy ? PMG_COLORING_GREEDY : PMG_COLORING_LEXLEVELS), NULL));
    PMGCall(pmg_mcsor_set_omega(h->mc, h->omega));
    PMGCall(pmg_mcsor_set_sweep_type(h->mc, (int)h->type));
    PMGCall(pmg_mcsor_setup(h->mc));
    if (islrc) { /* Bb for both directions from deterministic sweeps on the base matrix, src/mc_sor.c:578-590 */
      Mat                Abase, Bm;
      Vec                Sv;
      PetscInt           k;
      const PetscScalar *B, *Sarr;
      PetscScalar       *Bcopy;

      PetscCall(HipGetLRC(h->A, &Abase, &k, &B, &Bcopy, &Bm, &Sv));
      PetscCall(VecGetArrayRead(Sv, &Sarr));
      PMGCall(pmg_mcsor_set_lowrank(h->mc, (int32_t)k, Bcopy ? Bcopy : B, Sarr));
      PetscCall(VecRestoreArrayRead(Sv, &Sarr));
      PetscCall(MatDenseRestoreArrayRead(Bm, &B));
      PetscCall(PetscFree(Bcopy));
    }
  } else { /* one rank = one device: row blocks with one ghost update per colour (MCSORApply_MPIAIJ, src/mc_sor.c:298-381) */
    int64_t *rp, *ci, *starts;
    double  *v;

    PetscCheck(ismpi, PetscObjectComm((PetscObject)h->A), PETSC_ERR_SUP, "Matrix type not supported");
    PetscCall(HipHostComm(PetscObjectComm((PetscObject)A), &h->hc_comm, &h->hc));
    PetscCall(HipMPIAIJRows(A, &rp, &ci, &v, &starts, &h->nowned));
    PetscCall(HipCreateTransport(&h->hc, NULL, &h->transport));
    PMGCall(pmg_rowblock_sampler_create(&h->hc, h->transport, starts, rp, ci, v, 64, 0, NULL, h->omega, &h->mc, &h->dm));
    PetscCall(PetscFree(rp));
    PetscCall(PetscFree(ci));
    PetscCall(PetscFree(v));
    PetscCall(PetscFree(starts));
    if (islrc) { /* B's rows are distributed like A's: hand over the local block, zero on the ghost rows */
      PetscInt           k, lda, mloc;
      const PetscScalar *B, *Sarr;
      double            *Bl;
      int32_t            nlocal;

      PetscCall(MatGetSize(Bmat, NULL, &k));
      PetscCall(MatGetLocalSize(Bmat, &mloc, NULL));
      PetscCheck(mloc == h->nowned, PetscObjectComm((PetscObject)h->A), PETSC_ERR_ARG_SIZ, "the rows of B must be distributed like the rows of A");
      PetscCall(MatDenseGetLDA(Bmat, &lda));
      PetscCall(MatDenseGetArrayRead(Bmat, &B));
      PMGCall(pmg_mcsor_get_size(h->mc, &nlocal));
      PetscCall(PetscCalloc1((size_t)nlocal * (size_t)k, &Bl));
      for (PetscInt c = 0; c < k; ++c) PetscCall(PetscArraycpy(Bl + (size_t)nlocal * c, B + (size_t)lda * c, mloc));
      PetscCall(VecGetArrayRead(S, &Sarr));
      PMGCall(pmg_distmcsor_set_lowrank(h->dm, (int32_t)k, nlocal, (int32_t)h->nowned, Bl, Sarr));
      PetscCall(VecRestoreArrayRead(S, &Sarr));
      PetscCall(MatDenseRestoreArrayRead(Bmat, &B));
      PetscCall(PetscFree(Bl));
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* y <- one sweep of the current type on (b, y); host Vecs are staged once per call, device Vecs (VECHIP) hand over their
   arrays (hip_petsc_common.h) */
PetscErrorCode MCSORApply(MCSOR m, Vec b, Vec y)
{
  MCSORHip     h = (MCSORHip)m->ctx;
  HipVecAccess ab, ay;

  PetscFunctionBeginUser;
  PetscCheck(h->mc, PetscObjectComm((PetscObject)h->A), PETSC_ERR_ORDER, "MCSORSetUp must be called before MCSORApply");
  PetscCall(PetscLogEventBegin(MULTICOL_SOR, NULL, b, y, NULL)); /* src/mc_sor.c:221 */
  PetscCall(HipVecGet(b, PETSC_FALSE, &h->bbuf, &ab));
  PetscCall(HipVecGet(y, PETSC_TRUE, &h->ybuf, &ay));
  if (h->dm) PMGCall(pmg_distmcsor_apply(h->dm, (int32_t)h->nowned, ab.dev, ay.dev, (int)h->type, NULL));
  else PMGCall(pmg_mcsor_apply(h->mc, ab.dev, ay.dev, NULL));
  PetscCall(HipVecRestore(&ay, NULL));
  PetscCall(HipVecRestore(&ab, NULL));
  PetscCall(PetscLogEventEnd(MULTICOL_SOR, NULL, b, y, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MCSORGetNumColors(MCSOR m, PetscInt *colors)
{
  MCSORHip h = (MCSORHip)m->ctx;
  int32_t  nc;

  PetscFunctionBeginUser;
  PetscCheck(h->mc, PetscObjectComm((PetscObject)h->A), PETSC_ERR_ORDER, "MCSORSetUp must be called first");
  PMGCall(pmg_mcsor_get_num_colors(h->mc, &nc));
  *colors = h->dm ? nc - 1 : nc; /* a row block keeps its ghost rows in one more colour that is never swept */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the colouring of the owned rows as a new ISColoring (IS_COLORING_LOCAL, like src/mc_sor.c:393,408); the caller destroys it */
PetscErrorCode MCSORGetISColoring(MCSOR m, ISColoring *isc)
{
  MCSORHip         h = (MCSORHip)m->ctx;
  int32_t          nloc, *cols;
  PetscInt         nc;
  ISColoringValue *vals;

  PetscFunctionBeginUser;
  PetscCall(MCSORGetNumColors(m, &nc));
  PMGCall(pmg_mcsor_get_size(h->mc, &nloc)); /* owned rows first, then the ghost rows of a row block */
  PetscCall(PetscMalloc1((size_t)nloc + 1, &cols));
  PMGCall(pmg_mcsor_get_coloring(h->mc, cols));
  PetscCall(PetscMalloc1((size_t)h->nowned + 1, &vals));
  for (PetscInt r = 0; r < h->nowned; ++r) vals[r] = (ISColoringValue)cols[r];
  PetscCall(PetscFree(cols));
  PetscCall(ISColoringCreate(PetscObjectComm((PetscObject)h->A), nc, h->nowned, vals, PETSC_OWN_POINTER, isc));
  PetscCall(ISColoringSetType(*isc, IS_COLORING_LOCAL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* in-place inverse of a k x k column-major matrix (Gauss-Jordan, partial pivoting); nonzero return: singular */
static int SmallInverse(PetscInt k, PetscScalar *a, PetscScalar *inv)
{
  for (PetscInt i = 0; i < k * k; ++i) inv[i] = 0;
  for (PetscInt i = 0; i < k; ++i) inv[i + k * i] = 1;
  for (PetscInt c = 0; c < k; ++c) {
    PetscInt  p  = c;
    PetscReal mx = PetscAbsReal(a[c + k * c]);
    for (PetscInt r = c + 1; r < k; ++r)
      if (PetscAbsReal(a[r + k * c]) > mx) {
        mx = PetscAbsReal(a[r + k * c]);
        p  = r;
      }
    if (mx == 0) return 1;
    for (PetscInt j = 0; j < k && p != c; ++j) {
      PetscScalar t = a[c + k * j];
      a[c + k * j]  = a[p + k * j];
      a[p + k * j]  = t;
      t               = inv[c + k * j];
      inv[c + k * j]  = inv[p + k * j];
      inv[p + k * j]  = t;
    }
    const PetscScalar d = 1 / a[c + k * c];
    for (PetscInt j = 0; j < k; ++j) {
      a[c + k * j] *= d;
      inv[c + k * j] *= d;
    }
    for (PetscInt r = 0; r < k; ++r) {
      const PetscScalar f = a[r + k * c];
      if (r == c || f == 0) continue;
      for (PetscInt j = 0; j < k; ++j) {
        a[r + k * j] -= f * a[c + k * j];
        inv[r + k * j] -= f * inv[c + k * j];
      }
    }
  }
  return 0;
}

/* Bb = C (S^-1 + B^T C)^-1 with C(:, i) = det_sor(ctx, B(:, i), 0): the caller's deterministic sweep decides what M_A^-1
   is (reference src/mc_sor.c:455-544; MCGibbs hands in MCSORApply, SORGibbs MatSOR or PCPARSORApplySOR).  This generic form
   works with ANY det_sor, so the k sweeps go through Vecs; the device samplers of this adapter do not use it -- they build
   the same correction inside the library (pmg_mcsor_set_lowrank) without leaving the device.  The k x k system is solved
   exactly on every rank (the reference: KSPMatSolve, to solver tolerance); B^T C and S are summed over the ranks. */
PetscErrorCode MCSORBuildLRCCorrection(PetscErrorCode (*det_sor)(void *, Vec, Vec), void *ctx, Mat Asor, Mat B, Vec S, Mat *Bb)
{
  Mat                C;
  Vec                x;
  PetscInt           k, m, ldb, ldc, ldo, slo, shi;
  const PetscScalar *Ba, *Ca, *Sa;
  PetscScalar       *Oa, *T, *Tl, *Ti, *Sall, *Sloc;
  MPI_Comm           comm = PetscObjectComm((PetscObject)Asor);

  PetscFunctionBeginUser;
  PetscCall(MatGetSize(B, NULL, &k));
  PetscCall(MatGetLocalSize(B, &m, NULL));
  PetscCall(MatDuplicate(B, MAT_DO_NOT_COPY_VALUES, &C));
  PetscCall(MatCreateVecs(Asor, &x, NULL));
  for (PetscInt i = 0; i < k; ++i) { /* C(:, i) = M_A^-1 B(:, i): one sweep from a zero guess */
    Vec bcol, ccol;

    PetscCall(VecZeroEntries(x));
    PetscCall(MatDenseGetColumnVecRead(B, i, &bcol));
    PetscCall(det_sor(ctx, bcol, x));
    PetscCall(MatDenseRestoreColumnVecRead(B, i, &bcol));
    PetscCall(MatDenseGetColumnVecWrite(C, i, &ccol));
    PetscCall(VecCopy(x, ccol));
    PetscCall(MatDenseRestoreColumnVecWrite(C, i, &ccol));
  }
  PetscCall(VecDestroy(&x));

  /* T = S^-1 + B^T C: this rank's rows, then the sum over the ranks; S may live on any rank */
  PetscCall(PetscCalloc1((size_t)k * k + 1, &Tl));
  PetscCall(PetscCalloc1((size_t)k * k + 1, &T));
  PetscCall(PetscCalloc1((size_t)k * k + 1, &Ti));
  PetscCall(PetscCalloc1((size_t)k + 1, &Sloc));
  PetscCall(PetscCalloc1((size_t)k + 1, &Sall));
  PetscCall(MatDenseGetLDA(B, &ldb));
  PetscCall(MatDenseGetLDA(C, &ldc));
  PetscCall(MatDenseGetArrayRead(B, &Ba));
  PetscCall(MatDenseGetArrayRead(C, &Ca));
  for (PetscInt j = 0; j < k; ++j)
    for (PetscInt i = 0; i < k; ++i) {
      PetscScalar s = 0;
      for (PetscInt r = 0; r < m; ++r) s += Ba[r + (size_t)ldb * i] * Ca[r + (size_t)ldc * j];
      Tl[i + k * j] = s;
    }
  PetscCallMPI(MPI_Allreduce(Tl, T, (PetscMPIInt)(k * k), MPI_DOUBLE, MPI_SUM, comm));
  PetscCall(VecGetOwnershipRange(S, &slo, &shi));
  PetscCall(VecGetArrayRead(S, &Sa));
  for (PetscInt i = slo; i < shi; ++i) Sloc[i] = Sa[i - slo];
  PetscCall(VecRestoreArrayRead(S, &Sa));
  PetscCallMPI(MPI_Allreduce(Sloc, Sall, (PetscMPIInt)k, MPI_DOUBLE, MPI_SUM, comm));
  for (PetscInt i = 0; i < k; ++i) T[i + k * i] += 1 / Sall[i];
  PetscCheck(SmallInverse(k, T, Ti) == 0, comm, PETSC_ERR_MAT_LU_ZRPVT, "S^-1 + B^T M^-1 B is singular");

  /* Bb = C Ti on this rank's rows */
  PetscCall(MatDuplicate(B, MAT_DO_NOT_COPY_VALUES, Bb));
  PetscCall(MatDenseGetLDA(*Bb, &ldo));
  PetscCall(MatDenseGetArrayWrite(*Bb, &Oa));
  for (PetscInt j = 0; j < k; ++j)
    for (PetscInt r = 0; r < m; ++r) {
      PetscScalar s = 0;
      for (PetscInt i = 0; i < k; ++i) s += Ca[r + (size_t)ldc * i] * Ti[i + k * j];
      Oa[r + (size_t)ldo * j] = s;
    }
  PetscCall(MatDenseRestoreArrayWrite(*Bb, &Oa));
  PetscCall(MatDenseRestoreArrayRead(C, &Ca));
  PetscCall(MatDenseRestoreArrayRead(B, &Ba));
  PetscCall(MatDestroy(&C));
  PetscCall(PetscFree(Tl));
  PetscCall(PetscFree(T));
  PetscCall(PetscFree(Ti));
  PetscCall(PetscFree(Sloc));
  PetscCall(PetscFree(Sall));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* PARMGMC_HIP_HAVE_PETSC */
typedef int parmgmc_hip_mc_sor_hip_translation_unit_not_empty;
