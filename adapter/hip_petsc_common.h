/*
 * hip_petsc_common.h -- what the PETSc-side constructors of libparmgmc_hip share.
 *
 * Compiled only with -DPARMGMC_HIP_HAVE_PETSC inside a ParMGMC + PETSc tree (see adapter/README.md); PETSc is not in
 * the build image of this repository, so tests/test_adapter_syntax.py can only check that the files are valid C
 * against a declaration-only transcription of the PETSc calls they make.
 *
 * Vec staging.  The reference reads and writes host arrays (VecGetArray, reference src/mc_sor.c:252-255).  The
 * device kernels need HBM addresses:
 *   - a Vec that lives on the device (VECHIP, PETSc built --with-hip) hands its device array over unchanged;
 *   - a host Vec is staged ONCE PER CALL of apply / applyrichardson (not once per sample): b and y are copied to
 *     device buffers the PC owns, all `its` samples run there, and y is copied back at the end -- and before every
 *     sample callback, which must see the sample on the host (reference src/pc_mcgibbs.c:183).
 */
#ifndef PARMGMC_HIP_PETSC_COMMON_H
#define PARMGMC_HIP_PETSC_COMMON_H
#ifdef PARMGMC_HIP_HAVE_PETSC

#include <petsc/private/pcimpl.h> /* pc->ops, pc->data, pc->pmat: as reference src/pc_sorgibbs.c:14 */
#include <petscksp.h>
#include <petscmat.h>
#include <petscvec.h>
#include <parmgmc/parmgmc.h> /* PCRegisterSetSampleCallback, ParMGMCGetPetscRandom, PetscOptionItems_ARG, MULTICOL_SOR */
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>
#include <parmgmc_hip.h>

/* PetscCall for the C-ABI: its status codes ARE PetscErrorCode numbers (include/parmgmc_hip.h), the text comes along */
#define PMGCall(expr) \
  do { \
    int pmg_s_ = (expr); \
    PetscCheck(pmg_s_ == 0, PETSC_COMM_SELF, (PetscErrorCode)pmg_s_, "libparmgmc_hip: %s", pmg_last_error_string()); \
  } while (0)
#define PMGHip(expr) \
  do { \
    hipError_t pmg_e_ = (expr); \
    PetscCheck(pmg_e_ == hipSuccess, PETSC_COMM_SELF, PETSC_ERR_GPU, "%s: %s", #expr, hipGetErrorString(pmg_e_)); \
  } while (0)

#define PMG_IDX_WIDTH ((int)(8 * sizeof(PetscInt))) /* 32, or 64 with --with-64-bit-indices */

/* one vector checked out of PETSc for the duration of a call */
typedef struct {
  Vec          v;
  PetscScalar *arr;    /* what PETSc gave us (host or device address) */
  double      *dev;    /* what the kernels use */
  PetscBool    staged; /* dev is a copy of a host array */
  PetscBool    write;
  PetscInt     n;
} HipVecAccess;

/* growable device buffer owned by a PC */
typedef struct {
  double  *buf;
  PetscInt cap;
} HipStageBuf;

static inline PetscErrorCode HipStageBufFree(HipStageBuf *s)
{
  PetscFunctionBeginUser;
  if (s->buf) PMGHip(hipFree(s->buf));
  s->buf = NULL;
  s->cap = 0;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static inline PetscErrorCode HipVecGet(Vec v, PetscBool write, HipStageBuf *stage, HipVecAccess *a)
{
  PetscMemType mt;

  PetscFunctionBeginUser;
  a->v     = v;
  a->write = write;
  PetscCall(VecGetLocalSize(v, &a->n));
  if (write) PetscCall(VecGetArrayAndMemType(v, &a->arr, &mt));
  else PetscCall(VecGetArrayReadAndMemType(v, (const PetscScalar **)&a->arr, &mt));
  if (PetscMemTypeDevice(mt)) {
    a->dev    = (double *)a->arr;
    a->staged = PETSC_FALSE;
  } else {
    if (stage->cap < a->n) {
      if (stage->buf) PMGHip(hipFree(stage->buf));
      PMGHip(hipMalloc((void **)&stage->buf, sizeof(double) * (size_t)(a->n > 0 ? a->n : 1)));
      stage->cap = a->n;
    }
    PMGHip(hipMemcpy(stage->buf, a->arr, sizeof(double) * (size_t)a->n, hipMemcpyHostToDevice));
    a->dev    = stage->buf;
    a->staged = PETSC_TRUE;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* make the host side of a staged, written vector current (no-op for device Vecs); waits for the stream */
static inline PetscErrorCode HipVecSyncHost(HipVecAccess *a, hipStream_t stream)
{
  PetscFunctionBeginUser;
  PMGHip(hipStreamSynchronize(stream));
  if (a->staged && a->write) PMGHip(hipMemcpy(a->arr, a->dev, sizeof(double) * (size_t)a->n, hipMemcpyDeviceToHost));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static inline PetscErrorCode HipVecRestore(HipVecAccess *a, hipStream_t stream)
{
  PetscFunctionBeginUser;
  if (a->write) {
    PetscCall(HipVecSyncHost(a, stream));
    PetscCall(VecRestoreArrayAndMemType(a->v, &a->arr));
  } else {
    PetscCall(VecRestoreArrayReadAndMemType(a->v, (const PetscScalar **)&a->arr));
  }
  a->arr = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Calls the user's sample callback with y readable on the host side of the Vec, as the reference does after every
   sample (src/pc_mcgibbs.c:183, src/pc_sorgibbs.c:71, src/pc_gamgmc.c:258): hand the array back to PETSc for the
   duration of the callback, then check it out again (same addresses). */
static inline PetscErrorCode HipCallSampleCallback(PetscErrorCode (*scb)(PetscInt, Vec, void *), void *ctx, PetscInt it, HipVecAccess *y, hipStream_t stream)
{
  PetscMemType mt;

  PetscFunctionBeginUser;
  PetscCall(HipVecSyncHost(y, stream));
  PetscCall(VecRestoreArrayAndMemType(y->v, &y->arr));
  PetscCall(scb(it, y->v, ctx));
  PetscCall(VecGetArrayAndMemType(y->v, &y->arr, &mt));
  if (y->staged) PMGHip(hipMemcpy(y->dev, y->arr, sizeof(double) * (size_t)y->n, hipMemcpyHostToDevice)); /* the callback may have changed y */
  else y->dev = (double *)y->arr;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Seed of a PC's counter-based noise.  The reference's PCs all draw from ONE advancing PetscRandom (src/parmgmc.c:56-68), so
   two PCs of a process never see the same values.  The library's noise is a pure function of (seed, counter, row): a PC
   therefore gets a stream of its own -- the seed of ParMGMC's global PetscRandom (-random_seed) mixed with a process-wide
   instance number handed out in PCCreate (ParMGMCHipNextStreamId, pc_hipgamgmc.c).  PCs are created collectively and in the
   same order on every rank, so the ranks of a distributed PC agree on the number; two mcgibbs smoothers on the levels of a
   PETSc PCMG, a cholsampler beside a Gibbs PC, or two independent chains draw independent values.  (Same mixing as the
   library's own PC mirror: pc_seed in pmg_pc.c.) */
PETSC_EXTERN uint64_t ParMGMCHipNextStreamId(void);
static inline PetscErrorCode HipNoiseSeed(uint64_t stream_id, uint64_t *seed)
{
  PetscRandom r;
  PetscInt64  s;

  PetscFunctionBeginUser;
  PetscCall(ParMGMCGetPetscRandom(&r));
  PetscCall(PetscRandomGetSeed(r, &s));
  PetscCall(PetscRandomDestroy(&r)); /* drop the reference ParMGMCGetPetscRandom took */
  *seed = (uint64_t)s + 0xD1B54A32D192ED03ull * (stream_id + 1);
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- more than one rank: one rank = one device ------------------------------------------------------------------- */
/* the byte all-gather the library's row-block set-up calls back into (pmg_host_comm, include/parmgmc_hip.h) */
static inline int HipMPIAllgather(void *ctx, const void *send, int64_t nbytes, void *recv)
{
  /* MPI counts are ints: blocks of 2 GiB and more go in pieces of 1 GiB (the library pads every rank's block to the same
     length, so a large exchange of triples in the set-up of a hierarchy can get there); a piece lands at its place in
     every rank's slot of recv.  Never a silently truncated count. */
  const int64_t piece = (int64_t)1 << 30;
  MPI_Comm      comm  = *(MPI_Comm *)ctx;
  int           np    = 1;
  if (nbytes < 0) return 1;
  if (nbytes <= piece) return MPI_Allgather((void *)send, (int)nbytes, MPI_BYTE, recv, (int)nbytes, MPI_BYTE, comm) == MPI_SUCCESS ? 0 : 1;
  if (MPI_Comm_size(comm, &np) != MPI_SUCCESS) return 1;
  char *tmp = (char *)malloc((size_t)piece * (size_t)np);
  if (!tmp) return 1;
  int rc = 0;
  for (int64_t off = 0; off < nbytes && !rc; off += piece) {
    const int64_t len = nbytes - off < piece ? nbytes - off : piece;
    rc                = MPI_Allgather((const char *)send + off, (int)len, MPI_BYTE, tmp, (int)len, MPI_BYTE, comm) == MPI_SUCCESS ? 0 : 1;
    for (int r = 0; r < np && !rc; ++r) memcpy((char *)recv + (size_t)r * (size_t)nbytes + (size_t)off, tmp + (size_t)r * (size_t)len, (size_t)len);
  }
  free(tmp);
  return rc;
}

/* *store must outlive hc (it is what hc->ctx points to) */
static inline PetscErrorCode HipHostComm(MPI_Comm comm, MPI_Comm *store, pmg_host_comm *hc)
{
  PetscMPIInt rank, size;

  PetscFunctionBeginUser;
  PetscCallMPI(MPI_Comm_rank(comm, &rank));
  PetscCallMPI(MPI_Comm_size(comm, &size));
  *store        = comm;
  hc->rank      = (int32_t)rank;
  hc->nranks    = (int32_t)size;
  hc->allgather = HipMPIAllgather;
  hc->ctx       = store;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the halo / all-gather transport of the ranks of `hc`: "ipc" (hipIpc peer stores + flag words, ranks of ONE node), else RCCL.
   Both attempts are collective and agreed on inside the library, so every rank takes the same branch.  g: this rank's
   z-slab of a DMDA operator, or NULL for row blocks. */
static inline PetscErrorCode HipCreateTransport(const pmg_host_comm *hc, pmg_grid g, pmg_dist *d)
{
  PetscFunctionBeginUser;
  if (pmg_dist_create_comm(hc, "ipc", g, NULL, d) != 0) PMGCall(pmg_dist_create_comm(hc, "rccl", g, NULL, d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* This rank's rows of a MATMPIAIJ with GLOBAL column indices, entries in the order of the sequential row: the blocks of
   MatMPIAIJGetSeqAIJ (reference src/mc_sor.c:308) merged by pmg_rowblock_merge_mpiaij.  colstart: first global column of
   the diagonal block (= first owned row for a square operator, the coarse ownership start for an interpolation).  The three
   arrays are PetscMalloc'ed (caller frees); *starts (nranks + 1, PetscMalloc'ed) = MatGetOwnershipRanges as int64. */
static inline PetscErrorCode HipMPIAIJRows(Mat A, int64_t **rp, int64_t **ci, double **v, int64_t **starts, PetscInt *nloc)
{
  Mat             Ad, Ao;
  const PetscInt *garray, *ia, *ja, *ib, *jb, *ranges;
  PetscScalar    *aa, *ab;
  PetscInt        m, cstart;
  PetscMPIInt     size;

  PetscFunctionBeginUser;
  PetscCall(MatMPIAIJGetSeqAIJ(A, &Ad, &Ao, &garray));
  PetscCall(MatSeqAIJGetCSRAndMemType(Ad, &ia, &ja, &aa, NULL));
  PetscCall(MatSeqAIJGetCSRAndMemType(Ao, &ib, &jb, &ab, NULL));
  PetscCall(MatGetLocalSize(A, &m, NULL));
  PetscCall(MatGetOwnershipRangeColumn(A, &cstart, NULL));
  PetscCall(PetscMalloc1((size_t)m + 1, rp));
  PetscCall(PetscMalloc1((size_t)(ia[m] + ib[m]) + 1, ci));
  PetscCall(PetscMalloc1((size_t)(ia[m] + ib[m]) + 1, v));
  PMGCall(pmg_rowblock_merge_mpiaij((int32_t)m, (int64_t)cstart, ia, ja, aa, ib, jb, ab, garray, PMG_IDX_WIDTH, PMG_ROWBLOCK_ORDER_GLOBAL, *rp, *ci, *v));
  if (starts) {
    PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)A), &size));
    PetscCall(MatGetOwnershipRanges(A, &ranges));
    PetscCall(PetscMalloc1((size_t)size + 1, starts));
    for (PetscMPIInt r = 0; r <= size; ++r) (*starts)[r] = (int64_t)ranges[r];
  }
  if (nloc) *nloc = m;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MATLRC pieces as host arrays: B dense n x k (column-major, leading dimension n) and S (k) -- MatLRCGetMats as
   reference src/mc_sor.c:576.  *Bcopy is PetscMalloc'ed when B's leading dimension is not n (caller frees). */
static inline PetscErrorCode HipGetLRC(Mat lrc, Mat *A, PetscInt *k, const PetscScalar **B, PetscScalar **Bcopy, Mat *Bmat, Vec *S)
{
  PetscInt n, lda;

  PetscFunctionBeginUser;
  *Bcopy = NULL;
  PetscCall(MatLRCGetMats(lrc, A, Bmat, S, NULL));
  PetscCall(MatGetSize(*Bmat, &n, k));
  PetscCall(MatDenseGetLDA(*Bmat, &lda));
  PetscCall(MatDenseGetArrayRead(*Bmat, B));
  if (lda != n) {
    PetscCall(PetscMalloc1((size_t)n * (size_t)*k, Bcopy));
    for (PetscInt c = 0; c < *k; ++c) PetscCall(PetscArraycpy(*Bcopy + (size_t)n * c, *B + (size_t)lda * c, n));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode PCCreate_HipSORGibbs(PC);
PETSC_EXTERN PetscErrorCode PCCreate_HipMulticolorGibbs(PC);
PETSC_EXTERN PetscErrorCode PCCreate_HipGAMGMC(PC);
PETSC_EXTERN PetscErrorCode PCCreate_HipCholSampler(PC);
PETSC_EXTERN PetscErrorCode PCCreate_HipPARSOR(PC);
PETSC_EXTERN PetscErrorCode PCCreate_HipWoodbury(PC);
/* PCRegister of the six constructors under the reference's six type names "sorgibbs", "mcgibbs", "gamgmc", "cholsampler",
   "parsor", "woodbury" (include/parmgmc/parmgmc.h:26-31): call it from ParMGMCRegisterPCAll (reference src/parmgmc.c:44-54)
   in place of the CPU constructors */
PETSC_EXTERN PetscErrorCode ParMGMCHipRegisterPCAll(void);

#endif /* PARMGMC_HIP_HAVE_PETSC */
#endif
