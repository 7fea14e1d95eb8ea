/* The Woodbury term of PCWOODBURY (reference src/woodbury.c) as an object of its own -- host side (C11).
 *
 * PCWOODBURY samples from N((A + B S B^T)^-1 f, (A + B S B^T)^-1) with ANY sampler of A and any solver:
 *   set-up  (PCWoodburyBuildLRCCorrection, src/woodbury.c:21-91):  C = solver(B) column by column from a zero guess (:35-50),
 *           T = S^-1 + B^T C (:53-68), G = C T^-1 (:70-78);
 *   sample  (PCApplyRichardson_Woodbury, :263-289):  w = b + B (sqrt|S| o xi) (:275-277), y <- one sample of the A-sampler
 *           on w (:278), y -= G (B^T y) (:280-282).
 * The sampler and the solver are the caller's (the library's own PC mirror in pmg_pc.c, a PETSc PC in adapter/, a
 * distributed sampler in parmgmc_amd/dist.py); this object holds B, G and the k-vectors on the device and does the dense
 * products.  ROW-DISTRIBUTED operators (the reference's normal mode: B is a dense MPI matrix whose rows follow A's,
 * MatTransposeMatMult / MatMultTranspose reduce over the ranks): every rank holds ITS rows of B -- z-slabs of a DMDA or row
 * blocks of a MATMPIAIJ alike, the object only sees natural-order device vectors of the owned rows -- and the k x k
 * product and every B^T y are summed over the ranks in rank order through the transport of a pmg_dist object
 * (pmg_dist_allreduce_sum: identical bits on every rank); the k x k inverse and the noise k-vector are replicated. */
#include "pmg_internal.h"
#include <math.h>
#include <stdlib.h>

struct pmg_woodbury_s {
  int64_t  n;  /* rows on this rank */
  int32_t  k;
  pmg_dist dist; /* NULL: one device (borrowed) */
  double  *B, *C, *G;         /* device, n x k column-major; C is released by pmg_woodbury_finish */
  double  *S_sqrt, *wk, *partial; /* device: sqrt|S| (k), k-vector (k*k while T is formed), block sums */
  double   S[64];
  int      finished;
};

pmg_status pmg_woodbury_destroy(pmg_woodbury *wp)
{
  if (!wp || !*wp) return PMG_SUCCESS;
  pmg_woodbury w = *wp;
  pmg_dev_free(w->B), pmg_dev_free(w->C), pmg_dev_free(w->G), pmg_dev_free(w->S_sqrt), pmg_dev_free(w->wk), pmg_dev_free(w->partial);
  free(w);
  *wp = NULL;
  return PMG_SUCCESS;
}

/* B_host: this rank's n rows of B, k columns, column-major with leading dimension ldb >= n (MatDenseGetLDA); S_host: the
   k diagonal entries of S = Sigma^-1 (MatLRCGetMats, src/woodbury.c:162).  dist: NULL on one device, else any pmg_dist
   object of the ranks that share the rows (borrowed; only its all-reduce is used). */
pmg_status pmg_woodbury_create(int64_t n, int32_t k, const double *B_host, int64_t ldb, const double *S_host, pmg_dist dist, pmg_woodbury *out)
{
  PMG_CHECK(out && S_host && (n == 0 || B_host), PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(k >= 1 && k <= 64, PMG_ERR_ARG_OUTOFRANGE, "rank k = %d (1..64 supported)", k);
  PMG_CHECK(n >= 0 && ldb >= n, PMG_ERR_ARG_OUTOFRANGE, "n = %lld, leading dimension %lld", (long long)n, (long long)ldb);
  *out           = NULL;
  pmg_woodbury w = (pmg_woodbury)calloc(1, sizeof *w);
  PMG_CHECK(w, PMG_ERR_MEM, "out of host memory");
  w->n = n, w->k = k, w->dist = dist;
  double sq[64];
  for (int c = 0; c < k; ++c) w->S[c] = S_host[c], sq[c] = sqrt(fabs(S_host[c])); /* VecSqrtAbs, src/woodbury.c:177 */
  const size_t nb = sizeof(double) * (size_t)(n > 0 ? n : 1) * (size_t)k;
  pmg_status   st = pmg_dev_alloc((void **)&w->B, nb);
  for (int c = 0; c < k && !st && n; ++c)
    if (hipMemcpy(w->B + (size_t)n * c, B_host + (size_t)ldb * c, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "upload failed");
  if (!st) st = pmg_dev_alloc((void **)&w->C, nb);
  if (!st) st = pmg_dev_alloc((void **)&w->G, nb);
  if (!st) st = pmg_dev_upload((void **)&w->S_sqrt, sq, sizeof(double) * (size_t)k);
  if (!st) st = pmg_dev_alloc((void **)&w->wk, sizeof(double) * 64 * 64);
  if (!st) st = pmg_dev_alloc((void **)&w->partial, sizeof(double) * (size_t)pmgk_lrc_nblocks(n > 0 ? n : 1) * (size_t)k);
  if (st) {
    pmg_woodbury_destroy(&w);
    return st;
  }
  *out = w;
  return PMG_SUCCESS;
}

/* column c of B (read) and of C (to be written by the caller's solver from a ZERO guess: it is zero-filled here), device
   pointers borrowed from the object: C(:,c) = solver(B(:,c)), src/woodbury.c:39-49 */
pmg_status pmg_woodbury_column(pmg_woodbury w, int32_t c, const double **B_col_dev, double **C_col_dev, void *stream)
{
  PMG_CHECK(w && B_col_dev && C_col_dev, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(!w->finished, PMG_ERR_ARG_WRONGSTATE, "the correction is already built");
  PMG_CHECK(c >= 0 && c < w->k, PMG_ERR_ARG_OUTOFRANGE, "column %d of %d", c, w->k);
  *B_col_dev = w->B + (size_t)w->n * c;
  *C_col_dev = w->C + (size_t)w->n * c;
  if (w->n) PMG_HIP(hipMemsetAsync(*C_col_dev, 0, sizeof(double) * (size_t)w->n, (hipStream_t)stream)); /* VecZeroEntries(x), :42 */
  return PMG_SUCCESS;
}

/* C(:,c) <- a solver result the caller holds in a vector of its own (device, n values): what the loop of
   src/woodbury.c:46-48 does with VecCopy(x, c) */
pmg_status pmg_woodbury_set_c_column(pmg_woodbury w, int32_t c, const double *x_dev, void *stream)
{
  PMG_CHECK(w && (w->n == 0 || x_dev), PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(!w->finished, PMG_ERR_ARG_WRONGSTATE, "the correction is already built");
  PMG_CHECK(c >= 0 && c < w->k, PMG_ERR_ARG_OUTOFRANGE, "column %d of %d", c, w->k);
  if (w->n) PMG_HIP(hipMemcpyAsync(w->C + (size_t)w->n * c, x_dev, sizeof(double) * (size_t)w->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PMG_SUCCESS;
}

/* after every column of C has been written: T = S^-1 + B^T C (summed over the ranks), G = C T^-1; releases C.
   Collective over the ranks of `dist`.  Synchronous. */
pmg_status pmg_woodbury_finish(pmg_woodbury w)
{
  PMG_CHECK(w, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!w->finished, PMG_ERR_ARG_WRONGSTATE, "the correction is already built");
  const int     k = w->k;
  const int64_t n = w->n;
  for (int c = 0; c < k; ++c) { /* T(:,c) = B^T C(:,c), src/woodbury.c:53; an empty rank contributes zeros */
    if (n) PMG_KERNEL(pmgk_lrc_btx(n, k, w->B, n, w->C + (size_t)n * c, w->partial, NULL, w->wk + (size_t)k * c, NULL));
    else PMG_HIP(hipMemsetAsync(w->wk + (size_t)k * c, 0, sizeof(double) * (size_t)k, NULL));
  }
  if (w->dist) PMG_CALL(pmg_dist_allreduce_sum(w->dist, w->wk, k * k, NULL));
  double T[64 * 64], Sb[64 * 64], *Sb_dev = NULL;
  PMG_HIP(hipMemcpy(T, w->wk, sizeof(double) * (size_t)k * k, hipMemcpyDeviceToHost));
  for (int c = 0; c < k; ++c) T[c + (size_t)k * c] += 1.0 / w->S[c]; /* + S^-1, :66-68 */
  PMG_CHECK(!pmg_invert_small(k, T, Sb), PMG_ERR_LIB, "S^-1 + B^T M^-1 B is singular");
  PMG_CALL(pmg_dev_upload((void **)&Sb_dev, Sb, sizeof(double) * (size_t)k * k));
  pmg_status st = PMG_SUCCESS;
  if (n && pmgk_lrc_gemm_small(n, k, w->C, n, Sb_dev, w->G, NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "kernel launch failed"); /* G = C Sb, :78 */
  if (!st && hipDeviceSynchronize() != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "device error while building the Woodbury correction");
  pmg_dev_free(Sb_dev);
  PMG_CALL(st);
  pmg_dev_free(w->C);
  w->C        = NULL;
  w->finished = 1;
  return PMG_SUCCESS;
}

/* w = b + B (sqrt|S| o xi), xi = the k row-stream normals of (seed, counter) -- the same k-vector on every rank
   (src/woodbury.c:275-277) */
pmg_status pmg_woodbury_noisy_rhs(pmg_woodbury w, const double *b_dev, double *w_dev, uint64_t seed, uint64_t counter, void *stream)
{
  PMG_CHECK(w && (w->n == 0 || (b_dev && w_dev)), PMG_ERR_ARG_NULL, "null argument");
  PMG_KERNEL(pmgk_fill_normal_rows_scaled(w->k, seed, counter, w->S_sqrt, w->wk, stream)); /* VecSetRandomStandardNormal(wb->wk), VecPointwiseMult */
  if (w->n) PMG_KERNEL(pmgk_lrc_axpy_cols(w->n, w->k, w->B, w->n, w->wk, 1.0, b_dev, w_dev, stream)); /* MatMultAdd(B, wk, b, w) */
  return PMG_SUCCESS;
}

/* y -= G (B^T y), src/woodbury.c:280-282; B^T y summed over the ranks.  Collective. */
pmg_status pmg_woodbury_correct(pmg_woodbury w, double *y_dev, void *stream)
{
  PMG_CHECK(w && (w->n == 0 || y_dev), PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(w->finished, PMG_ERR_ARG_WRONGSTATE, "call pmg_woodbury_finish first");
  if (w->n) PMG_KERNEL(pmgk_lrc_btx(w->n, w->k, w->B, w->n, y_dev, w->partial, NULL, w->wk, stream)); /* wk = B^T y */
  else PMG_HIP(hipMemsetAsync(w->wk, 0, sizeof(double) * (size_t)w->k, (hipStream_t)stream));
  if (w->dist) PMG_CALL(pmg_dist_allreduce_sum(w->dist, w->wk, w->k, stream));
  if (w->n) PMG_KERNEL(pmgk_lrc_axpy_cols(w->n, w->k, w->G, w->n, w->wk, -1.0, y_dev, y_dev, stream)); /* y -= G wk */
  return PMG_SUCCESS;
}

/* G (n x k column-major, this rank's rows) to the host: diagnostics / tests */
pmg_status pmg_woodbury_get_correction(pmg_woodbury w, double *G_host)
{
  PMG_CHECK(w && G_host, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(w->finished, PMG_ERR_ARG_WRONGSTATE, "call pmg_woodbury_finish first");
  if (w->n) PMG_HIP(hipMemcpy(G_host, w->G, sizeof(double) * (size_t)w->n * (size_t)w->k, hipMemcpyDeviceToHost));
  return PMG_SUCCESS;
}
