/* Exact coarse-grid sampler -- host side (C11).
 *
 * Mirrors PCCHOLSAMPLER's dense path (reference src/pc_chols.c): set-up = MatConvert to dense + LAPACK potrf('L')
 * (:174-194), sample = y = L^-T (L^-1 b + xi) (:220-260, :284-288), failure = PETSC_ERR_MAT_CH_ZRPVT with the order
 * of the failing minor (:190).  The factorisation runs on the device (blocked, trailing updates on the f64
 * matrix cores); a triangular solve is a chain of N dependent steps, hopeless on a GPU, so set-up also forms
 * W = L^-1 once and a sample is two triangular matrix-vector products (kernels_dense.hip).  Meant for the coarsest
 * grid of the V-cycle (N up to a few tens of thousands); the reference's sparse-direct branch for large N
 * (Pardiso, :195-209) is out of scope.
 */
#include "pmg_internal.h"

struct pmg_chol_s {
  int32_t n, npad;
  double *L_dev;  /* device, column-major npad x npad, lower factor (padding = identity) */
  double *W_lo;   /* device, row-major n x n, lower triangle = L^-1                     */
  double *W_up;   /* device, row-major n x n, upper triangle = (L^-1)^T                 */
  double *v, *xi; /* device work vectors                                                */
};

pmg_status pmg_chol_create_csr(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, pmg_chol *out)
{
  return pmg_chol_create_csr_lowrank(n, rowptr, colidx, vals, 0, NULL, NULL, out);
}

pmg_status pmg_chol_create_csr_idx(int64_t n, const void *rowptr, const void *colidx, const double *vals, int idx_width, int32_t k, const double *B_host, const double *S_host, pmg_chol *out)
{
  const int32_t *rp, *ci;
  int32_t       *rpo, *cio;
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  PMG_CALL(pmg_narrow_csr(n, n, rowptr, colidx, idx_width, &rp, &ci, &rpo, &cio));
  pmg_status st = pmg_chol_create_csr_lowrank((int32_t)n, rp, ci, vals, k, B_host, S_host, out); /* copies what it needs */
  free(rpo);
  free(cio);
  return st;
}

/* PCSetUp_CholSampler on a MATLRC operator (src/pc_chols.c:119-153): the factored matrix is P = A + B S B^T, formed
   explicitly (MatMatTransposeMult + MatAXPY there; a dense rank-k accumulation here, the matrix is dense anyway). */
pmg_status pmg_chol_create_csr_lowrank(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int32_t k, const double *B_host, const double *S_host, pmg_chol *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(n >= 1, PMG_ERR_ARG_OUTOFRANGE, "n = %d", n);
  PMG_CHECK(rowptr && colidx && vals, PMG_ERR_ARG_NULL, "null CSR array");
  PMG_CHECK(k >= 0 && (k == 0 || (B_host && S_host)), PMG_ERR_ARG_NULL, "low-rank factors missing (k = %d)", k);
  const int32_t npad = (n + 31) / 32 * 32;
  PMG_CHECK((double)npad * npad * 8 * 4 < 96e9, PMG_ERR_SUP, "dense coarse sampler limited to a few tens of thousands of rows (n = %d); coarsen further", n);
  const size_t nn = (size_t)npad * npad;
  double      *A  = (double *)calloc(nn, sizeof(double));
  PMG_CHECK(A, PMG_ERR_MEM, "out of host memory for the %d x %d coarse matrix", n, n);
  for (int32_t r = 0; r < n; ++r)
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) A[r + (size_t)npad * colidx[k]] = vals[k]; /* MatConvert(S, MATSEQDENSE), :184 */
  for (int32_t c = 0; c < k; ++c) { /* + B S B^T, src/pc_chols.c:146-149 */
    const double *bc = B_host + (size_t)n * c;
    for (int32_t j = 0; j < n; ++j) {
      const double f = S_host[c] * bc[j];
      if (f == 0.0) continue;
      double *col = A + (size_t)npad * j;
      for (int32_t r = 0; r < n; ++r) col[r] += bc[r] * f;
    }
  }
  for (int32_t r = n; r < npad; ++r) A[r + (size_t)npad * r] = 1.0;
  pmg_chol ch = (pmg_chol)calloc(1, sizeof *ch);
  if (!ch) {
    free(A);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  ch->n    = n;
  ch->npad = npad;
  double    *W = NULL, *T = NULL, *Dinv = NULL;
  int       *info_dev = NULL, info = 0;
  pmg_status st = pmg_dev_upload((void **)&ch->L_dev, A, nn * sizeof(double));
  free(A);
  if (!st) st = pmg_dev_alloc((void **)&W, nn * sizeof(double));
  if (!st) st = pmg_dev_alloc((void **)&T, nn * sizeof(double)); /* scratch of the blocked inverse */
  if (!st) st = pmg_dev_alloc((void **)&Dinv, sizeof(double) * 32 * 32 * (size_t)(npad / 32));
  if (!st) st = pmg_dev_alloc((void **)&info_dev, sizeof(int));
  if (!st) st = pmg_dev_alloc((void **)&ch->W_lo, sizeof(double) * (size_t)n * n);
  if (!st) st = pmg_dev_alloc((void **)&ch->W_up, sizeof(double) * (size_t)n * n);
  if (!st) st = pmg_dev_alloc((void **)&ch->v, sizeof(double) * (size_t)n);
  if (!st) st = pmg_dev_alloc((void **)&ch->xi, sizeof(double) * (size_t)n);
  if (!st && pmgk_potrf_inverse(npad, ch->L_dev, W, T, Dinv, info_dev, NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "Cholesky kernels failed to launch"); /* LAPACKpotrf_("L"), :188 */
  if (!st && pmgk_pack_rowmajor(n, W, npad, 0, ch->W_lo, NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "pack kernel failed to launch");
  if (!st && pmgk_pack_rowmajor(n, W, npad, 1, ch->W_up, NULL)) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "pack kernel failed to launch");
  if (!st && hipMemcpy(&info, info_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "device Cholesky failed");
  pmg_dev_free(W);
  pmg_dev_free(T);
  pmg_dev_free(Dinv);
  pmg_dev_free(info_dev);
  if (!st && info) st = pmg_set_error(PMG_ERR_MAT_CH_ZRPVT, __FILE__, __LINE__, "Dense Cholesky failed: leading minor of order %d is not positive definite", info); /* :190 */
  if (st) {
    pmg_chol_destroy(&ch);
    return st;
  }
  *out = ch;
  return PMG_SUCCESS;
}

pmg_status pmg_chol_get_factor(pmg_chol ch, double *L_colmajor_host)
{
  PMG_CHECK(ch && L_colmajor_host, PMG_ERR_ARG_NULL, "null argument");
  double *tmp = (double *)malloc(sizeof(double) * (size_t)ch->npad * ch->npad);
  PMG_CHECK(tmp, PMG_ERR_MEM, "out of host memory");
  if (hipMemcpy(tmp, ch->L_dev, sizeof(double) * (size_t)ch->npad * ch->npad, hipMemcpyDeviceToHost) != hipSuccess) {
    free(tmp);
    PMG_FAIL(PMG_ERR_GPU, "download of the factor failed");
  }
  for (int j = 0; j < ch->n; ++j)
    for (int i = 0; i < ch->n; ++i) L_colmajor_host[i + (size_t)ch->n * j] = i >= j ? tmp[i + (size_t)ch->npad * j] : 0.0;
  free(tmp);
  return PMG_SUCCESS;
}

/* y = L^-T (L^-1 b + xi); noisy == 0 gives the deterministic solve y = A^-1 b */
pmg_status pmg_chol_sample(pmg_chol ch, const double *b_dev, double *y_dev, int noisy, uint64_t seed, uint64_t counter, void *stream)
{
  PMG_CHECK(ch && b_dev && y_dev, PMG_ERR_ARG_NULL, "null argument");
  if (noisy) PMG_KERNEL(pmgk_fill_normal_rows(ch->n, seed, counter, ch->xi, stream)); /* VecSetRandomStandardNormal(chol->r), :285 */
  PMG_KERNEL(pmgk_tri_gemv(ch->n, 0, ch->W_lo, b_dev, noisy ? ch->xi : NULL, ch->v, stream)); /* v = L^-1 b (+ xi), :284-286 */
  PMG_KERNEL(pmgk_tri_gemv(ch->n, 1, ch->W_up, ch->v, NULL, y_dev, stream));                  /* y = L^-T v, :287      */
  return PMG_SUCCESS;
}

pmg_status pmg_chol_destroy(pmg_chol *ch)
{
  if (!ch || !*ch) return PMG_SUCCESS;
  pmg_dev_free((*ch)->W_lo);
  pmg_dev_free((*ch)->W_up);
  pmg_dev_free((*ch)->v);
  pmg_dev_free((*ch)->xi);
  pmg_dev_free((*ch)->L_dev);
  free(*ch);
  *ch = NULL;
  return PMG_SUCCESS;
}
