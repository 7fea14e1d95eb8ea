/* Exact coarse-grid sampler -- host side (C11).
 *
 * Mirrors PCCHOLSAMPLER's dense path (reference src/pc_chols.c): set-up = MatConvert to dense + LAPACK potrf('L')
 * (:174-194), sample = y = L^-T (L^-1 b + xi) (:220-260, :284-288), failure = PETSC_ERR_MAT_CH_ZRPVT with the order
 * of the failing minor (:190).  A triangular solve is a chain of N dependent steps, hopeless on a GPU, so set-up
 * also forms W = L^-1 once (forward substitution on the identity) and a sample is two triangular
 * matrix-vector products on the device (kernels_dense.hip).  Meant for the coarsest grid of the V-cycle
 * (N up to a few thousand); the reference's sparse-direct branch for large N (Pardiso, :195-209) is out of scope.
 */
#include "pmg_internal.h"
#include <math.h>

struct pmg_chol_s {
  int32_t n;
  double *W_lo;  /* device, row-major n x n, lower triangle = L^-1         */
  double *W_up;  /* device, row-major n x n, upper triangle = (L^-1)^T     */
  double *v, *xi; /* device work vectors                                   */
  double *L_host; /* host, column-major lower factor (kept for inspection) */
};

/* unblocked lower Cholesky, column-major, in place; returns 0 or the 1-based order of the failing minor */
static int potrf_lower(int n, double *a)
{
  for (int j = 0; j < n; ++j) {
    double d = a[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) d -= a[j + (size_t)n * k] * a[j + (size_t)n * k];
    if (!(d > 0)) return j + 1;
    d                    = sqrt(d);
    a[j + (size_t)n * j] = d;
    /* column update: a[i,j] = (a[i,j] - sum_k a[i,k] a[j,k]) / d, accumulated column by column for stride-1 access */
    for (int k = 0; k < j; ++k) {
      const double ajk = a[j + (size_t)n * k];
      if (ajk != 0.0)
        for (int i = j + 1; i < n; ++i) a[i + (size_t)n * j] -= a[i + (size_t)n * k] * ajk;
    }
    for (int i = j + 1; i < n; ++i) a[i + (size_t)n * j] /= d;
  }
  return 0;
}

pmg_status pmg_chol_create_csr(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, pmg_chol *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(n >= 1, PMG_ERR_ARG_OUTOFRANGE, "n = %d", n);
  PMG_CHECK(rowptr && colidx && vals, PMG_ERR_ARG_NULL, "null CSR array");
  PMG_CHECK((double)n * n * 8 * 4 < 64e9, PMG_ERR_SUP, "dense coarse sampler limited to a few tens of thousands of rows (n = %d); coarsen further", n);
  const size_t nn = (size_t)n * n;
  double      *A  = (double *)calloc(nn, sizeof(double));
  double      *W  = (double *)calloc(nn, sizeof(double));
  double      *T  = (double *)calloc(nn, sizeof(double));
  if (!A || !W || !T) {
    free(A); free(W); free(T);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory for the %d x %d coarse factor", n, n);
  }
  for (int32_t r = 0; r < n; ++r)
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) A[r + (size_t)n * colidx[k]] = vals[k]; /* MatConvert(S, MATSEQDENSE), :184 */
  const int info = potrf_lower(n, A);
  if (info) {
    free(A); free(W); free(T);
    PMG_FAIL(PMG_ERR_MAT_CH_ZRPVT, "Dense Cholesky failed: leading minor of order %d is not positive definite", info); /* :190 */
  }
  /* W = L^-1 by forward substitution on the identity, column by column (W is lower triangular) */
  for (int c = 0; c < n; ++c) {
    double *w = W + (size_t)n * c; /* column c of W, column-major */
    w[c]      = 1.0 / A[c + (size_t)n * c];
    for (int i = c + 1; i < n; ++i) {
      double s = 0.0;
      for (int k = c; k < i; ++k) s -= A[i + (size_t)n * k] * w[k];
      w[i] = s / A[i + (size_t)n * i];
    }
  }
  pmg_chol ch = (pmg_chol)calloc(1, sizeof *ch);
  if (!ch) {
    free(A); free(W); free(T);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  ch->n      = n;
  ch->L_host = A;
  /* row-major lower copy: W_lo[i*n + k] = W(i,k) = W_colmajor[i + n*k] */
  for (int i = 0; i < n; ++i)
    for (int k = 0; k <= i; ++k) T[(size_t)i * n + k] = W[i + (size_t)n * k];
  pmg_status st = pmg_dev_upload((void **)&ch->W_lo, T, nn * sizeof(double));
  /* row-major upper copy of W^T: W_up[i*n + k] = W(k,i), k >= i; column-major W is exactly that array */
  if (!st) st = pmg_dev_upload((void **)&ch->W_up, W, nn * sizeof(double));
  if (!st) st = pmg_dev_alloc((void **)&ch->v, sizeof(double) * (size_t)n);
  if (!st) st = pmg_dev_alloc((void **)&ch->xi, sizeof(double) * (size_t)n);
  free(W);
  free(T);
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
  for (int j = 0; j < ch->n; ++j)
    for (int i = 0; i < ch->n; ++i) L_colmajor_host[i + (size_t)ch->n * j] = i >= j ? ch->L_host[i + (size_t)ch->n * j] : 0.0;
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
  free((*ch)->L_host);
  free(*ch);
  *ch = NULL;
  return PMG_SUCCESS;
}
