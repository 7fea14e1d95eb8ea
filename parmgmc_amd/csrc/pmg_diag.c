/* Chain diagnostics -- host side (C11), no device work.
 *
 * Mirrors the two diagnostics the reference's examples use to judge a sampler (they run on the host there too):
 *   - Autocorrelation / IACT (src/iact.c:17-92; caller examples/ex2.c:107): autocorrelation function of a scalar
 *     quantity of interest through a zero-padded FFT of length 2 * nextpow2(n), integrated autocorrelation time with
 *     the automatic window "first M with M >= 5 tau(M)";
 *   - EstimateCovarianceMatErrors (src/stats.c:94-117; caller examples/ex6.c:193): for every sample index, the
 *     unbiased sample covariance over independent chains against A^-1, relative Frobenius error.
 * FFTW is replaced by a plain radix-2 transform (the lengths are powers of two by construction).
 */
#include "pmg_internal.h"
#include <math.h>

static int64_t next_pow_two(int64_t n) /* src/iact.c:10-15 */
{
  int64_t i = 1;
  while (i < n) i <<= 1;
  return i;
}

/* in-place iterative radix-2 FFT on interleaved (re, im); sign = -1 forward, +1 backward (unnormalised, as FFTW) */
static void fft_radix2(int64_t n, double *a, int sign)
{
  for (int64_t i = 1, j = 0; i < n; ++i) {
    int64_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = a[2 * i]; a[2 * i] = a[2 * j]; a[2 * j] = t;
      t = a[2 * i + 1]; a[2 * i + 1] = a[2 * j + 1]; a[2 * j + 1] = t;
    }
  }
  const double pi = 3.14159265358979323846;
  for (int64_t len = 2; len <= n; len <<= 1) {
    const int64_t half = len >> 1;
    for (int64_t k = 0; k < half; ++k) {
      const double ang = sign * 2.0 * pi * (double)k / (double)len;
      const double wr = cos(ang), wi = sin(ang);
      for (int64_t s = k; s < n; s += len) {
        const int64_t u = s, v = s + half;
        const double  xr = a[2 * v] * wr - a[2 * v + 1] * wi, xi = a[2 * v] * wi + a[2 * v + 1] * wr;
        a[2 * v]     = a[2 * u] - xr;
        a[2 * v + 1] = a[2 * u + 1] - xi;
        a[2 * u] += xr;
        a[2 * u + 1] += xi;
      }
    }
  }
}

/* Autocorrelation (src/iact.c:17-47): acf[i] = c(i) / c(0), c = inverse FFT of |FFT(x - mean, zero-padded to 2N)|^2 */
pmg_status pmg_autocorrelation(int64_t n, const double *x, double *acf)
{
  PMG_CHECK(x && acf, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(n >= 1, PMG_ERR_ARG_OUTOFRANGE, "n = %lld", (long long)n);
  const int64_t N = next_pow_two(n), M = 2 * N;
  double       *in = (double *)calloc((size_t)(2 * M), sizeof(double));
  PMG_CHECK(in, PMG_ERR_MEM, "out of host memory");
  double mean = 0;
  for (int64_t i = 0; i < n; ++i) mean += 1. / (double)n * x[i]; /* :29 */
  for (int64_t i = 0; i < n; ++i) in[2 * i] = x[i] - mean;
  fft_radix2(M, in, -1);
  for (int64_t i = 0; i < M; ++i) { /* out * conj(out), :36 */
    in[2 * i]     = in[2 * i] * in[2 * i] + in[2 * i + 1] * in[2 * i + 1];
    in[2 * i + 1] = 0.0;
  }
  fft_radix2(M, in, +1);
  for (int64_t i = 0; i < n; ++i) acf[i] = in[2 * i] / in[0]; /* :42 */
  free(in);
  return PMG_SUCCESS;
}

/* IACT (src/iact.c:73-92) with AutoWindow(c = 5) (:49-71): tau(M) = 2 sum_{i<=M} acf[i] - 1, window = first M with
   M >= 5 tau(M) (n - 1 when M < 5 tau(M) never holds; 0 when no M qualifies); valid = 500 tau <= n */
pmg_status pmg_iact(int64_t n, const double *x, double *tau, double *acf_out, int *valid)
{
  PMG_CHECK(x && tau, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(n > 1, PMG_ERR_ARG_OUTOFRANGE, "Too few data points"); /* :79 */
  double *out = (double *)malloc(sizeof(double) * (size_t)n);
  PMG_CHECK(out, PMG_ERR_MEM, "out of host memory");
  pmg_status st = pmg_autocorrelation(n, x, out);
  if (st) {
    free(out);
    return st;
  }
  if (acf_out) memcpy(acf_out, out, sizeof(double) * (size_t)n);
  for (int64_t i = 1; i < n; ++i) out[i] = out[i] + out[i - 1];
  for (int64_t i = 0; i < n; ++i) out[i] = 2 * out[i] - 1;
  int64_t w    = n - 1;
  int     flag = 0;
  for (int64_t i = 0; i < n; ++i)
    if ((double)i < 5 * out[i]) {
      flag = 1;
      break;
    }
  if (flag) {
    w = 0;
    for (int64_t i = 0; i < n; ++i)
      if ((double)i >= 5 * out[i]) {
        w = i;
        break;
      }
  }
  *tau = out[w];
  if (valid) *valid = 500 * (*tau) <= (double)n;
  free(out);
  return PMG_SUCCESS;
}

/* dense SPD inverse through Cholesky (the reference uses a dense LU + MatMatSolve on the identity, src/stats.c:8-31) */
static int dense_spd_inverse(int n, double *a /* column-major, overwritten by the inverse */)
{
  double *L = (double *)malloc(sizeof(double) * (size_t)n * n), *W = (double *)calloc((size_t)n * n, sizeof(double));
  if (!L || !W) {
    free(L); free(W);
    return -1;
  }
  memcpy(L, a, sizeof(double) * (size_t)n * n);
  for (int j = 0; j < n; ++j) {
    double d = L[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) d -= L[j + (size_t)n * k] * L[j + (size_t)n * k];
    if (!(d > 0)) {
      free(L); free(W);
      return j + 1;
    }
    d = sqrt(d);
    L[j + (size_t)n * j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = L[i + (size_t)n * j];
      for (int k = 0; k < j; ++k) s -= L[i + (size_t)n * k] * L[j + (size_t)n * k];
      L[i + (size_t)n * j] = s / d;
    }
  }
  for (int c = 0; c < n; ++c) { /* W = L^-1, column by column */
    for (int i = c; i < n; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= L[i + (size_t)n * k] * W[k + (size_t)n * c];
      W[i + (size_t)n * c] = s / L[i + (size_t)n * i];
    }
  }
  for (int j = 0; j < n; ++j) /* A^-1 = W^T W */
    for (int i = 0; i <= j; ++i) {
      double s = 0;
      for (int k = j; k < n; ++k) s += W[k + (size_t)n * i] * W[k + (size_t)n * j];
      a[i + (size_t)n * j] = a[j + (size_t)n * i] = s;
    }
  free(L);
  free(W);
  return 0;
}

/* EstimateCovarianceMatErrors (src/stats.c:94-117): samples ordered {sample 0 of chain 0, sample 0 of chain 1, ...,
   sample 1 of chain 0, ...} as contiguous host rows of length n; errs[i] = ||C_i - A^-1||_F / ||A^-1||_F with
   C_i the unbiased (1 / (chains - 1), :80) sample covariance over the chains at sample index i */
pmg_status pmg_estimate_covariance_errors(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int32_t chains, int32_t samples_per_chain, const double *samples, double *errs)
{
  PMG_CHECK(rowptr && colidx && vals && samples && errs, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(n >= 1 && chains >= 2 && samples_per_chain >= 1, PMG_ERR_ARG_OUTOFRANGE, "n = %d, chains = %d, samples = %d", n, chains, samples_per_chain);
  PMG_CHECK(n <= 4096, PMG_ERR_SUP, "dense covariance diagnostics are meant for small problems (n = %d)", n);
  const size_t nn = (size_t)n * n;
  double      *Q = (double *)calloc(nn, sizeof(double)), *Cm = (double *)malloc(sizeof(double) * nn), *m = (double *)malloc(sizeof(double) * (size_t)n), *w = (double *)malloc(sizeof(double) * (size_t)n);
  pmg_status   st = PMG_SUCCESS;
  if (!Q || !Cm || !m || !w) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    for (int32_t r = 0; r < n; ++r)
      for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) Q[r + (size_t)n * colidx[q]] = vals[q];
    const int info = dense_spd_inverse(n, Q);
    if (info < 0) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    else if (info > 0) st = pmg_set_error(PMG_ERR_MAT_CH_ZRPVT, __FILE__, __LINE__, "matrix is not positive definite (leading minor %d)", info);
  }
  if (!st) {
    double qn = 0;
    for (size_t i = 0; i < nn; ++i) qn += Q[i] * Q[i];
    qn = sqrt(qn);
    for (int32_t i = 0; i < samples_per_chain; ++i) {
      const double *S = samples + (size_t)i * chains * n;
      for (int32_t r = 0; r < n; ++r) m[r] = 0; /* SampleMean, :55-61 */
      for (int32_t c = 0; c < chains; ++c)
        for (int32_t r = 0; r < n; ++r) m[r] += 1. / chains * S[(size_t)c * n + r];
      memset(Cm, 0, sizeof(double) * nn);
      for (int32_t c = 0; c < chains; ++c) { /* SampleCovariance, :63-84 */
        for (int32_t r = 0; r < n; ++r) w[r] = S[(size_t)c * n + r] - m[r];
        for (int32_t j = 0; j < n; ++j) {
          const double f = 1. / (chains - 1) * w[j];
          for (int32_t r = 0; r < n; ++r) Cm[r + (size_t)n * j] += f * w[r];
        }
      }
      double e = 0;
      for (size_t q = 0; q < nn; ++q) e += (Cm[q] - Q[q]) * (Cm[q] - Q[q]);
      errs[i] = sqrt(e) / qn; /* :111-113 */
    }
  }
  free(Q);
  free(Cm);
  free(m);
  free(w);
  return st;
}

/* MakeObservationMats (src/obs.c:135-180) on the unit-cube DMDA (the reference builds it on a DMPlex; its DMDA variant
   is the sketch at :69-95): observation i is the average of the field over the ball of radius radii[i] around
   coords[dim*i ..] -- column i of B = M u_i with u_i = 1/vol inside the ball (:39-50), vol = pi r^2 or 4/3 pi r^3
   (:27-37), M the LUMPED mass of the uniform grid, h^dim, in place of the FE mass matrix (FE assembly is out of scope) --,
   S = 1/sigma2 (:150), f = B (S o obsvals) (:160-178).  Rows are this rank's planes [kz0, kz0+nz_owned) in natural
   order, so a z-slab run builds only its own rows.  Host arrays: B is (nx*ny*nz_owned) x nobs column-major. */
pmg_status pmg_make_observation_mats_dmda(int32_t nx, int32_t ny, int32_t nzg, int32_t kz0, int32_t nz_owned, int32_t nobs, double sigma2, const double *coords, const double *radii, const double *obsvals, double *B_host, double *S_host, double *f_host)
{
  PMG_CHECK(coords && radii && B_host && S_host, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(nx >= 2 && ny >= 2 && nzg >= 1 && kz0 >= 0 && nz_owned >= 1 && kz0 + nz_owned <= nzg, PMG_ERR_ARG_OUTOFRANGE, "grid %d x %d x %d, planes [%d, %d)", nx, ny, nzg, kz0, kz0 + nz_owned);
  PMG_CHECK(nobs >= 1 && nobs <= 64 && sigma2 > 0, PMG_ERR_ARG_OUTOFRANGE, "nobs = %d, sigma2 = %g", nobs, sigma2);
  PMG_CHECK(!f_host || obsvals, PMG_ERR_ARG_NULL, "f needs the observed values");
  const int     dim = nzg > 1 ? 3 : 2; /* DMGetCoordinateDim, :27-31 */
  const double  pi = 3.14159265358979323846;
  const int64_t n  = (int64_t)nx * ny * nz_owned;
  const double  hx = 1.0 / (nx - 1), hy = 1.0 / (ny - 1), hz = dim == 3 ? 1.0 / (nzg - 1) : 1.0;
  const double  mass = hx * hy * (dim == 3 ? hz : 1.0);
  memset(B_host, 0, sizeof(double) * (size_t)n * (size_t)nobs);
  for (int32_t o = 0; o < nobs; ++o) {
    const double *p = coords + (size_t)dim * o, r = radii[o];
    const double  vol = dim == 2 ? pi * r * r : 4 * pi / 3. * r * r * r;
    double       *col = B_host + (size_t)n * o;
    for (int32_t k = 0; k < nz_owned; ++k)
      for (int32_t j = 0; j < ny; ++j)
        for (int32_t i = 0; i < nx; ++i) {
          double diff = (i * hx - p[0]) * (i * hx - p[0]) + (j * hy - p[1]) * (j * hy - p[1]);
          if (dim == 3) diff += ((kz0 + k) * hz - p[2]) * ((kz0 + k) * hz - p[2]);
          if (diff < r * r) col[i + (int64_t)nx * (j + (int64_t)ny * k)] = mass / vol; /* :46-47 */
        }
    S_host[o] = 1. / sigma2;
  }
  if (f_host) { /* f = B (S o y), :171-173 */
    memset(f_host, 0, sizeof(double) * (size_t)n);
    for (int32_t o = 0; o < nobs; ++o) {
      const double  w = S_host[o] * obsvals[o];
      const double *col = B_host + (size_t)n * o;
      for (int64_t q = 0; q < n; ++q) f_host[q] += col[q] * w;
    }
  }
  return PMG_SUCCESS;
}
