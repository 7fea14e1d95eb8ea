/* AddressSanitizer + UndefinedBehaviorSanitizer run of the host-side C that needs no GPU: the CPU oracle (every orc_*
 * entry point on small cases, incl. the sampled-row functions the full-size parity tests rely on) and the pure-host parts
 * of libparmgmc_hip (index narrowing, chain diagnostics, observation matrices, the PCPARSOR data-flow builder).  Built
 * and run by tests/test_sanitize_host.py with gcc -fsanitize=address,undefined; a finding aborts with a non-zero status.
 * (GPU AddressSanitizer is not available on the test pool: the kernels are covered by the parity tests only.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "parmgmc_hip.h"

/* oracle entry points (oracle/pmg_oracle.c is compiled into this program) */
int64_t orc_laplace_nnz(int, int, int);
void    orc_assemble_shifted_laplace(int, int, int, double, int32_t *, int32_t *, double *);
void    orc_assemble_ex6(int, double, int32_t *, int32_t *, double *);
void    orc_diag_pointers(int, const int32_t *, const int32_t *, int32_t *);
void    orc_idiag(int, const int32_t *, const double *, double, double *);
void    orc_sqrtdiag(int, const int32_t *, const double *, double, int, double *);
void    orc_mcsor_apply(int, int, const int32_t *, const int32_t *, const int32_t *, const int32_t *, const double *, const int32_t *, const double *, double, const double *, double *);
void    orc_noise_rows(int64_t, uint64_t, uint64_t, double *);
void    orc_noise_grid(int, int, int, uint64_t, uint64_t, double *);
void    orc_gibbs_sample_serial(int, const int32_t *, const int32_t *, const double *, const int32_t *, const double *, const double *, double, const double *, double *, double *, uint64_t, uint64_t);
void    orc_grid7_rows_sweep(int, int, int, double, double, int, int, int, uint64_t, uint64_t, int64_t, const int64_t *, const double *, const double *, const double *, double *);
void    orc_grid7_rows_residual(int, int, int, double, int64_t, const int64_t *, const double *, const double *, double *);
void    orc_st27_rows(int, int, int, int, const double *, const double *, double, int, int, uint64_t, uint64_t, int64_t, const int64_t *, const double *, const double *, const double *, double *);
void    orc_q1_rows(int, const int32_t *, const int32_t *, int64_t, const int64_t *, const double *, const double *, double *);
int     orc_potrf_lower(int, double *);
void    orc_chol_sample(int, const double *, const double *, const double *, double *);
void    orc_spmv(int, const int32_t *, const int32_t *, const double *, const double *, double *);

/* internal (not in the public header) */
pmg_status pmg_narrow_csr(int64_t nrows, int64_t ncols, const void *rowptr, const void *colidx, int idx_width, const int32_t **rp, const int32_t **ci, int32_t **rp_own, int32_t **ci_own);
pmg_status pmg_parsor_build_dataflow(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int32_t nparts, const int32_t *row_starts, const int32_t *proccols_in, int32_t **e_rowptr, int32_t **e_colidx, double **e_vals, int32_t **e_colors, int32_t *nlevels_out, int32_t *proccols_out, int32_t *classes_out);

/* the one kernel launcher pmg_common.c references (the kernels themselves are not part of this host-only program) */
int pmgk_fill_normal_rows(int64_t n, uint64_t seed, uint64_t sweep, double *xi, void *stream)
{
  (void)n; (void)seed; (void)sweep; (void)xi; (void)stream;
  return 1;
}

#define REQUIRE(c) \
  do { \
    if (!(c)) { \
      fprintf(stderr, "host_san: check failed at line %d: %s\n", __LINE__, #c); \
      return 1; \
    } \
  } while (0)

int main(void)
{
  const int nx = 7, ny = 5, nz = 4, n = nx * ny * nz;
  const int64_t nnz = orc_laplace_nnz(nx, ny, nz);
  int32_t *rp = malloc(sizeof(int32_t) * (n + 1)), *ci = malloc(sizeof(int32_t) * nnz), *dp = malloc(sizeof(int32_t) * n);
  double  *v = malloc(sizeof(double) * nnz), *idg = malloc(sizeof(double) * n), *sd = malloc(sizeof(double) * n);
  double  *b = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n), *y0 = malloc(sizeof(double) * n), *w = malloc(sizeof(double) * n), *xi = malloc(sizeof(double) * n);
  orc_assemble_shifted_laplace(nx, ny, nz, 2.0, rp, ci, v);
  REQUIRE(rp[n] == nnz);
  orc_diag_pointers(n, rp, ci, dp);
  orc_idiag(n, dp, v, 1.2, idg);
  orc_sqrtdiag(n, dp, v, 1.2, 1, sd);
  for (int r = 0; r < n; ++r) {
    b[r]  = sin(1.0 + r);
    y0[r] = y[r] = cos(0.5 * r);
  }
  /* red-black colour lists */
  int32_t cptr[3] = {0, 0, 0}, *crow = malloc(sizeof(int32_t) * n);
  int     p = 0;
  for (int c = 0; c < 2; ++c) {
    cptr[c] = p;
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i)
          if (((i + j + k) & 1) == c) crow[p++] = i + nx * (j + ny * k);
  }
  cptr[2] = p;
  orc_mcsor_apply(1, 2, cptr, crow, rp, ci, v, dp, idg, 1.2, b, y);
  /* the sampled-row restatement gives the same values for every row */
  int64_t *rows = malloc(sizeof(int64_t) * n);
  double  *out = malloc(sizeof(double) * n);
  for (int r = 0; r < n; ++r) rows[r] = r;
  orc_grid7_rows_sweep(nx, ny, nz, 2.0, 1.2, 0, 0, 1, 0, 0, n, rows, b, y0, y, out);
  for (int r = 0; r < n; ++r) REQUIRE(out[r] == y[r]);
  orc_grid7_rows_residual(nx, ny, nz, 2.0, n, rows, b, y, out);
  orc_spmv(n, rp, ci, v, y, w);
  for (int r = 0; r < n; ++r) REQUIRE(out[r] == b[r] - w[r]);
  orc_noise_rows(n, 7, 3, xi);
  orc_noise_grid(nx, ny, nz, 7, 3, xi);
  orc_gibbs_sample_serial(n, rp, ci, v, dp, idg, sd, 1.2, b, y, w, 9, 1);
  for (int r = 0; r < n; ++r) REQUIRE(isfinite(y[r]));
  /* class-stencil rows with a synthetic table, Q1 rows */
  double coef[27 * 27], sq[27];
  for (int q = 0; q < 27 * 27; ++q) coef[q] = (q % 27 == 13) ? 30.0 : -1.0 / (1 + q % 5);
  for (int q = 0; q < 27; ++q) sq[q] = 1.0;
  orc_st27_rows(0, nx, ny, nz, coef, sq, 1.0, 0, 1, 5, 2, n, rows, b, y0, y, out);
  orc_st27_rows(1, nx, ny, nz, coef, sq, 1.0, 0, 0, 0, 0, n, rows, b, y0, y0, out);
  const int32_t nf[3] = {9, 5, 3}, ncs[3] = {5, 3, 2};
  double        fine[9 * 5 * 3], coarse[5 * 3 * 2], o2[9 * 5 * 3];
  int64_t       r2[9 * 5 * 3];
  for (int q = 0; q < 9 * 5 * 3; ++q) { fine[q] = q; r2[q] = q; }
  for (int q = 0; q < 5 * 3 * 2; ++q) coarse[q] = 1.0 + q;
  orc_q1_rows(0, nf, ncs, 5 * 3 * 2, r2, fine, fine, o2);
  orc_q1_rows(1, nf, ncs, 9 * 5 * 3, r2, fine, coarse, o2);
  /* dense coarse sample */
  double A3[9] = {4, 1, 0, 1, 3, 1, 0, 1, 2}, x3[3] = {1, 2, 3}, z3[3] = {0.1, -0.2, 0.3}, y3[3];
  REQUIRE(orc_potrf_lower(3, A3) == 0);
  orc_chol_sample(3, A3, x3, z3, y3);
  /* ex6 matrix */
  int32_t erp[17], eci[16 * 5];
  double  ev[16 * 5];
  orc_assemble_ex6(4, 1e-4, erp, eci, ev);

  /* ---- libparmgmc_hip, host-only pieces ---- */
  int64_t rp64[4] = {0, 2, 4, 6}, ci64[6] = {0, 1, 0, 1, 1, 2};
  const int32_t *nrp, *nci;
  int32_t       *orp, *oci;
  REQUIRE(pmg_narrow_csr(3, 3, rp64, ci64, 64, &nrp, &nci, &orp, &oci) == 0 && orp && oci && nrp[3] == 6 && nci[5] == 2);
  free(orp);
  free(oci);
  ci64[2] = (int64_t)1 << 40;
  REQUIRE(pmg_narrow_csr(3, 3, rp64, ci64, 64, &nrp, &nci, &orp, &oci) == PMG_ERR_ARG_OUTOFRANGE && !orp && !oci);
  REQUIRE(pmg_narrow_csr(3, 3, rp64, ci64, 48, &nrp, &nci, &orp, &oci) == PMG_ERR_ARG_OUTOFRANGE);
  double series[600], acf[600], tau;
  int    valid;
  double prev = 0;
  for (int q = 0; q < 600; ++q) series[q] = prev = 0.8 * prev + sin(12.9898 * q) * 43758.5453 - floor(sin(12.9898 * q) * 43758.5453) - 0.5;
  REQUIRE(pmg_autocorrelation(600, series, acf) == 0 && fabs(acf[0] - 1.0) < 1e-12);
  REQUIRE(pmg_iact(600, series, &tau, acf, &valid) == 0 && tau > 1.0);
  double Bm[7 * 5 * 4 * 2], Sm[2], fm[7 * 5 * 4];
  const double centres[6] = {0.3, 0.3, 0.3, 0.7, 0.6, 0.5}, radii[2] = {0.3, 0.25}, obs[2] = {1.0, -1.0};
  REQUIRE(pmg_make_observation_mats_dmda(nx, ny, nz, 0, nz, 2, 1e-2, centres, radii, obs, Bm, Sm, fm) == 0);
  double samples[2 * 3 * 16];
  for (int q = 0; q < 2 * 3 * 16; ++q) samples[q] = sin(0.37 * q);
  double errs[3];
  REQUIRE(pmg_estimate_covariance_errors(16, erp, eci, ev, 2, 3, samples, errs) == 0);
  int32_t starts[4] = {0, 50, 100, n}, *erpp, *ecii, *ecol, nlev, pcol[3], *cls = malloc(sizeof(int32_t) * n);
  double *evv;
  REQUIRE(pmg_parsor_build_dataflow(n, rp, ci, v, 3, starts, NULL, &erpp, &ecii, &evv, &ecol, &nlev, pcol, cls) == 0 && nlev >= 1);
  free(erpp);
  free(ecii);
  free(evv);
  free(ecol);
  free(cls);
  free(rp); free(ci); free(dp); free(v); free(idg); free(sd); free(b); free(y); free(y0); free(w); free(xi); free(crow); free(rows); free(out);
  printf("host_san ok\n");
  return 0;
}
