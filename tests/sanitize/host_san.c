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

/* the two kernel launchers pmg_common.c references (the kernels themselves are not part of this host-only program) */
int pmgk_fill_normal_rows(int64_t n, uint64_t seed, uint64_t sweep, double *xi, void *stream)
{
  (void)n; (void)seed; (void)sweep; (void)xi; (void)stream;
  return 1;
}
int pmgk_stream_triad(int64_t n, const double *a, const double *b, double *c, void *stream)
{
  (void)n; (void)a; (void)b; (void)c; (void)stream;
  return 1;
}

/* ---- pmg_rowblock.c (the multi-rank set-up in C) is host code apart from its last two entry points: it runs here on THREE
   ranks = three threads whose byte all-gather is a shared buffer between two barriers.  The device-side functions it can
   call (sampler / transport creation) are not part of this program: stubs that fail if reached. ------------------------ */
#include <pthread.h>
#define STUB(name, ...) pmg_status name(__VA_ARGS__) { return PMG_ERR_SUP; }
typedef struct pmg_mcsor_s     *san_mc;
typedef struct pmg_distmcsor_s *san_dm;
STUB(pmg_mcsor_create_csr, int32_t n, const int32_t *a, const int32_t *b, const double *c, pmg_mcsor *o) 
STUB(pmg_mcsor_set_coloring, pmg_mcsor m, int r, const int32_t *c)
STUB(pmg_mcsor_set_omega, pmg_mcsor m, double o)
STUB(pmg_mcsor_setup, pmg_mcsor m)
STUB(pmg_mcsor_set_noise_row_offset, pmg_mcsor m, int64_t r)
STUB(pmg_mcsor_get_layout, pmg_mcsor m, int32_t *p)
STUB(pmg_mcsor_destroy, pmg_mcsor *m)
void pmg_mcsor_adopt_arrays(pmg_mcsor m, int32_t *a, int32_t *b, double *c) { (void)m; (void)a; (void)b; (void)c; }
STUB(pmg_distmcsor_create, pmg_mcsor m, pmg_dist d, int32_t nc, const int64_t *a, const int32_t *b, const int64_t *c, const int64_t *e, const int32_t *f, const int32_t *g, pmg_distmcsor *o)
STUB(pmg_distmcsor_destroy, pmg_distmcsor *h)
STUB(pmg_dist_get_unique_id, const char *p, void *id)
STUB(pmg_dist_create, pmg_grid g, int32_t r, int32_t n, const void *id, const char *p, int l, pmg_dist *d)
/* the ipc transport's device layer: unsupported (as everywhere above) unless san_fail_on_rank >= 0 -- then the creation
   SUCCEEDS on every rank but that one, which is the situation the tear-down of pmg_dist_create_comm must survive: some ranks
   hold an object, one does not, and all of them have to pass the same collectives (advisor finding, round 3) */
static int  san_fail_on_rank = -1;
static char san_fake_dist[8];
static int  san_disconnects[8], san_destroys[8];
pmg_status pmg_dist_create_ipc(pmg_grid g, int32_t r, int32_t n, pmg_dist *d)
{
  (void)g; (void)n;
  if (san_fail_on_rank < 0 || r == san_fail_on_rank) return PMG_ERR_SUP;
  *d = (pmg_dist)&san_fake_dist[r];
  return PMG_SUCCESS;
}
pmg_status pmg_dist_ipc_blob_bytes(int32_t *b)
{
  if (san_fail_on_rank < 0) return PMG_ERR_SUP;
  *b = 16;
  return PMG_SUCCESS;
}
pmg_status pmg_dist_ipc_export(pmg_dist d, void *b)
{
  if (san_fail_on_rank < 0) return PMG_ERR_SUP;
  memset(b, (int)((char *)d - san_fake_dist) + 1, 16);
  return PMG_SUCCESS;
}
STUB(pmg_dist_ipc_connect, pmg_dist d, const void *a, const void *b)
STUB(pmg_dist_ipc_connect_all, pmg_dist d, const void *const *b)
pmg_status pmg_dist_ipc_disconnect(pmg_dist d)
{
  if (san_fail_on_rank < 0) return PMG_ERR_SUP;
  san_disconnects[(char *)d - san_fake_dist]++;
  return PMG_SUCCESS;
}
pmg_status pmg_dist_destroy(pmg_dist *d)
{
  if (san_fail_on_rank < 0) return PMG_ERR_SUP;
  if (d && *d) san_destroys[(char *)*d - san_fake_dist]++, *d = NULL;
  return PMG_SUCCESS;
}
STUB(pmg_mgmc_create_hierarchy, int32_t l, pmg_mgmc *m)
STUB(pmg_mgmc_set_coloring, pmg_mgmc m, int rule)
STUB(pmg_mgmc_set_level_operator, pmg_mgmc m, int32_t l, int32_t n, const int32_t *a, const int32_t *b, const double *c)
STUB(pmg_mgmc_set_level_interpolation, pmg_mgmc m, int32_t l, int32_t n, int32_t k, const int32_t *a, const int32_t *b, const double *c)
STUB(pmg_mgmc_set_rowblock_transport, pmg_mgmc m, pmg_dist d, const int64_t *c)
STUB(pmg_mgmc_set_level_rowblock, pmg_mgmc m, int32_t l, int64_t r0, int32_t no, int32_t nc, const int32_t *c, const int64_t *a, const int32_t *b, const int64_t *d, const int64_t *e, const int32_t *f, const int32_t *g)
STUB(pmg_mgmc_set_level_restriction, pmg_mgmc m, int32_t l, int32_t n, int32_t k, const int32_t *a, const int32_t *b, const double *c)
STUB(pmg_mgmc_destroy, pmg_mgmc *m)

#define SAN_RANKS 3
static pthread_barrier_t san_bar;
static char              san_buf[1 << 20];
static int san_allgather(void *ctx, const void *send, int64_t nbytes, void *recv)
{
  const int rank = *(const int *)ctx;
  if (nbytes * SAN_RANKS > (int64_t)sizeof san_buf) return 1;
  if (nbytes) memcpy(san_buf + (size_t)nbytes * (size_t)rank, send, (size_t)nbytes);
  pthread_barrier_wait(&san_bar);
  if (nbytes) memcpy(recv, san_buf, (size_t)nbytes * SAN_RANKS);
  pthread_barrier_wait(&san_bar);
  return 0;
}

/* a 3-level hierarchy of chains: level 2 = 1-D Laplacian on n2 points with a few long-range couplings (so that ghost rows
   are not only the neighbours of the block edges), level l-1 = pairs aggregated; every rank holds its rows */
static void *san_rank(void *arg)
{
  const int     rank = *(int *)arg;
  int           rk   = rank;
  pmg_host_comm hc   = {rank, SAN_RANKS, san_allgather, &rk};
  const int64_t n[3] = {35, 70, 140};
  intptr_t      fail = 0;
  pmg_rbh       h    = NULL;
  if (pmg_rbh_create(&hc, 3, 40, &h)) return (void *)1;
  if (pmg_rbh_set_coloring(h, PMG_COLORING_ITERATED)) return (void *)1; /* first-fit + the second round with its per-class exchanges (round 4): both under the sanitizers; the build checks the result */
  for (int l = 0; l < 3 && !fail; ++l) {
    const int64_t r0 = n[l] * rank / SAN_RANKS, r1 = n[l] * (rank + 1) / SAN_RANKS, nl = r1 - r0;
    int64_t      *rp = malloc(sizeof(int64_t) * (size_t)(nl + 1)), *ci = malloc(sizeof(int64_t) * (size_t)(5 * nl));
    double       *v  = malloc(sizeof(double) * (size_t)(5 * nl));
    int64_t       q  = 0;
    rp[0] = 0;
    for (int64_t r = r0; r < r1; ++r) {
      /* long-range couplings across the ranks, STRUCTURALLY SYMMETRIC (both are involutions: if r lists c, c lists r): a
         multicolour sweep is race-free only if no row shares its colour with a column it reads, first fit guarantees that
         for symmetric patterns, and pmg_rbh_build now checks it (pmg_rowblock_check_coloring) */
      const int64_t cand[5] = {r - 1, r, r + 1, n[l] - 1 - r, n[l] % 2 == 0 ? (r + n[l] / 2) % n[l] : -1};
      for (int c = 0; c < 5; ++c) {
        int dup = cand[c] < 0 || cand[c] >= n[l];
        for (int d = 0; d < c && !dup; ++d) dup = cand[d] == cand[c];
        if (!dup) ci[q] = cand[c], v[q] = cand[c] == r ? 4.0 : -0.5, ++q;
      }
      rp[r - r0 + 1] = q;
    }
    if (pmg_rbh_set_level_operator(h, l, n[l], r0, nl, rp, ci, v, 64)) fail = 2;
    free(rp), free(ci), free(v);
    if (l >= 1 && !fail) { /* P_l: fine row r -> coarse r / 2 */
      int32_t *prp = malloc(sizeof(int32_t) * (size_t)(nl + 1)), *pci = malloc(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1));
      double  *pv  = malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1));
      for (int64_t r = 0; r <= nl; ++r) prp[r] = (int32_t)r;
      for (int64_t r = 0; r < nl; ++r) pci[r] = (int32_t)((r0 + r) / 2), pv[r] = 1.0;
      if (pmg_rbh_set_level_interpolation(h, l, nl, prp, pci, pv, 32)) fail = 3;
      free(prp), free(pci), free(pv);
    }
  }
  if (!fail && pmg_rbh_build(h)) fail = 4;
  int32_t nl = 0, fold = 0;
  if (!fail && (pmg_rbh_get_info(h, &nl, &fold) || nl != 3 || fold != 1)) fail = 5;
  for (int l = 0; l < 3 && !fail; ++l) {
    pmg_rbh_level_view vw;
    if (pmg_rbh_get_level(h, l, &vw)) fail = 6;
    else if (!vw.replicated) { /* every local index inside the local matrix, every ghost refreshed exactly once */
      for (int32_t k = 0; k < vw.rp[vw.nlocal] && !fail; ++k)
        if (vw.ci[k] < 0 || vw.ci[k] >= vw.nlocal) fail = 7;
      for (int32_t k = 0; k < vw.R_rp[vw.R_nrows] && !fail; ++k)
        if (vw.R_ci[k] < 0 || vw.R_ci[k] >= vw.nlocal) fail = 8;
      if (vw.recv_ptr[vw.ncolors] != vw.nghost) fail = 9;
    } else if (vw.nlocal != n[l]) fail = 10;
  }
  pmg_mgmc mg = NULL;
  if (!fail && pmg_rbh_create_mgmc(h, (pmg_dist)&hc, &mg) != PMG_ERR_SUP) fail = 11; /* reaches the (stubbed) device layer and unwinds */
  pmg_rbh_destroy(&h);
  return (void *)fail;
}

/* pmg_dist_create_comm when the creation fails on ONE rank: an error on every rank, no rank left in a collective the
   others skipped (the threads' all-gather is a pair of barriers: a mismatch hangs this program until the test's timeout) */
static void *san_rank_create(void *arg)
{
  const int     rank = *(int *)arg;
  int           rk   = rank;
  pmg_host_comm hc   = {rank, SAN_RANKS, san_allgather, &rk};
  pmg_dist      d    = (pmg_dist)&hc; /* must come back NULL */
  const int     st   = pmg_dist_create_comm(&hc, "ipc", NULL, NULL, &d);
  if (st == 0) return (void *)21;
  if (d != NULL) return (void *)22;
  /* the communicator is still usable: everybody passes one more all-gather */
  int32_t z = rank, all[SAN_RANKS];
  if (hc.allgather(hc.ctx, &z, sizeof z, all) || all[0] != 0 || all[SAN_RANKS - 1] != SAN_RANKS - 1) return (void *)23;
  return NULL;
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
  /* ---- pmg_rowblock.c on three ranks (threads) ---- */
  {
    pthread_t th[SAN_RANKS];
    int       ids[SAN_RANKS];
    REQUIRE(pthread_barrier_init(&san_bar, NULL, SAN_RANKS) == 0);
    for (int r = 0; r < SAN_RANKS; ++r) {
      ids[r] = r;
      REQUIRE(pthread_create(&th[r], NULL, san_rank, &ids[r]) == 0);
    }
    for (int r = 0; r < SAN_RANKS; ++r) {
      void *ret = NULL;
      REQUIRE(pthread_join(th[r], &ret) == 0);
      if (ret) fprintf(stderr, "host_san: row-block rank %d failed at step %ld: %s\n", r, (long)(intptr_t)ret, pmg_last_error_string());
      REQUIRE(ret == NULL);
    }
    san_fail_on_rank = 1;
    for (int r = 0; r < SAN_RANKS; ++r) REQUIRE(pthread_create(&th[r], NULL, san_rank_create, &ids[r]) == 0);
    for (int r = 0; r < SAN_RANKS; ++r) {
      void *ret = NULL;
      REQUIRE(pthread_join(th[r], &ret) == 0);
      if (ret) fprintf(stderr, "host_san: transport-creation rank %d failed at step %ld\n", r, (long)(intptr_t)ret);
      REQUIRE(ret == NULL);
    }
    REQUIRE(san_disconnects[0] == 1 && san_destroys[0] == 1 && san_disconnects[2] == 1 && san_destroys[2] == 1 && san_destroys[1] == 0);
    san_fail_on_rank = -1;
    /* MatMPIAIJGetSeqAIJ's blocks -> rows, both orders, 32-bit indices */
    const int32_t ad_rp[3] = {0, 2, 4}, ad_ci[4] = {0, 1, 0, 1}, ao_rp[3] = {0, 1, 3}, ao_ci[3] = {1, 0, 2}, ga[3] = {0, 1, 6};
    const double  ad_v[4] = {4, -1, -1, 4}, ao_v[3] = {-2, -3, -5};
    int64_t       mrp[3], mci[7];
    double        mv[7];
    REQUIRE(pmg_rowblock_merge_mpiaij(2, 3, ad_rp, ad_ci, ad_v, ao_rp, ao_ci, ao_v, ga, 32, PMG_ROWBLOCK_ORDER_GLOBAL, mrp, mci, mv) == 0);
    REQUIRE(mrp[2] == 7 && mci[0] == 1 && mci[1] == 3 && mci[2] == 4 && mci[3] == 0 && mci[6] == 6 && mv[3] == -3 && mv[6] == -5);
    REQUIRE(pmg_rowblock_merge_mpiaij(2, 3, ad_rp, ad_ci, ad_v, ao_rp, ao_ci, ao_v, ga, 32, PMG_ROWBLOCK_ORDER_MPIAIJ, mrp, mci, mv) == 0);
    REQUIRE(mci[0] == 3 && mci[2] == 1 && mci[3] == 3 && mci[5] == 0 && mci[6] == 6);
    REQUIRE(pmg_rowblock_merge_mpiaij(2, 3, ad_rp, ad_ci, ad_v, ao_rp, ao_ci, ao_v, ga, 48, 0, mrp, mci, mv) == PMG_ERR_ARG_OUTOFRANGE);
  }
  printf("host_san ok\n");
  return 0;
}
