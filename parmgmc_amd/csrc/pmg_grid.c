/* Matrix-free MCSOR / Gibbs sampler on a DMDA grid -- host side (C11).
 *
 * Mirrors the MCSOR object of the reference (src/mc_sor.c:41-58: omega, omega_changed, type, idiag, colouring)
 * and the sample loops that drive it (src/pc_mcgibbs.c:155-188, src/pc_sorgibbs.c:76-134) for the operator of
 * MatAssembleShiftedLaplaceFD (src/problems.c:14-75).  The operator is never stored: its diagonal takes one
 * of 7 values, tabulated here with the reference's rounding sequence and handed to the kernel by value.
 */
#include "pmg_internal.h"
#include <stdlib.h>
#include <math.h>

struct pmg_grid_s {
  pmgk_grid_layout L;
  double           kappa, h2;
  double           omega;
  int              omega_changed;
  int              type;
  double           diag[8], idiag[8], sqrtd[8];
  double          *b_cv, *y_cv; /* scratch cvecs for the natural-order entry points */
  pmg_lrc          lrc;         /* MATLRC: rank-k update B S B^T (src/mc_sor.c:572-595) */
};

/* idiag = (1/d)*omega (MCSORUpdateIDiag, src/mc_sor.c:114-124) and sqrt|d| (VecSqrtAbs, src/pc_mcgibbs.c:149) */
static void pmg_grid_update_tables(pmg_grid g)
{
  for (int nn = 0; nn < 8; ++nn) {
    double d = g->kappa * g->kappa;
    for (int q = 0; q < nn; ++q) d += g->h2; /* repeated addition, src/problems.c:27-58 */
    g->diag[nn]  = d;
    double t     = 1.0 / d;
    g->idiag[nn] = t * g->omega;
    g->sqrtd[nn] = sqrt(fabs(d));
  }
  g->omega_changed = 0;
}

/* Line stride of a colour array in doubles (>= ceil(nx/2), even: every thread moves 16 bytes; a function of the GLOBAL
   extents, so that the slabs of all ranks agree on the plane size).  Whole 128-byte lines (a multiple of 16 doubles) by
   default; the tightest even stride where the two vectors of a sweep (b, y: 16 N bytes) fit the 256 MiB Infinity Cache --
   there the padding of a 2^k+1 line (257^3: 144 doubles for 129) is traffic the cache would otherwise not see, measured
   -3 % per 257^3 V-cycle sample and -2 % at 129^3 / 193^3, while beyond the cache (385^3, 449^3, 513^3) lines that straddle
   128-byte boundaries cost 5-8 % per sweep (tools/stridebench.py, three interleaved runs each).  PMG_GRID_SX_ALIGN forces an
   alignment in doubles (2 = tightest, 16 = whole lines). */
int32_t pmg_grid_line_stride(int32_t nx, int32_t ny, int32_t nzg)
{
  const int32_t half = (nx + 1) / 2;
  const char   *e    = getenv("PMG_GRID_SX_ALIGN");
  int32_t       al   = e ? atoi(e) : (16.0 * nx * ny * nzg <= 1.02 * 268435456.0 ? 2 : 16);
  if (al < 2 || (al & 1)) al = 16;
  return (half + al - 1) / al * al;
}

pmg_status pmg_grid_create(int32_t nx, int32_t ny, int32_t nzg, int32_t kz0, int32_t nz, double kappa, pmg_grid *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(nx >= 2 && ny >= 1 && nzg >= 1, PMG_ERR_ARG_OUTOFRANGE, "grid %d x %d x %d: need nx >= 2 (h2 = 1/(nx-1)^2, src/problems.c:24), ny, nz >= 1", nx, ny, nzg);
  PMG_CHECK(kz0 >= 0 && nz >= 1 && kz0 + nz <= nzg, PMG_ERR_ARG_OUTOFRANGE, "owned planes [%d,%d) outside [0,%d)", kz0, kz0 + nz, nzg);
  PMG_CHECK((int64_t)ny * nzg < ((int64_t)1 << 32), PMG_ERR_ARG_OUTOFRANGE, "ny*nz must fit 32 bits (noise counter word)");
  pmg_grid g = (pmg_grid)calloc(1, sizeof *g);
  PMG_CHECK(g, PMG_ERR_MEM, "out of host memory");
  g->L.nx  = nx;
  g->L.ny  = ny;
  g->L.nz  = nz;
  g->L.kz0 = kz0;
  g->L.nzg = nzg;
  g->L.sx  = pmg_grid_line_stride(nx, ny, nzg);
  g->L.sp  = (int64_t)ny * g->L.sx; /* (round 3's PMG_GRID_SP_PAD experiment -- planes shifted off their alignment, tools/alignbench.py -- is gone from the production build: an odd value misaligned every 16-byte access) */
  g->L.cs  = (int64_t)(nz + 2) * g->L.sp;
  g->kappa = kappa;
  g->h2    = 1. / ((nx - 1) * (nx - 1)); /* integer product, then the divide: src/problems.c:24 */
  g->omega = 1.0;
  g->type  = PMG_SOR_FORWARD_SWEEP;
  pmg_grid_update_tables(g);
  *out = g;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_get_kernel_layout(pmg_grid g, pmgk_grid_layout *L)
{
  PMG_CHECK(g && L, PMG_ERR_ARG_NULL, "null argument");
  *L = g->L;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_destroy(pmg_grid *g)
{
  if (!g || !*g) return PMG_SUCCESS;
  pmg_lrc_destroy(&(*g)->lrc);
  pmg_dev_free((*g)->b_cv);
  pmg_dev_free((*g)->y_cv);
  free(*g);
  *g = NULL;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_set_omega(pmg_grid g, double omega)
{
  PMG_CHECK(g, PMG_ERR_ARG_NULL, "null grid");
  g->omega         = omega;
  g->omega_changed = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_set_sweep_type(pmg_grid g, int type)
{
  PMG_CHECK(g, PMG_ERR_ARG_NULL, "null grid");
  PMG_CHECK(pmg_sweep_type_ok(type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported"); /* src/mc_sor.c:427 */
  g->type = type;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_get_sweep_type(pmg_grid g, int *type)
{
  PMG_CHECK(g && type, PMG_ERR_ARG_NULL, "null argument");
  *type = g->type;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_get_num_colors(pmg_grid g, int32_t *ncolors)
{
  PMG_CHECK(g && ncolors, PMG_ERR_ARG_NULL, "null argument");
  *ncolors = ((int64_t)g->L.nx * g->L.ny * g->L.nzg > 1) ? 2 : 1;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_get_coloring(pmg_grid g, int32_t *colors)
{
  PMG_CHECK(g && colors, PMG_ERR_ARG_NULL, "null argument");
  for (int k = 0; k < g->L.nz; ++k)
    for (int j = 0; j < g->L.ny; ++j)
      for (int i = 0; i < g->L.nx; ++i) colors[i + (int64_t)g->L.nx * (j + (int64_t)g->L.ny * k)] = (i + j + k + g->L.kz0) & 1;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_cvec_len(pmg_grid g, int64_t *len)
{
  PMG_CHECK(g && len, PMG_ERR_ARG_NULL, "null argument");
  *len = 2 * g->L.cs;
  return PMG_SUCCESS;
}

/* cvec position of every owned point, in DMDA natural order (i fastest) */
pmg_status pmg_grid_get_layout(pmg_grid g, int64_t *pos_of_point)
{
  PMG_CHECK(g && pos_of_point, PMG_ERR_ARG_NULL, "null argument");
  for (int k = 0; k < g->L.nz; ++k)
    for (int j = 0; j < g->L.ny; ++j)
      for (int i = 0; i < g->L.nx; ++i) {
        const int c = (i + j + k + g->L.kz0) & 1;
        pos_of_point[i + (int64_t)g->L.nx * (j + (int64_t)g->L.ny * k)] = (int64_t)c * g->L.cs + (int64_t)(k + 1) * g->L.sp + (int64_t)j * g->L.sx + (i >> 1);
      }
  return PMG_SUCCESS;
}

pmg_status pmg_grid_to_cvec(pmg_grid g, const double *nat, double *cvec, void *stream)
{
  PMG_CHECK(g && nat && cvec, PMG_ERR_ARG_NULL, "null argument");
  PMG_KERNEL(pmgk_grid_to_cvec(&g->L, nat, cvec, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_grid_from_cvec(pmg_grid g, const double *cvec, double *nat, void *stream)
{
  PMG_CHECK(g && nat && cvec, PMG_ERR_ARG_NULL, "null argument");
  PMG_KERNEL(pmgk_grid_from_cvec(&g->L, cvec, nat, stream));
  return PMG_SUCCESS;
}

static void pmg_grid_fill_op(pmg_grid g, pmgk_grid_op *op, int noisy, int scaled, uint64_t seed, uint64_t sweep)
{
  if (g->omega_changed) pmg_grid_update_tables(g);
  memset(op, 0, sizeof *op);
  op->h2              = g->h2;
  op->one_minus_omega = 1. - g->omega;
  /* VecScale(sqrtdiag, sqrt((2-omega)/omega)), src/pc_mcgibbs.c:150 */
  const double s = sqrt((2 - g->omega) / g->omega);
  for (int nn = 0; nn < 8; ++nn) {
    op->idiag[nn]    = g->idiag[nn];
    op->diag[nn]     = g->diag[nn];
    op->sqrtdiag[nn] = scaled ? g->sqrtd[nn] * s : g->sqrtd[nn];
  }
  op->key0         = (uint32_t)seed;
  op->key1         = (uint32_t)(seed >> 32);
  op->sweep        = sweep;
  op->noisy        = noisy;
  op->omega_is_one = g->omega == 1.0;
}

/* One forward (colours 0,1) or backward (colours 1,0; src/mc_sor.c:274) sweep, with the low-rank noise term and
   repair (src/pc_mcgibbs.c:130-140, src/mc_sor.c:101-112) when a MATLRC update is attached. */
static pmg_status pmg_grid_one_sweep(pmg_grid g, int dir, int noisy, int scaled, uint64_t seed, uint64_t sweep, const double *b, double *y, void *stream)
{
  if (g->lrc && noisy) PMG_CALL(pmg_lrc_rhs(g->lrc, b, seed, sweep, &b, stream));
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, noisy, scaled, seed, sweep);
  const int c0 = dir == PMG_SOR_FORWARD_SWEEP ? 0 : 1;
  pmg_trace_begin(PMG_EVENT_MULTICOL_SOR); /* PetscLogEventBegin(MULTICOL_SOR), src/mc_sor.c:221 */
  int rc = pmgk_grid_color_sweep(&g->L, &op, c0, 0, g->L.nz, 1, NULL, b, y, stream);
  if (!rc) rc = pmgk_grid_color_sweep(&g->L, &op, 1 - c0, 0, g->L.nz, 1, NULL, b, y, stream);
  pmg_trace_end();
  PMG_KERNEL(rc);
  if (g->lrc && noisy) PMG_CALL(pmg_lrc_rhs_done(g->lrc, stream));
  if (g->lrc) PMG_CALL(pmg_lrc_post(g->lrc, dir, y, stream));
  return PMG_SUCCESS;
}

static pmg_status grid_det_sweep(void *ctx, int dir, const double *b, double *y, void *stream)
{
  pmg_grid     g = (pmg_grid)ctx;
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, 0, 0, 0, 0);
  const int c0 = dir == PMG_SOR_FORWARD_SWEEP ? 0 : 1;
  PMG_KERNEL(pmgk_grid_color_sweep(&g->L, &op, c0, 0, g->L.nz, 1, NULL, b, y, stream));
  PMG_KERNEL(pmgk_grid_color_sweep(&g->L, &op, 1 - c0, 0, g->L.nz, 1, NULL, b, y, stream));
  return PMG_SUCCESS;
}

/* the same from B already in cvec layout on the device (cvec_len x k); used by the multigrid set-up, which restricts
   the observation vectors level by level on the device */
pmg_status pmg_grid_set_lowrank_dev(pmg_grid g, int32_t k, const double *B_cvec_dev, const double *S_host)
{
  PMG_CHECK(g && B_cvec_dev && S_host, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(g->L.nz == g->L.nzg, PMG_ERR_SUP, "low-rank updates are single-device");
  pmg_lrc_destroy(&g->lrc);
  return pmg_lrc_build_dev(&g->lrc, k, 2 * g->L.cs, B_cvec_dev, S_host, grid_det_sweep, g, NULL, NULL);
}

/* MCSORSetUp's MATLRC branch (src/mc_sor.c:572-595) for the grid operator: B is (nx*ny*nz) x k column-major in
   DMDA natural order, S the k diagonal entries of Sigma^-1.  Uses the CURRENT omega.  k = 0 removes the update.
   Single-device grids only (the dense B^T y reduction over devices is not built). */
pmg_status pmg_grid_set_lowrank(pmg_grid g, int32_t k, const double *B_host, const double *S_host)
{
  PMG_CHECK(g, PMG_ERR_ARG_NULL, "null grid");
  PMG_CHECK(g->L.nz == g->L.nzg, PMG_ERR_SUP, "low-rank updates are single-device");
  pmg_lrc_destroy(&g->lrc);
  if (k == 0) return PMG_SUCCESS;
  const int64_t n   = (int64_t)g->L.nx * g->L.ny * g->L.nz;
  int64_t      *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  PMG_CHECK(pos, PMG_ERR_MEM, "out of host memory");
  pmg_status st = pmg_grid_get_layout(g, pos);
  if (!st) st = pmg_lrc_build(&g->lrc, k, 2 * g->L.cs, (int32_t)n, B_host, pos, S_host, grid_det_sweep, g);
  free(pos);
  return st;
}

pmg_status pmg_grid_sweep_color_cvec(pmg_grid g, int color, int noisy, int scaled, uint64_t seed, uint64_t counter, const double *b, double *y, void *stream)
{
  PMG_CHECK(g && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(color == 0 || color == 1, PMG_ERR_ARG_OUTOFRANGE, "colour %d", color);
  PMG_CHECK(!noisy || scaled || g->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, noisy != 0, scaled, seed, counter);
  PMG_KERNEL(pmgk_grid_color_sweep(&g->L, &op, color, 0, g->L.nz, 1, NULL, b, y, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_grid_sweep_color_planes_cvec(pmg_grid g, int color, int32_t kbegin, int32_t kcount, int noisy, int scaled, uint64_t seed, uint64_t counter, const double *b, double *y, void *stream)
{
  PMG_CHECK(g && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(color == 0 || color == 1, PMG_ERR_ARG_OUTOFRANGE, "colour %d", color);
  PMG_CHECK(kbegin >= 0 && kcount >= 0 && kbegin + kcount <= g->L.nz, PMG_ERR_ARG_OUTOFRANGE, "planes [%d,%d) outside the %d owned planes", kbegin, kbegin + kcount, g->L.nz);
  PMG_CHECK(!noisy || scaled || g->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, noisy != 0, scaled, seed, counter);
  PMG_KERNEL(pmgk_grid_color_sweep(&g->L, &op, color, kbegin, kcount, 1, NULL, b, y, stream));
  return PMG_SUCCESS;
}

/* both slab faces (planes 0 and nz-1) of one colour in ONE launch: the part of a colour pass that reads ghost planes */
pmg_status pmg_grid_sweep_color_faces_cvec(pmg_grid g, int color, int noisy, int scaled, uint64_t seed, uint64_t counter, const pmgk_grid_halo *halo, const double *b, double *y, void *stream)
{
  PMG_CHECK(g && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(color == 0 || color == 1, PMG_ERR_ARG_OUTOFRANGE, "colour %d", color);
  PMG_CHECK(!noisy || scaled || g->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, noisy != 0, scaled, seed, counter);
  const int nz = g->L.nz;
  PMG_KERNEL(pmgk_grid_color_sweep(&g->L, &op, color, 0, nz > 1 ? 2 : 1, nz > 1 ? nz - 1 : 1, halo, b, y, stream));
  return PMG_SUCCESS;
}

/* ALL owned planes of one colour in one launch, face planes first, with the halo hand-shake inside the kernel
   (halo->full must be set; see pmgk_grid_halo) */
pmg_status pmg_grid_sweep_color_halo_cvec(pmg_grid g, int color, int noisy, int scaled, uint64_t seed, uint64_t counter, const pmgk_grid_halo *halo, const double *b, double *y, void *stream)
{
  PMG_CHECK(g && b && y && halo && halo->full, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(color == 0 || color == 1, PMG_ERR_ARG_OUTOFRANGE, "colour %d", color);
  PMG_CHECK(!noisy || scaled || g->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, noisy != 0, scaled, seed, counter);
  PMG_KERNEL(pmgk_grid_color_sweep(&g->L, &op, color, 0, g->L.nz, 1, halo, b, y, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_grid_halo_plane(pmg_grid g, int color, int side, int64_t *owned_offset, int64_t *ghost_offset, int64_t *count)
{
  PMG_CHECK(g && owned_offset && ghost_offset && count, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK((color == 0 || color == 1) && (side == 0 || side == 1), PMG_ERR_ARG_OUTOFRANGE, "colour %d side %d", color, side);
  const int64_t base = (int64_t)color * g->L.cs;
  *owned_offset      = base + (int64_t)(side == 0 ? 1 : g->L.nz) * g->L.sp;     /* owned plane 0 or nz-1 */
  *ghost_offset      = base + (int64_t)(side == 0 ? 0 : g->L.nz + 1) * g->L.sp; /* ghost plane -1 or nz  */
  *count             = g->L.sp;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_apply_cvec(pmg_grid g, const double *b, double *y, void *stream)
{
  PMG_CHECK(g && b && y, PMG_ERR_ARG_NULL, "null argument");
  if (g->type == PMG_SOR_SYMMETRIC_SWEEP) { /* src/mc_sor.c:223-232 */
    PMG_CALL(pmg_grid_one_sweep(g, PMG_SOR_FORWARD_SWEEP, 0, 0, 0, 0, b, y, stream));
    PMG_CALL(pmg_grid_one_sweep(g, PMG_SOR_BACKWARD_SWEEP, 0, 0, 0, 0, b, y, stream));
  } else {
    PMG_CALL(pmg_grid_one_sweep(g, g->type, 0, 0, 0, 0, b, y, stream));
  }
  return PMG_SUCCESS;
}

pmg_status pmg_grid_sample_cvec(pmg_grid g, const double *b, double *y, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(g && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CHECK(scaled || g->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  uint64_t ctr = counter0;
  for (int it = 0; it < its; ++it) {
    if (g->type == PMG_SOR_SYMMETRIC_SWEEP) { /* two sweeps, two fresh draws: src/pc_mcgibbs.c:172-181 */
      PMG_CALL(pmg_grid_one_sweep(g, PMG_SOR_FORWARD_SWEEP, 1, scaled, seed, ctr++, b, y, stream));
      PMG_CALL(pmg_grid_one_sweep(g, PMG_SOR_BACKWARD_SWEEP, 1, scaled, seed, ctr++, b, y, stream));
    } else {
      PMG_CALL(pmg_grid_one_sweep(g, g->type, 1, scaled, seed, ctr++, b, y, stream));
    }
  }
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}

pmg_status pmg_grid_residual_cvec(pmg_grid g, const double *b, const double *y, double *r, void *stream)
{
  PMG_CHECK(g && b && y && r, PMG_ERR_ARG_NULL, "null argument");
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, 0, 0, 0, 0);
  PMG_KERNEL(pmgk_grid_residual(&g->L, &op, b, y, r, stream));
  if (g->lrc) PMG_CALL(pmg_lrc_residual_sub(g->lrc, y, r, stream)); /* MatMult of the MATLRC operator */
  return PMG_SUCCESS;
}

pmg_lrc pmg_grid_lrc(pmg_grid g) { return g ? g->lrc : NULL; }

pmg_status pmg_grid_residual_restrict(pmg_grid g, const double *b, const double *y, const double *ylo2, const double *yhi2, const pmgk_st27_dims *C, double *b_coarse, int *done, void *stream)
{
  PMG_CHECK(g && b && y && C && b_coarse && done, PMG_ERR_ARG_NULL, "null argument");
  *done = 0; /* a low-rank term is the caller's: pmg_lrc_residual_sub_restricted with pmg_grid_lrc(g) */
  pmgk_grid_op op;
  pmg_grid_fill_op(g, &op, 0, 0, 0, 0);
  const int rc = pmgk_grid_residual_restrict(&g->L, &op, C, b, y, ylo2, yhi2, b_coarse, stream);
  if (rc < 0) return PMG_SUCCESS;
  PMG_CHECK(rc == 0, PMG_ERR_GPU, "kernel launch failed: pmgk_grid_residual_restrict");
  *done = 1;
  return PMG_SUCCESS;
}

/* would pmg_grid_residual_restrict run its kernel for this slab and coarse level? (no launch) */
int pmg_grid_residual_restrict_applies(pmg_grid g, const pmgk_st27_dims *C, int have_lo2, int have_hi2)
{
  return g && C ? pmgk_grid_residual_restrict_applies(&g->L, C, have_lo2, have_hi2) : 0;
}

static pmg_status pmg_grid_scratch(pmg_grid g)
{
  if (!g->b_cv) PMG_CALL(pmg_dev_alloc((void **)&g->b_cv, sizeof(double) * (size_t)(2 * g->L.cs)));
  if (!g->y_cv) PMG_CALL(pmg_dev_alloc((void **)&g->y_cv, sizeof(double) * (size_t)(2 * g->L.cs)));
  return PMG_SUCCESS;
}

pmg_status pmg_grid_apply(pmg_grid g, const double *b_nat, double *y_nat, void *stream)
{
  PMG_CHECK(g && b_nat && y_nat, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(pmg_grid_scratch(g));
  PMG_CALL(pmg_grid_to_cvec(g, b_nat, g->b_cv, stream));
  PMG_CALL(pmg_grid_to_cvec(g, y_nat, g->y_cv, stream));
  PMG_CALL(pmg_grid_apply_cvec(g, g->b_cv, g->y_cv, stream));
  PMG_CALL(pmg_grid_from_cvec(g, g->y_cv, y_nat, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_grid_sample(pmg_grid g, const double *b_nat, double *y_nat, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(g && b_nat && y_nat, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(pmg_grid_scratch(g));
  PMG_CALL(pmg_grid_to_cvec(g, b_nat, g->b_cv, stream));
  PMG_CALL(pmg_grid_to_cvec(g, y_nat, g->y_cv, stream));
  PMG_CALL(pmg_grid_sample_cvec(g, g->b_cv, g->y_cv, its, scaled, seed, counter0, counter_out, stream));
  PMG_CALL(pmg_grid_from_cvec(g, g->y_cv, y_nat, stream));
  return PMG_SUCCESS;
}
