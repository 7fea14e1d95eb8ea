/* Multigrid Monte Carlo sampler on a DMDA hierarchy -- host side (C11).
 *
 * Mirrors PCGAMGMC with `-pc_gamgmc_mg_type mg` (reference src/pc_gamgmc.c) together with the part of PETSc's PCMG
 * it drives (third party, restated from PETSc's documented semantics -- parity unpinned by any reference fixture):
 *   - hierarchy: vertex-centred 2:1 coarsening nc = (nf-1)/2 + 1, Q1 (bi/tri-linear) interpolation as DMDA's
 *     DMCreateInterpolation builds it, restriction R = P^T, Galerkin coarse operators A_c = P^T A P
 *     (-pc_mg_galerkin both, injected at src/pc_gamgmc.c:345-349);
 *   - level "smoothers" = Gibbs samplers (default sorgibbs, KSPRICHARDSON, max_it 1: src/pc_gamgmc.c:318-334),
 *     coarse "solver" = exact Cholesky sampler (src/pc_gamgmc.c:336-342) or Gibbs sweeps (examples/ex1.c:35);
 *   - one V-cycle: pre-smooth, r = b - A x, b_c = P^T r, recurse from x_c = 0, x += P x_c, post-smooth;
 *   - the outer chain in correction form, PCApplyRichardson_GAMGMC (src/pc_gamgmc.c:227-264):
 *     first iteration from a zero guess y = MG(b), afterwards w = b - A y, y += MG(w); callback per sample.
 * The finest level is the matrix-free red-black grid operator (pmg_grid); coarser levels are 9/27-point Galerkin
 * matrices swept with the sliced-ELL kernel under the 4/8-colour parity colouring (red-black is not a valid
 * colouring of a 27-point stencil); transfers are CSR products between the levels' storage layouts.
 */
#include "pmg_internal.h"
#include <math.h>

typedef struct {
  int32_t  nr, nc;
  int32_t *rp, *ci;
  double  *v;
} hcsr;

static void hcsr_free(hcsr *m)
{
  free(m->rp);
  free(m->ci);
  free(m->v);
  memset(m, 0, sizeof *m);
}

typedef struct {
  int32_t   nx, ny, nz, n; /* global extents */
  int32_t   kz0, nzl;      /* owned planes (0, nz on a single device and on replicated levels) */
  int       padded;        /* plane-padded natural layout (class-stencil levels, Cholesky level): element (i,j,k) at off + i + nx (j + ny (k - kz0)) */
  int64_t   off;           /* = nx*ny when padded */
  int       distributed;   /* z-slab level of a multi-device hierarchy */
  int       is_grid;
  pmg_grid  g;
  pmg_mcsor mc;
  int64_t   ld;
  double   *b, *x, *r;
  int       x_zeroed; /* the restriction kernel has set the zero guess already */
  int       x_unset; /* the iterate is zero but the memset was skipped: the next out-of-place sweep starts from NULL */
  double   *y2lo, *y2hi; /* z-slab grid level: the iterate's planes kz0 - 2 and kz0 + nz + 1 (colour 0 plane, colour 1 plane) for the fused residual + restriction */
  int       rr_slab;     /* ... which every rank can run (agreed from the slab cuts) */
  double   *x2; /* second buffer of the out-of-place class-stencil sweep (single-device levels): x and x2 swap after every directional sweep */
  /* transfers to the next coarser level, in layout numbering on the device */
  int32_t  P_nrows, R_nrows;
  int32_t *cpos_dev; /* grid level only: layout position of every point of the next coarser level */
  /* class-stencil form of a structured Galerkin level (natural-order vectors) */
  int       is_st27, nat_transfer; /* nat_transfer: this level and the next coarser one are both in padded natural order */
  int       grid_transfer;         /* matrix-free Q1 transfers from this grid level (cpos_dev == NULL: padded natural coarse level) */
  pmgk_st27 st;
  double   *st_coef, *st_idiag, *st_sqrtd, *st_sqrtd_scaled;
  pmg_lrc   lrc; /* MATLRC update of a class-stencil level (grid / sliced-ELL levels keep theirs inside g / mc) */
  int32_t *P_rowpos, *P_rowptr, *P_col, *R_rowpos, *R_rowptr, *R_col;
  double  *P_val, *R_val;
  /* optional host copies (natural numbering) for inspection */
  hcsr A_host, P_host;
  /* caller-supplied hierarchy (borrowed host CSR until set-up) */
  hcsr A_user, P_user;
  int32_t *A_rp_own, *A_ci_own, *P_rp_own, *P_ci_own; /* 32-bit copies of 64-bit PetscInt arrays, freed after set-up */
  /* ROW BLOCK of a distributed hierarchy (pmg_mgmc_set_level_rowblock): A_user is this rank's rows in LOCAL numbering --
     owned rows first, then one identity row per ghost (every row of another rank that this rank's operator, restriction
     or the finer level's interpolation reads); the plan lists are host copies until set-up builds `dm` from them */
  int           rb;
  int32_t       rb_nowned, rb_ncolors;
  int64_t       rb_row0;
  int32_t      *rb_colors, *rb_send_idx, *rb_recv_src, *rb_recv_idx;
  int64_t      *rb_send_ptr, *rb_recv_ptr, *rb_counts;
  hcsr          R_user; /* rows of the restriction INTO the next coarser level that this rank owns there (borrowed) */
  pmg_distmcsor dm;
  int64_t       A_nnz, P_nnz; /* stored entries of a sliced-ELL level's operator / of its CSR interpolation (traffic accounting) */
} mg_level;

struct pmg_mgmc_s {
  int       nlevels;
  mg_level *lv; /* lv[0] = coarsest */
  double    kappa, omega;
  int       nu, scaled, sweep_type;
  int       coarse_type, coarse_its; /* 0 = cholsampler, 1 = Gibbs sweeps */
  int       keep_host, is_setup, user_hier;
  int       no_fused;        /* 1: residual and restriction as two kernels everywhere (pmg_mgmc_set_fused_transfers(mg, 0)) */
  int       correction_form; /* 1: w = b - A y, y += MG(w) literally (src/pc_gamgmc.c:253-256); 0: the same cycle run in place on (b, y) */
  pmg_chol  chol;
  double   *y_lay, *b_lay;
  /* MATLRC fine operator A + B S B^T (host copies until set-up; src/pc_gamgmc.c:157-196) */
  int32_t   lrc_k;
  double   *lrc_B, *lrc_S;
  int       aij_coloring; /* rule of the AIJ levels (pmg_mgmc_set_coloring); PMG_COLORING_GREEDY = 0 */
  double   *eta_batch; /* device: the low-rank noise terms of one cycle, drawn together (mg_draw_lowrank_noise) */
  int       eta_batch_mode; /* 0: not asked yet, 1: on, -1: PMG_LRC_BATCH=0 when this sampler ran its first cycle */
  int       own_grid; /* the fine grid operator was created here (not handed in with a slab) */
  /* multi-device: z-slabs of the fine grid, one rank per device (borrowed dist object); cuts[l*(nranks+1) + r] =
     first plane of rank r on level l */
  pmg_dist  dist;
  int32_t   rank, nranks;
  int32_t  *cuts;
  int32_t   n_io; /* length of the caller's fine-level vectors (the owned planes) */
  /* row-block distributed caller-supplied hierarchy: transport (borrowed) and the row blocks of the replicated coarsest level */
  pmg_dist  rb_dist;
  int64_t  *rb_c0_starts; /* row blocks of the highest REPLICATED level (rb_fold - 1): who restricts which of its rows */
  int       rb_fold;      /* lowest row-block level; the levels below are replicated on every rank */
  int32_t  *rb_fold_pos, *rb_fold_iota; /* device: layout position of natural row q of level rb_fold - 1, and 0, 1, 2, ... */
  double   *rb_fold_buf;                /* device: that level's vector in natural order (the all-gather buffer) */
};

typedef struct {
  double coef[27 * 27];
  int    have[27];
} st27_table;

#define MG_DRAWS_PER_SAMPLE 64u

/* ---------------------------------------------------------------------------------------------------- */
/* host sparse tools                                                                                    */
/* ---------------------------------------------------------------------------------------------------- */

/* Q1 interpolation from the (ncx,ncy,ncz) grid to the (nfx,nfy,nfz) grid, natural ordering, columns ascending.
   Per direction: fine 2I coincides with coarse I (weight 1), fine 2I+1 lies midway (1/2, 1/2); a direction with
   one point is not coarsened. */
static pmg_status q1_interp(const int32_t nf[3], const int32_t nc[3], hcsr *P)
{
  const int64_t nrow = (int64_t)nf[0] * nf[1] * nf[2];
  P->nr              = (int32_t)nrow;
  P->nc              = nc[0] * nc[1] * nc[2];
  P->rp              = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nrow + 1));
  /* count */
  int64_t nnz = 0;
  int64_t cnt[3][2]; /* per direction: number of fine points with 1 / 2 contributions */
  for (int d = 0; d < 3; ++d) {
    if (nf[d] == nc[d]) {
      cnt[d][0] = nf[d];
      cnt[d][1] = 0;
    } else {
      cnt[d][0] = (nf[d] + 1) / 2;
      cnt[d][1] = nf[d] / 2;
    }
  }
  nnz   = (cnt[0][0] + 2 * cnt[0][1]) * (cnt[1][0] + 2 * cnt[1][1]) * (cnt[2][0] + 2 * cnt[2][1]);
  P->ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  P->v  = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  PMG_CHECK(P->rp && P->ci && P->v, PMG_ERR_MEM, "out of host memory for the interpolation");
  int64_t p = 0;
  for (int k = 0; k < nf[2]; ++k)
    for (int j = 0; j < nf[1]; ++j)
      for (int i = 0; i < nf[0]; ++i) {
        const int f[3] = {i, j, k};
        int       c0[3], m[3];
        double    w[3][2];
        for (int d = 0; d < 3; ++d) {
          if (nf[d] == nc[d]) { c0[d] = f[d]; m[d] = 1; w[d][0] = 1.0; }
          else if ((f[d] & 1) == 0) { c0[d] = f[d] / 2; m[d] = 1; w[d][0] = 1.0; }
          else { c0[d] = f[d] / 2; m[d] = 2; w[d][0] = 0.5; w[d][1] = 0.5; }
        }
        P->rp[i + (int64_t)nf[0] * (j + (int64_t)nf[1] * k)] = (int32_t)p;
        for (int c = 0; c < m[2]; ++c)
          for (int bq = 0; bq < m[1]; ++bq)
            for (int a = 0; a < m[0]; ++a) {
              P->ci[p] = (c0[0] + a) + nc[0] * ((c0[1] + bq) + nc[1] * (c0[2] + c));
              P->v[p]  = w[0][a] * w[1][bq] * w[2][c];
              ++p;
            }
      }
  P->rp[nrow] = (int32_t)p;
  return PMG_SUCCESS;
}

static pmg_status hcsr_transpose(const hcsr *A, hcsr *T)
{
  const int32_t nnz = A->rp[A->nr];
  T->nr             = A->nc;
  T->nc             = A->nr;
  T->rp             = (int32_t *)calloc((size_t)T->nr + 1, sizeof(int32_t));
  T->ci             = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  T->v              = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  PMG_CHECK(T->rp && T->ci && T->v, PMG_ERR_MEM, "out of host memory");
  for (int32_t k = 0; k < nnz; ++k) T->rp[A->ci[k] + 1]++;
  for (int32_t r = 0; r < T->nr; ++r) T->rp[r + 1] += T->rp[r];
  int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T->nr > 0 ? T->nr : 1));
  PMG_CHECK(fill, PMG_ERR_MEM, "out of host memory");
  memcpy(fill, T->rp, sizeof(int32_t) * (size_t)T->nr);
  for (int32_t r = 0; r < A->nr; ++r)
    for (int32_t k = A->rp[r]; k < A->rp[r + 1]; ++k) {
      const int32_t q = fill[A->ci[k]]++;
      T->ci[q]        = r;
      T->v[q]         = A->v[k];
    }
  free(fill);
  return PMG_SUCCESS;
}

/* row generator: either a stored CSR or the matrix-free 7-point operator of src/problems.c:14-75 */
typedef struct {
  const hcsr *A;
  int32_t     nx, ny, nz;
  double      kappa, h2, diag[8];
} rowsrc;

static int rowsrc_get(const rowsrc *s, int32_t row, int32_t *cols, double *vals)
{
  if (s->A) {
    const int32_t a = s->A->rp[row], n = s->A->rp[row + 1] - a;
    memcpy(cols, s->A->ci + a, sizeof(int32_t) * (size_t)n);
    memcpy(vals, s->A->v + a, sizeof(double) * (size_t)n);
    return n;
  }
  const int32_t i = row % s->nx, j = (row / s->nx) % s->ny, k = row / (s->nx * s->ny);
  int           n = 0, nn = (k > 0) + (j > 0) + (i > 0) + (i < s->nx - 1) + (j < s->ny - 1) + (k < s->nz - 1);
  if (k > 0) { cols[n] = row - s->nx * s->ny; vals[n++] = -s->h2; }
  if (j > 0) { cols[n] = row - s->nx; vals[n++] = -s->h2; }
  if (i > 0) { cols[n] = row - 1; vals[n++] = -s->h2; }
  cols[n] = row; vals[n++] = s->diag[nn];
  if (i < s->nx - 1) { cols[n] = row + 1; vals[n++] = -s->h2; }
  if (j < s->ny - 1) { cols[n] = row + s->nx; vals[n++] = -s->h2; }
  if (k < s->nz - 1) { cols[n] = row + s->nx * s->ny; vals[n++] = -s->h2; }
  return n;
}

static int cmp_i32(const void *a, const void *b) { return (*(const int32_t *)a > *(const int32_t *)b) - (*(const int32_t *)a < *(const int32_t *)b); }

/* C = P^T A P, fused: for coarse row I, for i in R_I, for (j,a) in A_i, for (J,p) in P_j: C[I,J] += r a p */
static pmg_status galerkin_rap(const rowsrc *A, int maxrow, const hcsr *P, const hcsr *R, hcsr *Cm)
{
  const int32_t nc = P->nc;
  Cm->nr = Cm->nc = nc;
  Cm->rp          = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
  size_t   cap    = (size_t)nc * 32 + 64;
  Cm->ci          = (int32_t *)malloc(sizeof(int32_t) * cap);
  Cm->v           = (double *)malloc(sizeof(double) * cap);
  double  *acc    = (double *)calloc((size_t)nc, sizeof(double));
  int32_t *mark   = (int32_t *)malloc(sizeof(int32_t) * (size_t)nc);
  int32_t *list   = (int32_t *)malloc(sizeof(int32_t) * (size_t)nc);
  int32_t *cols   = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxrow);
  double  *vals   = (double *)malloc(sizeof(double) * (size_t)maxrow);
  PMG_CHECK(Cm->rp && Cm->ci && Cm->v && acc && mark && list && cols && vals, PMG_ERR_MEM, "out of host memory in the Galerkin product");
  for (int32_t q = 0; q < nc; ++q) mark[q] = -1;
  size_t nnz = 0;
  for (int32_t I = 0; I < nc; ++I) {
    int32_t nl = 0;
    for (int32_t kr = R->rp[I]; kr < R->rp[I + 1]; ++kr) {
      const int32_t i  = R->ci[kr];
      const double  rv = R->v[kr];
      const int     na = rowsrc_get(A, i, cols, vals);
      for (int ka = 0; ka < na; ++ka) {
        const int32_t j  = cols[ka];
        const double  ra = rv * vals[ka];
        for (int32_t kp = P->rp[j]; kp < P->rp[j + 1]; ++kp) {
          const int32_t J = P->ci[kp];
          if (mark[J] != I) {
            mark[J]    = I;
            list[nl++] = J;
            acc[J]     = 0.0;
          }
          acc[J] += ra * P->v[kp];
        }
      }
    }
    qsort(list, (size_t)nl, sizeof(int32_t), cmp_i32);
    if (nnz + (size_t)nl > cap) {
      cap     = (cap + (size_t)nl) * 2;
      Cm->ci  = (int32_t *)realloc(Cm->ci, sizeof(int32_t) * cap);
      Cm->v   = (double *)realloc(Cm->v, sizeof(double) * cap);
      PMG_CHECK(Cm->ci && Cm->v, PMG_ERR_MEM, "out of host memory in the Galerkin product");
    }
    for (int32_t q = 0; q < nl; ++q) {
      Cm->ci[nnz] = list[q];
      Cm->v[nnz]  = acc[list[q]];
      ++nnz;
    }
    PMG_CHECK(nnz < 2147483647u, PMG_ERR_ARG_OUTOFRANGE, "coarse operator exceeds 32-bit nonzero count");
    Cm->rp[I + 1] = (int32_t)nnz;
  }
  free(acc);
  free(mark);
  free(list);
  free(cols);
  free(vals);
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* public                                                                                               */
/* ---------------------------------------------------------------------------------------------------- */

pmg_status pmg_mgmc_create_dmda(int32_t nx, int32_t ny, int32_t nz, double kappa, int32_t levels, pmg_mgmc *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(levels >= 2, PMG_ERR_ARG_OUTOFRANGE, "need at least 2 levels (got %d)", levels);
  PMG_CHECK(nx >= 3 && ny >= 1 && nz >= 1, PMG_ERR_ARG_OUTOFRANGE, "grid %d x %d x %d", nx, ny, nz);
  pmg_mgmc h = (pmg_mgmc)calloc(1, sizeof *h);
  PMG_CHECK(h, PMG_ERR_MEM, "out of host memory");
  h->lv = (mg_level *)calloc((size_t)levels, sizeof(mg_level));
  if (!h->lv) {
    free(h);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  h->nlevels    = levels;
  h->kappa      = kappa;
  h->omega      = 1.0;
  h->nu         = 1;                      /* -mg_levels_ksp_max_it 1, src/pc_gamgmc.c:324-328 */
  h->scaled     = 0;                      /* -mg_levels_pc_type sorgibbs, :330-334            */
  h->sweep_type = PMG_SOR_FORWARD_SWEEP;
  h->coarse_type = 0;                     /* -mg_coarse_pc_type cholsampler, :336-342         */
  h->coarse_its  = 1;
  int32_t d[3]   = {nx, ny, nz};
  for (int l = levels - 1; l >= 0; --l) {
    h->lv[l].nx = d[0];
    h->lv[l].ny = d[1];
    h->lv[l].nz = d[2];
    h->lv[l].n  = d[0] * d[1] * d[2];
    h->lv[l].nzl = d[2];
    if (l > 0)
      for (int q = 0; q < 3; ++q)
        if (d[q] > 1) {
          if ((d[q] - 1) % 2 != 0 || d[q] < 3) {
            const int32_t bad = d[q];
            free(h->lv);
            free(h);
            PMG_FAIL(PMG_ERR_ARG_SIZ, "level %d has %d points in direction %d: vertex-centred 2:1 coarsening needs (n-1) even and n >= 3 on every refined level (use 2^k+1 points)", l, bad, q);
          }
          d[q] = (d[q] - 1) / 2 + 1;
        }
  }
  h->n_io = nx * ny * nz;
  *out    = h;
  return PMG_SUCCESS;
}

/* The same sampler on z-slabs of the DMDA, one rank per device (the reference distributes every PCMG
   level over all MPI ranks and lets GAMG reduce the coarse grids to rank 0, src/pc_chols.c:38-47,272-282):
     - `g` is this rank's slab of the fine operator (pmg_grid_create with kz0 = cuts[rank], nz = cuts[rank+1] - kz0),
       `dist` the halo transport created on it; both stay the caller's;
     - a coarse plane K belongs to the owner of fine plane 2K, so coarse levels inherit the partition with no data
       motion; per level and cycle there is the sweeps' halo (one plane per z-parity phase and side) and one halo of
       the residual for the restriction; the prolongation also fills the fine ghost planes, from the coarse ghost
       planes, so it needs no exchange;
     - levels with at most PMG_MG_REPLICATE_BELOW (default 2^19) unknowns, or with fewer planes than ranks, are
       REPLICATED: their right-hand side is all-gathered once and every rank runs the remaining coarse part of the
       cycle redundantly -- the noise is a function of (seed, counter, global index), so all ranks compute the same
       bits and no scatter is needed on the way up.
   Samples are bit-identical to the single-device sampler for any number of ranks. */
pmg_status pmg_mgmc_create_dmda_slab(int32_t nx, int32_t ny, int32_t nz, double kappa, int32_t levels, pmg_grid g, pmg_dist dist, const int32_t *cuts, pmg_mgmc *out)
{
  PMG_CHECK(out && g && dist && cuts, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(pmg_mgmc_create_dmda(nx, ny, nz, kappa, levels, out));
  pmg_mgmc   h  = *out;
  pmg_status st = pmg_dist_get_info(dist, &h->rank, &h->nranks, NULL);
  pmgk_grid_layout L;
  if (!st) st = pmg_grid_get_kernel_layout(g, &L);
  if (!st && (cuts[0] != 0 || cuts[h->nranks] != nz)) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "cuts must run from 0 to nz = %d", nz);
  if (!st && (L.nx != nx || L.ny != ny || L.nzg != nz || L.kz0 != cuts[h->rank] || L.nz != cuts[h->rank + 1] - cuts[h->rank])) st = pmg_set_error(PMG_ERR_ARG_SIZ, __FILE__, __LINE__, "the grid slab does not match cuts[%d..%d] of a %d x %d x %d grid", h->rank, h->rank + 1, nx, ny, nz);
  const int top = levels - 1;
  if (!st) {
    h->cuts = (int32_t *)malloc(sizeof(int32_t) * (size_t)levels * (size_t)(h->nranks + 1));
    if (!h->cuts) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  }
  if (!st) {
    const int nr1 = h->nranks + 1;
    memcpy(h->cuts + (size_t)top * nr1, cuts, sizeof(int32_t) * (size_t)nr1);
    for (int l = top; l >= 1 && !st; --l) {
      if (h->lv[l].nz == h->lv[l - 1].nz) st = pmg_set_error(PMG_ERR_SUP, __FILE__, __LINE__, "z-slabs need a grid that is coarsened in z on every level");
      for (int r = 0; r < nr1; ++r) h->cuts[(size_t)(l - 1) * nr1 + r] = (h->cuts[(size_t)l * nr1 + r] + 1) / 2; /* plane K <-> fine plane 2K */
    }
    for (int r = 0; r < h->nranks && !st; ++r)
      if (cuts[r + 1] <= cuts[r]) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "rank %d owns no plane", r);
  }
  if (st) {
    pmg_mgmc_destroy(out);
    return st;
  }
  h->dist          = dist;
  h->lv[top].g     = g;
  h->lv[top].kz0   = L.kz0;
  h->lv[top].nzl   = L.nz;
  h->n_io          = nx * ny * L.nz;
  return PMG_SUCCESS;
}

/* A hierarchy handed over level by level: what PCGAMGMC finds inside PETSc's PCMG/PCGAMG after PCSetUp -- the
   level operators (PCMGGetSmoother + PCGetOperators) and interpolations (PCMGGetInterpolation), reference
   src/pc_gamgmc.c:165-176 -- e.g. a GAMG hierarchy of an unstructured P1 matrix.  Every level is swept with the
   sliced-ELL multicolour kernel (greedy colouring), transfers are CSR products. */
pmg_status pmg_mgmc_create_hierarchy(int32_t levels, pmg_mgmc *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(levels >= 2 && levels <= 64, PMG_ERR_ARG_OUTOFRANGE, "levels = %d", levels);
  pmg_mgmc h = (pmg_mgmc)calloc(1, sizeof *h);
  PMG_CHECK(h, PMG_ERR_MEM, "out of host memory");
  h->lv = (mg_level *)calloc((size_t)levels, sizeof(mg_level));
  if (!h->lv) {
    free(h);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  h->nlevels     = levels;
  h->user_hier   = 1;
  h->omega       = 1.0;
  h->nu          = 1;
  h->scaled      = 0;
  h->sweep_type  = PMG_SOR_FORWARD_SWEEP;
  h->coarse_type = 0;
  h->coarse_its  = 1;
  *out           = h;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_level_operator(pmg_mgmc h, int32_t level, int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals)
{
  PMG_CHECK(h && rowptr && colidx && vals, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->user_hier && !h->is_setup, PMG_ERR_ARG_WRONGSTATE, "level operators belong to pmg_mgmc_create_hierarchy, before set-up");
  PMG_CHECK(level >= 0 && level < h->nlevels && n >= 1, PMG_ERR_ARG_OUTOFRANGE, "level %d, n %d", level, n);
  mg_level *Lv = &h->lv[level];
  Lv->n        = n;
  Lv->nx       = n;
  Lv->ny = Lv->nz = 1;
  Lv->A_user.nr = Lv->A_user.nc = n;
  Lv->A_user.rp = (int32_t *)rowptr; /* borrowed, only read */
  Lv->A_user.ci = (int32_t *)colidx;
  Lv->A_user.v  = (double *)vals;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_level_interpolation(pmg_mgmc h, int32_t level, int32_t nrows, int32_t ncols, const int32_t *rowptr, const int32_t *colidx, const double *vals)
{
  PMG_CHECK(h && rowptr && colidx && vals, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->user_hier && !h->is_setup, PMG_ERR_ARG_WRONGSTATE, "interpolations belong to pmg_mgmc_create_hierarchy, before set-up");
  PMG_CHECK(level >= 1 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  mg_level *Lv  = &h->lv[level];
  Lv->P_user.nr = nrows;
  Lv->P_user.nc = ncols;
  Lv->P_user.rp = (int32_t *)rowptr;
  Lv->P_user.ci = (int32_t *)colidx;
  Lv->P_user.v  = (double *)vals;
  return PMG_SUCCESS;
}

/* ---- caller-supplied hierarchy distributed by ROW BLOCKS (the reference runs PCGAMGMC on any MATMPIAIJ,
   src/pc_gamgmc.c:157-223; MCSORApply_MPIAIJ src/mc_sor.c:298-381 is the level sampler) --------------------------------
   Every rank describes ITS rows:
   * level 0 (coarsest, exact sampler): the whole matrix on every rank (pmg_mgmc_set_level_operator) -- it is factored
     redundantly, the restricted right-hand side is all-gathered by the row blocks `coarse_starts`;
   * level l >= 1: pmg_mgmc_set_level_operator with the rank's rows in LOCAL numbering (owned rows 0 .. nowned-1 in the
     order of the global rows row0 .. row0+nowned-1, entries in the order of the global CSR row, then one identity row
     per ghost), and pmg_mgmc_set_level_rowblock with a globally valid distance-1 colouring of the owned rows and the
     ghost-update plan of pmg_distmcsor_create in LOCAL ROW indices;
   * pmg_mgmc_set_level_interpolation(l): the owned rows of P_l, columns in the local numbering of level l-1 (global
     indices for l-1 = 0); pmg_mgmc_set_level_restriction(l): the rows of R_l = P_l^T that this rank owns on level l-1,
     columns in the local numbering of level l, entries by ascending global fine row (the order of a transposition).
   Noise is keyed on global rows and every row keeps its global entry order: the chain is the single-device chain of
   pmg_mgmc_create_hierarchy bit for bit; with a low-rank update (pmg_mgmc_set_lowrank: this rank's rows of B) to rounding. */
pmg_status pmg_mgmc_set_rowblock_transport(pmg_mgmc h, pmg_dist dist, const int64_t *coarse_starts)
{
  PMG_CHECK(h && dist && coarse_starts, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->user_hier && !h->is_setup, PMG_ERR_ARG_WRONGSTATE, "row blocks belong to pmg_mgmc_create_hierarchy, before set-up");
  int32_t rank, nranks;
  PMG_CALL(pmg_dist_get_info(dist, &rank, &nranks, NULL));
  PMG_CHECK(nranks >= 1 && nranks <= 64, PMG_ERR_ARG_OUTOFRANGE, "%d ranks", nranks);
  free(h->rb_c0_starts);
  h->rb_c0_starts = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nranks + 1));
  PMG_CHECK(h->rb_c0_starts, PMG_ERR_MEM, "out of host memory");
  memcpy(h->rb_c0_starts, coarse_starts, sizeof(int64_t) * ((size_t)nranks + 1));
  for (int r = 0; r < nranks; ++r) PMG_CHECK(coarse_starts[r] <= coarse_starts[r + 1], PMG_ERR_ARG_WRONG, "coarse row blocks must be ascending");
  PMG_CHECK(coarse_starts[0] == 0, PMG_ERR_ARG_WRONG, "coarse row blocks must start at 0");
  h->rb_dist = dist;
  h->rank    = rank;
  h->nranks  = nranks;
  return PMG_SUCCESS;
}

static void *dup_bytes(const void *src, size_t bytes)
{
  void *p = malloc(bytes ? bytes : 1);
  if (p && bytes) memcpy(p, src, bytes);
  return p;
}

pmg_status pmg_mgmc_set_level_rowblock(pmg_mgmc h, int32_t level, int64_t row0, int32_t nowned, int32_t ncolors, const int32_t *colors_owned, const int64_t *send_ptr, const int32_t *send_idx, const int64_t *counts, const int64_t *recv_ptr, const int32_t *recv_src, const int32_t *recv_idx)
{
  PMG_CHECK(h && colors_owned && send_ptr && counts && recv_ptr, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->user_hier && !h->is_setup && h->rb_dist, PMG_ERR_ARG_WRONGSTATE, "call pmg_mgmc_set_rowblock_transport first, before set-up");
  PMG_CHECK(level >= 1 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d (the coarsest level is replicated)", level);
  PMG_CHECK(row0 >= 0 && nowned >= 0 && ncolors >= 1, PMG_ERR_ARG_OUTOFRANGE, "row0 %lld, %d owned rows, %d colours", (long long)row0, nowned, ncolors);
  mg_level    *Lv = &h->lv[level];
  const size_t nc1 = (size_t)ncolors + 1, ns = (size_t)send_ptr[ncolors], nr = (size_t)recv_ptr[ncolors];
  PMG_CHECK((ns == 0 || send_idx) && (nr == 0 || (recv_src && recv_idx)), PMG_ERR_ARG_NULL, "null index list");
  free(Lv->rb_colors), free(Lv->rb_send_ptr), free(Lv->rb_recv_ptr), free(Lv->rb_counts), free(Lv->rb_send_idx), free(Lv->rb_recv_src), free(Lv->rb_recv_idx);
  Lv->rb_colors   = (int32_t *)dup_bytes(colors_owned, sizeof(int32_t) * (size_t)nowned);
  Lv->rb_send_ptr = (int64_t *)dup_bytes(send_ptr, sizeof(int64_t) * nc1);
  Lv->rb_recv_ptr = (int64_t *)dup_bytes(recv_ptr, sizeof(int64_t) * nc1);
  Lv->rb_counts   = (int64_t *)dup_bytes(counts, sizeof(int64_t) * (size_t)ncolors * (size_t)h->nranks);
  Lv->rb_send_idx = (int32_t *)dup_bytes(send_idx, sizeof(int32_t) * ns);
  Lv->rb_recv_src = (int32_t *)dup_bytes(recv_src, sizeof(int32_t) * nr);
  Lv->rb_recv_idx = (int32_t *)dup_bytes(recv_idx, sizeof(int32_t) * nr);
  PMG_CHECK(Lv->rb_colors && Lv->rb_send_ptr && Lv->rb_recv_ptr && Lv->rb_counts && Lv->rb_send_idx && Lv->rb_recv_src && Lv->rb_recv_idx, PMG_ERR_MEM, "out of host memory");
  Lv->rb         = 1;
  Lv->rb_row0    = row0;
  Lv->rb_nowned  = nowned;
  Lv->rb_ncolors = ncolors;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_level_restriction(pmg_mgmc h, int32_t level, int32_t nrows, int32_t ncols, const int32_t *rowptr, const int32_t *colidx, const double *vals)
{
  PMG_CHECK(h && rowptr && colidx && vals, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->user_hier && !h->is_setup, PMG_ERR_ARG_WRONGSTATE, "restrictions belong to pmg_mgmc_create_hierarchy, before set-up");
  PMG_CHECK(level >= 1 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  mg_level *Lv  = &h->lv[level];
  Lv->R_user.nr = nrows;
  Lv->R_user.nc = ncols;
  Lv->R_user.rp = (int32_t *)rowptr; /* borrowed, only read */
  Lv->R_user.ci = (int32_t *)colidx;
  Lv->R_user.v  = (double *)vals;
  return PMG_SUCCESS;
}

/* pmg_mgmc_set_level_operator / _interpolation for either PetscInt width: idx_width = sizeof(PetscInt) * 8 */
pmg_status pmg_mgmc_set_level_operator_idx(pmg_mgmc h, int32_t level, int64_t n, const void *rowptr, const void *colidx, const double *vals, int idx_width)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(level >= 0 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  const int32_t *rp, *ci;
  int32_t       *rpo, *cio;
  PMG_CALL(pmg_narrow_csr(n, n, rowptr, colidx, idx_width, &rp, &ci, &rpo, &cio));
  pmg_status st = pmg_mgmc_set_level_operator(h, level, (int32_t)n, rp, ci, vals);
  mg_level  *Lv = &h->lv[level];
  if (!st) {
    free(Lv->A_rp_own);
    free(Lv->A_ci_own);
    Lv->A_rp_own = rpo;
    Lv->A_ci_own = cio;
  } else {
    free(rpo);
    free(cio);
  }
  return st;
}

pmg_status pmg_mgmc_set_level_interpolation_idx(pmg_mgmc h, int32_t level, int64_t nrows, int64_t ncols, const void *rowptr, const void *colidx, const double *vals, int idx_width)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(level >= 1 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  const int32_t *rp, *ci;
  int32_t       *rpo, *cio;
  PMG_CALL(pmg_narrow_csr(nrows, ncols, rowptr, colidx, idx_width, &rp, &ci, &rpo, &cio));
  pmg_status st = pmg_mgmc_set_level_interpolation(h, level, (int32_t)nrows, (int32_t)ncols, rp, ci, vals);
  mg_level  *Lv = &h->lv[level];
  if (!st) {
    free(Lv->P_rp_own);
    free(Lv->P_ci_own);
    Lv->P_rp_own = rpo;
    Lv->P_ci_own = cio;
  } else {
    free(rpo);
    free(cio);
  }
  return st;
}

pmg_status pmg_mgmc_set_smoother(pmg_mgmc h, int scaled, double omega, int sweep_type, int32_t its)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->is_setup, PMG_ERR_ARG_WRONGSTATE, "set the smoother before pmg_mgmc_setup");
  PMG_CHECK(pmg_sweep_type_ok(sweep_type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported");
  PMG_CHECK(scaled || omega == 1.0, PMG_ERR_SUP, "sorgibbs smoothing requires omega = 1");
  PMG_CHECK(its >= 1 && (uint32_t)its * 4u <= MG_DRAWS_PER_SAMPLE, PMG_ERR_ARG_OUTOFRANGE, "smoothing iterations %d", its);
  h->scaled     = scaled;
  h->omega      = omega;
  h->sweep_type = sweep_type;
  h->nu         = its;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_coarse(pmg_mgmc h, int type, int32_t its)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->is_setup, PMG_ERR_ARG_WRONGSTATE, "set the coarse sampler before pmg_mgmc_setup");
  PMG_CHECK(type == 0 || type == 1, PMG_ERR_ARG_OUTOFRANGE, "coarse sampler type %d", type);
  PMG_CHECK(its >= 1 && (uint32_t)its * 2u <= MG_DRAWS_PER_SAMPLE, PMG_ERR_ARG_OUTOFRANGE, "coarse iterations %d", its);
  h->coarse_type = type;
  h->coarse_its  = its;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_correction_form(pmg_mgmc h, int literal)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  h->correction_form = literal != 0;
  return PMG_SUCCESS;
}

/* on = 0: the cycle forms r = b - A x and b_c = P^T r with two kernels on every level, a low-rank term is subtracted
   from r before the restriction -- the reference's operation order (src/pc_gamgmc.c:194, PCMG's residual then
   MatRestrict).  Default (1): grid levels fuse the two and subtract a low-rank term in restricted form. */
pmg_status pmg_mgmc_set_fused_transfers(pmg_mgmc h, int on)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->is_setup || !h->dist, PMG_ERR_ARG_WRONGSTATE, "z-slab hierarchies decide this at set-up");
  h->no_fused = !on;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_coloring(pmg_mgmc h, int rule)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->is_setup, PMG_ERR_ARG_WRONGSTATE, "the colouring rule must be chosen before set-up");
  PMG_CHECK(rule == PMG_COLORING_GREEDY || rule == PMG_COLORING_ITERATED, PMG_ERR_ARG_OUTOFRANGE, "colouring rule %d: greedy or iterated expected", rule);
  h->aij_coloring = rule;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_set_keep_host(pmg_mgmc h, int keep)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  h->keep_host = keep;
  return PMG_SUCCESS;
}

/* the phase-fused out-of-place sweep and the paired residual (kernels_stencil27_pair.hip) serve class-stencil levels
   that live on one device; PMG_ST27_PAIR=0 keeps the one-launch-per-colour kernels (same bits) */
static int st27_use_pair(const mg_level *Lv)
{
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("PMG_ST27_PAIR");
    env           = e ? atoi(e) : 1;
  }
  return env && Lv->is_st27 && !Lv->distributed && Lv->kz0 == 0 && Lv->nzl == Lv->nz;
}

/* the same kernels on the z-slab of a distributed class-stencil level: one launch per z-parity phase instead of four, the
   halo of the boundary planes behind each as before; PMG_ST27_PAIR_SLAB=0 keeps the per-colour kernels (same bits) */
static int st27_use_pair_slab(const mg_level *Lv)
{
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("PMG_ST27_PAIR_SLAB");
    env           = e ? atoi(e) : 1;
  }
  return env && Lv->is_st27 && Lv->distributed;
}
static int st27_pair_slab_mode(void) { const char *e = getenv("PMG_ST27_PAIR_SLAB"); return e ? atoi(e) : 1; } /* probe: 2 = residual only, 3 = sweeps only */

/* one directional sweep of a class-stencil level on (b, *x): in place, or out of place into Lv->x2 followed by a swap of
   the two buffers when x is the level's own iterate */
static pmg_status st27_one_sweep(mg_level *Lv, const pmgk_st27 *S, int backward, double omega, int noisy, uint64_t seed, uint64_t sweep, const double *b, int x_is_zero, void *stream)
{
  if (Lv->x2 && st27_use_pair(Lv)) {
    /* x_is_zero: the iterate is the zero vector and has NOT been stored (level_iterate_is_unset): the sweep neither reads it
       nor needs the memset */
    PMG_KERNEL(pmgk_st27_sweep_pp(S, backward, omega, noisy, seed, sweep, b, x_is_zero ? NULL : Lv->x, Lv->x2, stream));
    double *t = Lv->x;
    Lv->x     = Lv->x2;
    Lv->x2    = t;
    return PMG_SUCCESS;
  }
  PMG_KERNEL(pmgk_st27_sweep(S, backward, omega, noisy, seed, sweep, b, Lv->x, stream));
  return PMG_SUCCESS;
}

static void level_set_padded(mg_level *Lv)
{
  Lv->padded = 1;
  Lv->off    = (int64_t)Lv->nx * Lv->ny;
  Lv->ld     = (int64_t)Lv->nx * Lv->ny * ((int64_t)Lv->nzl + 2);
}

static pmgk_st27_dims level_dims(const mg_level *Lv)
{
  pmgk_st27_dims d = {Lv->nx, Lv->ny, Lv->nzl, Lv->kz0, Lv->nz};
  return d;
}

/* MATLRC fine-level operator A + B S B^T (MatCreateLRC in examples/ex4.c; PCSetUp_GAMGMC builds the hierarchy from
   the base matrix A, src/pc_gamgmc.c:282-286).  PCGAMGMC_SetUpHierarchy (src/pc_gamgmc.c:157-196) then gives every
   level l the operator A_l + B_l S B_l^T with B_{l-1} = P_l^T B_l, for the level sampler AND the level residual; the
   coarse Cholesky sampler factors the explicit sum (src/pc_chols.c:119-153).  B is n_fine x k column-major in the
   finest level's natural numbering, S the k diagonal entries; both are copied.  Call before pmg_mgmc_setup. */
pmg_status pmg_mgmc_set_lowrank(pmg_mgmc h, int32_t k, const double *B_host, const double *S_host)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->is_setup, PMG_ERR_ARG_WRONGSTATE, "set the low-rank update before pmg_mgmc_setup");
  PMG_CHECK(k >= 0 && k <= 64, PMG_ERR_ARG_OUTOFRANGE, "rank k = %d (0..64 supported)", k);
  free(h->lrc_B);
  free(h->lrc_S);
  h->lrc_B = h->lrc_S = NULL;
  h->lrc_k = 0;
  if (k == 0) return PMG_SUCCESS;
  PMG_CHECK(B_host && S_host, PMG_ERR_ARG_NULL, "null low-rank factor");
  const int32_t n = h->dist ? h->n_io : h->lv[h->nlevels - 1].n; /* z-slabs: the rows of this rank's planes */
  PMG_CHECK(n > 0, PMG_ERR_ARG_WRONGSTATE, "set the finest level operator before the low-rank update");
  h->lrc_B = (double *)malloc(sizeof(double) * (size_t)n * k);
  h->lrc_S = (double *)malloc(sizeof(double) * (size_t)k);
  PMG_CHECK(h->lrc_B && h->lrc_S, PMG_ERR_MEM, "out of host memory");
  memcpy(h->lrc_B, B_host, sizeof(double) * (size_t)n * k);
  memcpy(h->lrc_S, S_host, sizeof(double) * (size_t)k);
  h->lrc_k = k;
  return PMG_SUCCESS;
}

/* Bc = R Bf = P^T Bf column by column (MatTransposeMatMult(Ip, Bf), src/pc_gamgmc.c:177) */
static pmg_status lrc_restrict_B(const hcsr *R, int32_t k, int32_t nf, const double *Bf, double **Bc_out)
{
  double *Bc = (double *)malloc(sizeof(double) * (size_t)R->nr * k);
  PMG_CHECK(Bc, PMG_ERR_MEM, "out of host memory");
  for (int32_t c = 0; c < k; ++c) {
    const double *bf = Bf + (size_t)nf * c;
    double       *bc = Bc + (size_t)R->nr * c;
    for (int32_t r = 0; r < R->nr; ++r) {
      double acc = 0.0;
      for (int32_t q = R->rp[r]; q < R->rp[r + 1]; ++q) acc += R->v[q] * bf[R->ci[q]];
      bc[r] = acc;
    }
  }
  *Bc_out = Bc;
  return PMG_SUCCESS;
}

typedef struct {
  pmg_mgmc  h;
  mg_level *Lv;
} st27_det_ctx;

static pmg_status halo_level(pmg_mgmc h, mg_level *Lv, double *v, void *stream);

static pmg_status st27_det_sweep(void *ctx, int dir, const double *b, double *y, void *stream)
{
  st27_det_ctx *c = (st27_det_ctx *)ctx;
  const int     backward = dir == PMG_SOR_BACKWARD_SWEEP;
  if (c->Lv->distributed) { /* z-slab: the two z-parity phases with a halo of the boundary planes after each */
    PMG_KERNEL(pmgk_st27_sweep_phase(&c->Lv->st, backward, 0, c->h->omega, 0, 0, 0, b, y, stream));
    PMG_CALL(halo_level(c->h, c->Lv, y, stream));
    PMG_KERNEL(pmgk_st27_sweep_phase(&c->Lv->st, backward, 1, c->h->omega, 0, 0, 0, b, y, stream));
    PMG_CALL(halo_level(c->h, c->Lv, y, stream));
    return PMG_SUCCESS;
  }
  PMG_KERNEL(pmgk_st27_sweep(&c->Lv->st, backward, c->h->omega, 0, 0, 0, b, y, stream));
  return PMG_SUCCESS;
}

/* deterministic sweep of the fine level of a z-slab hierarchy; sum of k-vectors over the ranks */
static pmg_status dist_det_sweep(void *ctx, int dir, const double *b, double *y, void *stream)
{
  return pmg_dist_apply_cvec(((pmg_mgmc)ctx)->dist, b, y, dir, stream);
}
static pmg_status mg_reduce(void *ctx, double *vals_dev, int count, void *stream)
{
  return pmg_dist_allreduce_sum(((pmg_mgmc)ctx)->dist, vals_dev, count, stream);
}

/* MatCreateLRC(Ac, Bc, Sf) + KSPSetOperators on the level sampler (src/pc_gamgmc.c:178, :185-187): B_nat is the
   level's n x k block in natural numbering */
static pmg_status level_attach_lrc(pmg_mgmc h, mg_level *Lv, const double *B_nat)
{
  if (Lv->is_grid) return pmg_grid_set_lowrank(Lv->g, h->lrc_k, B_nat, h->lrc_S);
  if (Lv->mc) return pmg_mcsor_set_lowrank(Lv->mc, h->lrc_k, B_nat, h->lrc_S);
  if (Lv->is_st27) {
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)Lv->n);
    PMG_CHECK(pos, PMG_ERR_MEM, "out of host memory");
    for (int32_t q = 0; q < Lv->n; ++q) pos[q] = q + Lv->off;
    st27_det_ctx ctx = {h, Lv};
    pmg_status   st  = pmg_lrc_build(&Lv->lrc, h->lrc_k, Lv->ld, Lv->n, B_nat, pos, h->lrc_S, st27_det_sweep, &ctx);
    free(pos);
    return st;
  }
  return PMG_SUCCESS; /* coarsest level with the Cholesky sampler: the update goes into the factored matrix */
}

static pmg_status upload_transfer(const hcsr *M, const int32_t *rowpos_of, const int32_t *colpos_of, int32_t **rowpos, int32_t **rowptr, int32_t **col, double **val)
{
  const int32_t nnz = M->rp[M->nr];
  int32_t      *rp  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M->nr > 0 ? M->nr : 1));
  int32_t      *cc  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  PMG_CHECK(rp && cc, PMG_ERR_MEM, "out of host memory");
  for (int32_t r = 0; r < M->nr; ++r) rp[r] = rowpos_of[r];
  for (int32_t k = 0; k < nnz; ++k) cc[k] = colpos_of[M->ci[k]];
  pmg_status st = pmg_dev_upload((void **)rowpos, rp, sizeof(int32_t) * (size_t)M->nr);
  if (!st) st = pmg_dev_upload((void **)rowptr, M->rp, sizeof(int32_t) * ((size_t)M->nr + 1));
  if (!st) st = pmg_dev_upload((void **)col, cc, sizeof(int32_t) * (size_t)nnz);
  if (!st) st = pmg_dev_upload((void **)val, M->v, sizeof(double) * (size_t)nnz);
  free(rp);
  free(cc);
  return st;
}

static pmg_status upload_transfer(const hcsr *M, const int32_t *rowpos_of, const int32_t *colpos_of, int32_t **rowpos, int32_t **rowptr, int32_t **col, double **val);
static int        st27_from_csr(mg_level *Lv, const hcsr *A, double omega, pmg_status *st);
static pmg_status stencil_tables_from_proxy(pmg_mgmc h, st27_table *tab, int *ok);
static pmg_status mgmc_setup_stencil(pmg_mgmc h, const st27_table *tab);

/* a vector of the highest replicated level of a row-block hierarchy, of which every rank has computed the rows of its block
   (coarse row blocks rb_c0_starts): all ranks end up with all rows.  The coarsest level keeps natural order (one all-gather
   in place); a sliced-ELL level goes through natural order: gather my block, all-gather, scatter everything. */
static pmg_status rb_fold_allgather(pmg_mgmc h, double *v, void *stream)
{
  int64_t cnts[64];
  for (int r = 0; r < h->nranks; ++r) cnts[r] = h->rb_c0_starts[r + 1] - h->rb_c0_starts[r];
  if (!h->rb_fold_pos) return pmg_dist_allgather(h->rb_dist, v, h->rb_c0_starts, cnts, stream);
  const int64_t s0 = h->rb_c0_starts[h->rank], n = h->rb_c0_starts[h->nranks];
  PMG_KERNEL(pmgk_gather_idx(cnts[h->rank], h->rb_fold_pos + s0, v, h->rb_fold_buf + s0, stream));
  PMG_CALL(pmg_dist_allgather(h->rb_dist, h->rb_fold_buf, h->rb_c0_starts, cnts, stream));
  PMG_KERNEL(pmgk_scatter_idx(n, h->rb_fold_iota, h->rb_fold_pos, h->rb_fold_buf, v, stream));
  return PMG_SUCCESS;
}

static pmg_status mgmc_setup_user(pmg_mgmc h)
{
  const int top = h->nlevels - 1;
  int32_t **pos = (int32_t **)calloc((size_t)h->nlevels, sizeof(int32_t *));
  PMG_CHECK(pos, PMG_ERR_MEM, "out of host memory");
  for (int l = 0; l <= top; ++l) {
    mg_level *Lv = &h->lv[l];
    PMG_CHECK(Lv->A_user.rp, PMG_ERR_ARG_WRONGSTATE, "level %d has no operator", l);
    if (h->rb_dist) { /* row blocks from level rb_fold upwards; the levels below are replicated (the coarsest sampled exactly) */
      if (l == 0) {
        h->rb_fold = 1;
        while (h->rb_fold <= top && !h->lv[h->rb_fold].rb) ++h->rb_fold;
        PMG_CHECK(h->rb_fold <= top, PMG_ERR_ARG_WRONGSTATE, "row-block hierarchy without a row-block level (pmg_mgmc_set_level_rowblock)");
        PMG_CHECK(h->rb_c0_starts[h->nranks] == h->lv[h->rb_fold - 1].n, PMG_ERR_ARG_SIZ, "the row blocks of the highest replicated level cover %lld rows, level %d has %d", (long long)h->rb_c0_starts[h->nranks], h->rb_fold - 1, h->lv[h->rb_fold - 1].n);
      }
      PMG_CHECK(l < h->rb_fold ? !Lv->rb : Lv->rb, PMG_ERR_ARG_WRONGSTATE, "row-block hierarchy: level %d %s", l, l < h->rb_fold ? "lies below a replicated level and must be replicated too" : "has no row block (pmg_mgmc_set_level_rowblock)");
      PMG_CHECK(h->coarse_type == 0, PMG_ERR_SUP, "row-block hierarchies: exact coarse sampler");
      if (l >= h->rb_fold) {
        PMG_CHECK(Lv->P_user.rp && Lv->R_user.rp && Lv->P_user.nr == Lv->rb_nowned && Lv->P_user.nc == h->lv[l - 1].n && Lv->R_user.nc == Lv->n, PMG_ERR_ARG_SIZ, "level %d: interpolation rows = owned rows, its columns and the restriction's in local numbering", l);
        PMG_CHECK(Lv->R_user.nr == (l == h->rb_fold ? (int32_t)(h->rb_c0_starts[h->rank + 1] - h->rb_c0_starts[h->rank]) : h->lv[l - 1].rb_nowned), PMG_ERR_ARG_SIZ, "level %d: the restriction has one row per owned row of level %d", l, l - 1);
      } else
        PMG_CHECK(l == 0 || (Lv->P_user.rp && Lv->P_user.nr == Lv->n && Lv->P_user.nc == h->lv[l - 1].n), PMG_ERR_ARG_SIZ, "interpolation of the replicated level %d missing or of the wrong shape", l);
    } else
      PMG_CHECK(l == 0 || (Lv->P_user.rp && Lv->P_user.nr == Lv->n && Lv->P_user.nc == h->lv[l - 1].n), PMG_ERR_ARG_SIZ, "interpolation of level %d missing or of the wrong shape", l);
    pos[l] = (int32_t *)malloc(sizeof(int32_t) * (size_t)Lv->n);
    PMG_CHECK(pos[l], PMG_ERR_MEM, "out of host memory");
    if (l > 0 || h->coarse_type == 1) {
      PMG_CALL(pmg_mcsor_create_csr(Lv->n, Lv->A_user.rp, Lv->A_user.ci, Lv->A_user.v, &Lv->mc));
      PMG_CALL(pmg_mcsor_set_natural_order(Lv->mc, 1));
      Lv->A_nnz = Lv->A_user.rp[Lv->n];
      if (Lv->rb) { /* the caller's global colouring on the owned rows, the ghost rows in a colour of their own that is never swept */
        PMG_CHECK(Lv->rb_nowned <= Lv->n, PMG_ERR_ARG_SIZ, "level %d: %d owned rows of %d local rows", l, Lv->rb_nowned, Lv->n);
        int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)Lv->n);
        PMG_CHECK(col, PMG_ERR_MEM, "out of host memory");
        for (int32_t r = 0; r < Lv->n; ++r) col[r] = r < Lv->rb_nowned ? Lv->rb_colors[r] : Lv->rb_ncolors;
        pmg_status st = pmg_mcsor_set_coloring(Lv->mc, PMG_COLORING_USER, col);
        free(col);
        PMG_CALL(st);
        PMG_CALL(pmg_mcsor_set_noise_row_offset(Lv->mc, Lv->rb_row0));
      } else PMG_CALL(pmg_mcsor_set_coloring(Lv->mc, h->aij_coloring, NULL)); /* PMG_COLORING_GREEDY unless pmg_mgmc_set_coloring said otherwise */
      PMG_CALL(pmg_mcsor_set_omega(Lv->mc, h->omega));
      PMG_CALL(pmg_mcsor_set_sweep_type(Lv->mc, h->sweep_type));
      PMG_CALL(pmg_mcsor_setup(Lv->mc));
      int32_t ld32;
      PMG_CALL(pmg_mcsor_layout_len(Lv->mc, &ld32));
      Lv->ld = ld32;
      PMG_CALL(pmg_mcsor_get_layout(Lv->mc, pos[l]));
      if (Lv->rb) { /* the ghost-update plan in layout positions */
        const int64_t ns = Lv->rb_send_ptr[Lv->rb_ncolors], nr = Lv->rb_recv_ptr[Lv->rb_ncolors];
        int32_t      *sp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ns + 1)), *rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr + 1));
        pmg_status    st = (sp && rp) ? PMG_SUCCESS : pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
        for (int64_t q = 0; q < ns && !st; ++q) {
          if (Lv->rb_send_idx[q] < 0 || Lv->rb_send_idx[q] >= Lv->rb_nowned) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "level %d: send row %d is not an owned row", l, Lv->rb_send_idx[q]);
          else sp[q] = pos[l][Lv->rb_send_idx[q]];
        }
        for (int64_t q = 0; q < nr && !st; ++q) {
          if (Lv->rb_recv_idx[q] < Lv->rb_nowned || Lv->rb_recv_idx[q] >= Lv->n) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "level %d: receive row %d is not a ghost row", l, Lv->rb_recv_idx[q]);
          else rp[q] = pos[l][Lv->rb_recv_idx[q]];
        }
        if (!st) st = pmg_distmcsor_create(Lv->mc, h->rb_dist, Lv->rb_ncolors, Lv->rb_send_ptr, sp, Lv->rb_counts, Lv->rb_recv_ptr, Lv->rb_recv_src, rp, &Lv->dm);
        free(sp);
        free(rp);
        PMG_CALL(st);
      }
    } else {
      Lv->ld = Lv->n;
      for (int32_t q = 0; q < Lv->n; ++q) pos[l][q] = q;
    }
    if (l == 0 && h->coarse_type == 0 && !h->lrc_k) PMG_CALL(pmg_chol_create_csr(Lv->n, Lv->A_user.rp, Lv->A_user.ci, Lv->A_user.v, &h->chol));
  }
  double *Bcur = h->lrc_B; /* level-l block of the low-rank factor, natural numbering (owned by h at the top) */
  double *Bdev = NULL;     /* row blocks: the level-l block in the level's LAYOUT on the device, zero on the ghost rows */
  if (h->lrc_k && h->rb_dist) {
    mg_level *T  = &h->lv[top];
    double   *Bl = (double *)calloc((size_t)T->ld * (size_t)h->lrc_k, sizeof(double));
    PMG_CHECK(Bl, PMG_ERR_MEM, "out of host memory");
    for (int32_t c = 0; c < h->lrc_k; ++c)
      for (int32_t r = 0; r < T->rb_nowned; ++r) Bl[(size_t)T->ld * c + pos[top][r]] = Bcur[(size_t)T->n * c + r];
    pmg_status st = pmg_dev_upload((void **)&Bdev, Bl, sizeof(double) * (size_t)T->ld * (size_t)h->lrc_k);
    free(Bl);
    PMG_CALL(st);
    PMG_CALL(pmg_distmcsor_set_lowrank_dev(T->dm, h->lrc_k, Bdev, h->lrc_S));
  } else if (h->lrc_k) PMG_CALL(level_attach_lrc(h, &h->lv[top], Bcur));
  for (int l = top; l >= 1; --l) {
    mg_level *U = &h->lv[l];
    hcsr      R;
    memset(&R, 0, sizeof R);
    if (!U->rb) PMG_CALL(hcsr_transpose(&U->P_user, &R));
    if (h->lrc_k && !U->rb) { /* B_{l-1} = P_l^T B_l, src/pc_gamgmc.c:177-178 */
      mg_level *Cc = &h->lv[l - 1];
      double   *Bc = NULL;
      PMG_CALL(lrc_restrict_B(&R, h->lrc_k, U->n, Bcur, &Bc));
      if (Bcur != h->lrc_B) free(Bcur);
      Bcur = Bc;
      PMG_CALL(level_attach_lrc(h, Cc, Bcur));
      if (l - 1 == 0 && h->coarse_type == 0) PMG_CALL(pmg_chol_create_csr_lowrank(Cc->n, Cc->A_user.rp, Cc->A_user.ci, Cc->A_user.v, h->lrc_k, Bcur, h->lrc_S, &h->chol));
    }
    U->P_nrows = U->P_user.nr;
    U->R_nrows = U->rb ? U->R_user.nr : R.nr;
    PMG_CALL(upload_transfer(&U->P_user, pos[l], pos[l - 1], &U->P_rowpos, &U->P_rowptr, &U->P_col, &U->P_val));
    U->P_nnz = U->P_user.rp[U->P_user.nr];
    if (!U->rb) PMG_CALL(upload_transfer(&R, pos[l - 1], pos[l], &U->R_rowpos, &U->R_rowptr, &U->R_col, &U->R_val));
    else /* the caller's rows of P^T: owned rows of level l-1 (on the replicated coarsest level: this rank's block of the global rows) */
      PMG_CALL(upload_transfer(&U->R_user, pos[l - 1] + (l == h->rb_fold ? h->rb_c0_starts[h->rank] : 0), pos[l], &U->R_rowpos, &U->R_rowptr, &U->R_col, &U->R_val));
    if (U->rb && l == h->rb_fold && l - 1 >= 1) { /* the replicated level below keeps its vectors in a sliced-ELL layout: all-gather through natural order */
      mg_level *Cc   = &h->lv[l - 1];
      int32_t  *iota = (int32_t *)malloc(sizeof(int32_t) * (size_t)Cc->n);
      PMG_CHECK(iota, PMG_ERR_MEM, "out of host memory");
      for (int32_t q = 0; q < Cc->n; ++q) iota[q] = q;
      pmg_status st = pmg_dev_upload((void **)&h->rb_fold_pos, pos[l - 1], sizeof(int32_t) * (size_t)Cc->n);
      if (!st) st = pmg_dev_upload((void **)&h->rb_fold_iota, iota, sizeof(int32_t) * (size_t)Cc->n);
      if (!st) st = pmg_dev_alloc((void **)&h->rb_fold_buf, sizeof(double) * (size_t)Cc->n);
      free(iota);
      PMG_CALL(st);
    }
    hcsr_free(&R);
    if (h->lrc_k && U->rb) { /* row blocks: B_{l-1} = P_l^T B_l column by column on the device, with the V-cycle's own restriction */
      mg_level *Cc = &h->lv[l - 1];
      double   *Bc = NULL;
      PMG_CALL(pmg_dev_alloc((void **)&Bc, sizeof(double) * (size_t)Cc->ld * (size_t)h->lrc_k));
      PMG_HIP(hipMemset(Bc, 0, sizeof(double) * (size_t)Cc->ld * (size_t)h->lrc_k));
      for (int32_t c = 0; c < h->lrc_k; ++c) {
        double *bf = Bdev + (size_t)U->ld * c, *bc = Bc + (size_t)Cc->ld * c;
        PMG_CALL(pmg_distmcsor_refresh_layout(U->dm, bf, NULL)); /* the rows of P^T read other ranks' rows of B */
        PMG_KERNEL(pmgk_csr_spmv_rows(U->R_nrows, U->R_rowpos, U->R_rowptr, U->R_col, U->R_val, bf, bc, 0, NULL, NULL));
        if (l == h->rb_fold) PMG_CALL(rb_fold_allgather(h, bc, NULL)); /* the replicated level below takes the whole column */
      }
      /* the refresh left copies on the ghost rows of the fine block: back to zeros there (B counts every row once) -- not
         needed any more, the block is dropped */
      pmg_dev_free(Bdev);
      Bdev = Bc;
      if (l > h->rb_fold) PMG_CALL(pmg_distmcsor_set_lowrank_dev(Cc->dm, h->lrc_k, Bdev, h->lrc_S));
      else { /* the highest replicated level: its whole block goes to the host in natural numbering, where the replicated levels below take over */
        double *Bl = (double *)malloc(sizeof(double) * (size_t)Cc->ld * (size_t)h->lrc_k), *B0 = (double *)malloc(sizeof(double) * (size_t)Cc->n * (size_t)h->lrc_k);
        PMG_CHECK(Bl && B0, PMG_ERR_MEM, "out of host memory");
        pmg_status st = hipMemcpy(Bl, Bdev, sizeof(double) * (size_t)Cc->ld * (size_t)h->lrc_k, hipMemcpyDeviceToHost) == hipSuccess ? PMG_SUCCESS : pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "download failed");
        for (int32_t c = 0; c < h->lrc_k && !st; ++c)
          for (int32_t r = 0; r < Cc->n; ++r) B0[(size_t)Cc->n * c + r] = Bl[(size_t)Cc->ld * c + pos[l - 1][r]];
        free(Bl);
        if (Bcur != h->lrc_B) free(Bcur);
        Bcur = B0;
        if (!st && l - 1 == 0) st = pmg_chol_create_csr_lowrank(Cc->n, Cc->A_user.rp, Cc->A_user.ci, Cc->A_user.v, h->lrc_k, Bcur, h->lrc_S, &h->chol);
        else if (!st) st = level_attach_lrc(h, Cc, Bcur);
        PMG_CALL(st);
      }
    }
  }
  if (Bcur != h->lrc_B) free(Bcur);
  pmg_dev_free(Bdev);
  for (int l = 0; l <= top; ++l) {
    free(pos[l]);
    mg_level *Lv = &h->lv[l];
    memset(&Lv->A_user, 0, sizeof Lv->A_user); /* borrowed arrays are released */
    memset(&Lv->P_user, 0, sizeof Lv->P_user);
    memset(&Lv->R_user, 0, sizeof Lv->R_user);
    free(Lv->A_rp_own);
    free(Lv->A_ci_own);
    free(Lv->P_rp_own);
    free(Lv->P_ci_own);
    Lv->A_rp_own = Lv->A_ci_own = Lv->P_rp_own = Lv->P_ci_own = NULL;
    PMG_CALL(pmg_dev_alloc((void **)&Lv->b, sizeof(double) * (size_t)Lv->ld));
    PMG_CALL(pmg_dev_alloc((void **)&Lv->x, sizeof(double) * (size_t)Lv->ld));
    PMG_CALL(pmg_dev_alloc((void **)&Lv->r, sizeof(double) * (size_t)Lv->ld));
  }
  free(pos);
  PMG_CALL(pmg_dev_alloc((void **)&h->y_lay, sizeof(double) * (size_t)h->lv[top].ld));
  PMG_CALL(pmg_dev_alloc((void **)&h->b_lay, sizeof(double) * (size_t)h->lv[top].ld));
  h->n_io     = h->lv[top].n; /* the caller's vectors: one entry per (local) row of the finest level */
  h->is_setup = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_setup(pmg_mgmc h)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  if (h->is_setup) return PMG_SUCCESS;
  if (h->user_hier) return mgmc_setup_user(h);
  const int top = h->nlevels - 1;
  if (h->dist) PMG_CHECK(!h->keep_host, PMG_ERR_SUP, "host copies of the level matrices are a single-device feature");
  if (h->dist || (!h->keep_host && !getenv("PMG_MG_FULL_GALERKIN") && !getenv("PMG_MG_NO_STENCIL") && !getenv("PMG_MG_CSR_TRANSFERS"))) {
    /* class-stencil tables from the proxy hierarchy: no product with the full-size matrices */
    st27_table *tab = (st27_table *)malloc(sizeof(st27_table) * (size_t)top);
    PMG_CHECK(tab, PMG_ERR_MEM, "out of host memory");
    int        ok = 0;
    pmg_status st = stencil_tables_from_proxy(h, tab, &ok);
    if (!st && ok) st = mgmc_setup_stencil(h, tab);
    free(tab);
    if (st || ok) return st;
    PMG_CHECK(!h->dist, PMG_ERR_SUP, "the coarse operators of this grid are not class stencils; z-slabs need them");
  }
  /* finest level: matrix-free grid operator */
  mg_level *F = &h->lv[top];
  F->is_grid  = 1;
  PMG_CALL(pmg_grid_create(F->nx, F->ny, F->nz, 0, F->nz, h->kappa, &F->g));
  h->own_grid = 1;
  PMG_CALL(pmg_grid_set_omega(F->g, h->omega));
  PMG_CALL(pmg_grid_set_sweep_type(F->g, h->sweep_type));
  PMG_CALL(pmg_grid_cvec_len(F->g, &F->ld));
  PMG_CHECK(F->ld < 2147483647, PMG_ERR_ARG_OUTOFRANGE, "fine level layout exceeds 32-bit positions");

  int32_t **pos = (int32_t **)calloc((size_t)h->nlevels, sizeof(int32_t *)); /* natural index -> layout position per level */
  PMG_CHECK(pos, PMG_ERR_MEM, "out of host memory");
  {
    int64_t *p64 = (int64_t *)malloc(sizeof(int64_t) * (size_t)F->n);
    pos[top]     = (int32_t *)malloc(sizeof(int32_t) * (size_t)F->n);
    PMG_CHECK(p64 && pos[top], PMG_ERR_MEM, "out of host memory");
    PMG_CALL(pmg_grid_get_layout(F->g, p64));
    for (int32_t q = 0; q < F->n; ++q) pos[top][q] = (int32_t)p64[q];
    free(p64);
  }
  rowsrc src;
  memset(&src, 0, sizeof src);
  src.nx    = F->nx;
  src.ny    = F->ny;
  src.nz    = F->nz;
  src.kappa = h->kappa;
  src.h2    = 1. / ((F->nx - 1) * (F->nx - 1)); /* src/problems.c:24 */
  for (int nn = 0; nn < 8; ++nn) {
    double dgl = h->kappa * h->kappa;
    for (int q = 0; q < nn; ++q) dgl += src.h2;
    src.diag[nn] = dgl;
  }
  hcsr Aprev;
  memset(&Aprev, 0, sizeof Aprev);
  double *Bcur = h->lrc_B; /* level-l block of the low-rank factor, natural numbering (owned by h at the top) */
  if (h->lrc_k) PMG_CALL(level_attach_lrc(h, F, Bcur));
  for (int l = top; l >= 1; --l) {
    mg_level     *U = &h->lv[l], *Cc = &h->lv[l - 1];
    const int32_t nf[3] = {U->nx, U->ny, U->nz}, ncd[3] = {Cc->nx, Cc->ny, Cc->nz};
    hcsr          P, R, Ac;
    memset(&P, 0, sizeof P);
    memset(&R, 0, sizeof R);
    memset(&Ac, 0, sizeof Ac);
    PMG_CALL(q1_interp(nf, ncd, &P));
    PMG_CALL(hcsr_transpose(&P, &R));
    rowsrc s = src;
    if (l < top) s.A = &Aprev;
    PMG_CALL(galerkin_rap(&s, l == top ? 7 : 64, &P, &R, &Ac));
    /* coarse level operator object */
    const int is_coarsest = (l - 1 == 0);
    pos[l - 1]            = (int32_t *)malloc(sizeof(int32_t) * (size_t)Cc->n);
    PMG_CHECK(pos[l - 1], PMG_ERR_MEM, "out of host memory");
    pmg_status st27_status = PMG_SUCCESS;
    if ((!is_coarsest || h->coarse_type == 1) && !getenv("PMG_MG_NO_STENCIL") && st27_from_csr(Cc, &Ac, h->omega, &st27_status)) {
      level_set_padded(Cc);
      for (int32_t q = 0; q < Cc->n; ++q) pos[l - 1][q] = q + (int32_t)Cc->off;
    } else if (!is_coarsest || h->coarse_type == 1) {
      PMG_CALL(st27_status);
      int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)Cc->n);
      PMG_CHECK(col, PMG_ERR_MEM, "out of host memory");
      /* parity colouring (i&1) + 2(j&1) + 4(k&1), compressed to consecutive colours: valid for the 9/27-point box */
      int present[8] = {0}, remap[8], ncol = 0;
      for (int32_t k = 0; k < Cc->nz; ++k)
        for (int32_t j = 0; j < Cc->ny; ++j)
          for (int32_t i = 0; i < Cc->nx; ++i) present[(i & 1) + 2 * (j & 1) + 4 * (k & 1)] = 1;
      for (int q = 0; q < 8; ++q) remap[q] = present[q] ? ncol++ : -1;
      for (int32_t k = 0; k < Cc->nz; ++k)
        for (int32_t j = 0; j < Cc->ny; ++j)
          for (int32_t i = 0; i < Cc->nx; ++i) col[i + Cc->nx * (j + Cc->ny * k)] = remap[(i & 1) + 2 * (j & 1) + 4 * (k & 1)];
      PMG_CALL(pmg_mcsor_create_csr(Cc->n, Ac.rp, Ac.ci, Ac.v, &Cc->mc));
      PMG_CALL(pmg_mcsor_set_natural_order(Cc->mc, 1));
      Cc->A_nnz = Ac.rp[Cc->n];
      PMG_CALL(pmg_mcsor_set_coloring(Cc->mc, PMG_COLORING_USER, col));
      PMG_CALL(pmg_mcsor_set_omega(Cc->mc, h->omega));
      PMG_CALL(pmg_mcsor_set_sweep_type(Cc->mc, h->sweep_type));
      PMG_CALL(pmg_mcsor_setup(Cc->mc));
      free(col);
      int32_t ld32;
      PMG_CALL(pmg_mcsor_layout_len(Cc->mc, &ld32));
      Cc->ld = ld32;
      PMG_CALL(pmg_mcsor_get_layout(Cc->mc, pos[l - 1]));
    } else {
      level_set_padded(Cc);
      for (int32_t q = 0; q < Cc->n; ++q) pos[l - 1][q] = q + (int32_t)Cc->off;
    }
    if (h->lrc_k) { /* B_{l-1} = P_l^T B_l and the MATLRC level operator, src/pc_gamgmc.c:177-187 */
      double *Bc = NULL;
      PMG_CALL(lrc_restrict_B(&R, h->lrc_k, U->n, Bcur, &Bc));
      if (Bcur != h->lrc_B) free(Bcur);
      Bcur = Bc;
      PMG_CALL(level_attach_lrc(h, Cc, Bcur));
    }
    if (is_coarsest && h->coarse_type == 0) PMG_CALL(pmg_chol_create_csr_lowrank(Cc->n, Ac.rp, Ac.ci, Ac.v, h->lrc_k, Bcur, h->lrc_S, &h->chol));
    /* transfers: matrix-free Q1 kernels from the grid level and between natural-order levels, CSR products in
       layout numbering otherwise */
    if (U->is_grid && !getenv("PMG_MG_CSR_TRANSFERS")) {
      U->grid_transfer = 1;
      if (!Cc->padded) PMG_CALL(pmg_dev_upload((void **)&U->cpos_dev, pos[l - 1], sizeof(int32_t) * (size_t)Cc->n));
    } else if (U->is_st27 && Cc->padded && !getenv("PMG_MG_CSR_TRANSFERS")) {
      U->nat_transfer = 1;
    } else {
      U->P_nrows = P.nr;
      U->R_nrows = R.nr;
      PMG_CALL(upload_transfer(&P, pos[l], pos[l - 1], &U->P_rowpos, &U->P_rowptr, &U->P_col, &U->P_val));
      U->P_nnz = P.rp[P.nr];
      PMG_CALL(upload_transfer(&R, pos[l - 1], pos[l], &U->R_rowpos, &U->R_rowptr, &U->R_col, &U->R_val));
    }
    hcsr_free(&R);
    if (h->keep_host) U->P_host = P;
    else hcsr_free(&P);
    hcsr_free(&Aprev);
    Aprev = Ac;
    if (h->keep_host) { /* deep copy for inspection */
      const int32_t nnz = Ac.rp[Ac.nr];
      Cc->A_host.nr = Cc->A_host.nc = Ac.nr;
      Cc->A_host.rp = (int32_t *)malloc(sizeof(int32_t) * ((size_t)Ac.nr + 1));
      Cc->A_host.ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)nnz);
      Cc->A_host.v  = (double *)malloc(sizeof(double) * (size_t)nnz);
      PMG_CHECK(Cc->A_host.rp && Cc->A_host.ci && Cc->A_host.v, PMG_ERR_MEM, "out of host memory");
      memcpy(Cc->A_host.rp, Ac.rp, sizeof(int32_t) * ((size_t)Ac.nr + 1));
      memcpy(Cc->A_host.ci, Ac.ci, sizeof(int32_t) * (size_t)nnz);
      memcpy(Cc->A_host.v, Ac.v, sizeof(double) * (size_t)nnz);
    }
  }
  hcsr_free(&Aprev);
  if (Bcur != h->lrc_B) free(Bcur);
  for (int l = 0; l < h->nlevels; ++l) {
    free(pos[l]);
    mg_level *Lv = &h->lv[l];
    PMG_CALL(pmg_dev_alloc((void **)&Lv->b, sizeof(double) * (size_t)Lv->ld));
    PMG_CALL(pmg_dev_alloc((void **)&Lv->x, sizeof(double) * (size_t)Lv->ld));
    PMG_CALL(pmg_dev_alloc((void **)&Lv->r, sizeof(double) * (size_t)Lv->ld));
    if (st27_use_pair(Lv) || st27_use_pair_slab(Lv)) PMG_CALL(pmg_dev_alloc((void **)&Lv->x2, sizeof(double) * (size_t)Lv->ld));
  }
  free(pos);
  PMG_CALL(pmg_dev_alloc((void **)&h->y_lay, sizeof(double) * (size_t)F->ld));
  PMG_CALL(pmg_dev_alloc((void **)&h->b_lay, sizeof(double) * (size_t)F->ld));
  h->is_setup = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_get_num_levels(pmg_mgmc h, int32_t *levels)
{
  PMG_CHECK(h && levels, PMG_ERR_ARG_NULL, "null argument");
  *levels = h->nlevels;
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_get_level_dims(pmg_mgmc h, int32_t level, int32_t *nx, int32_t *ny, int32_t *nz)
{
  PMG_CHECK(h && nx && ny && nz, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(level >= 0 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  *nx = h->lv[level].nx;
  *ny = h->lv[level].ny;
  *nz = h->lv[level].nz;
  return PMG_SUCCESS;
}

/* which = 0: Galerkin operator of `level` (< finest); which = 1: interpolation from level-1 to `level` (>= 1).
   Call with NULL arrays to query nrows/nnz.  Needs pmg_mgmc_set_keep_host(h, 1) before set-up. */
pmg_status pmg_mgmc_get_level_matrix(pmg_mgmc h, int32_t level, int which, int32_t *nrows, int32_t *nnz, int32_t *rowptr, int32_t *colidx, double *vals)
{
  PMG_CHECK(h && nrows && nnz, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->is_setup && h->keep_host, PMG_ERR_ARG_WRONGSTATE, "needs pmg_mgmc_set_keep_host(h,1) and pmg_mgmc_setup");
  PMG_CHECK(level >= 0 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  const hcsr *M = which == 0 ? &h->lv[level].A_host : &h->lv[level].P_host;
  PMG_CHECK(M->rp, PMG_ERR_ARG_OUTOFRANGE, "level %d has no such matrix", level);
  *nrows = M->nr;
  *nnz   = M->rp[M->nr];
  if (rowptr) memcpy(rowptr, M->rp, sizeof(int32_t) * ((size_t)M->nr + 1));
  if (colidx) memcpy(colidx, M->ci, sizeof(int32_t) * (size_t)*nnz);
  if (vals) memcpy(vals, M->v, sizeof(double) * (size_t)*nnz);
  return PMG_SUCCESS;
}

/* Try to express the CSR operator of a structured level as 27 position-class stencils; returns 1 if every row equals
   its class stencil bit for bit (always the case for Galerkin operators of the constant-coefficient fine operator),
   0 otherwise (the caller keeps the sliced-ELL form). */
/* class-stencil table of a structured 27-point (9-point) matrix on an nx*ny*nz grid: coef[27*cls + e] and which
   classes occur; returns 0 when the matrix is not of that form (a row is not the full in-domain 27-box, or two points
   of one position class have different rows) */
static int st27_extract(int nx, int ny, int nz, const hcsr *A, double *coef /* [27*27] */, int *have /* [27] */)
{
  memset(coef, 0, sizeof(double) * 27 * 27);
  memset(have, 0, sizeof(int) * 27);
  for (int32_t k = 0; k < nz; ++k)
    for (int32_t j = 0; j < ny; ++j)
      for (int32_t i = 0; i < nx; ++i) {
        const int32_t row = i + nx * (j + ny * k);
        const int     cls = (i == 0 ? 0 : (i == nx - 1 ? 2 : 1)) + 3 * (j == 0 ? 0 : (j == ny - 1 ? 2 : 1)) + 9 * (k == 0 ? 0 : (k == nz - 1 ? 2 : 1));
        double        loc[27];
        memset(loc, 0, sizeof loc);
        int32_t expect = 0;
        for (int dz = -1; dz <= 1; ++dz)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
              if (i + dx >= 0 && i + dx < nx && j + dy >= 0 && j + dy < ny && k + dz >= 0 && k + dz < nz) ++expect;
        if (A->rp[row + 1] - A->rp[row] != expect) return 0; /* not the full in-domain 27-box */
        for (int32_t q = A->rp[row]; q < A->rp[row + 1]; ++q) {
          const int32_t c = A->ci[q], ci = c % nx, cj = (c / nx) % ny, ck = c / (nx * ny);
          const int     dx = ci - i, dy = cj - j, dz = ck - k;
          if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) return 0;
          loc[9 * (dz + 1) + 3 * (dy + 1) + (dx + 1)] = A->v[q];
        }
        if (!have[cls]) {
          memcpy(coef + 27 * cls, loc, sizeof loc);
          have[cls] = 1;
        } else if (memcmp(coef + 27 * cls, loc, sizeof loc) != 0) {
          return 0;
        }
      }
  return 1;
}

/* upload a class-stencil table and make Lv a class-stencil level (owned planes kz0 .. kz0+nzl-1 of its nz) */
static pmg_status st27_install(mg_level *Lv, const double *coef, const int *have, double omega)
{
  double       dg[27], idg[27], sq[27], sqs[27];
  const double sc = sqrt((2 - omega) / omega);
  for (int c = 0; c < 27; ++c) {
    dg[c] = have[c] ? coef[27 * c + 13] : 1.0;
    const double t = 1.0 / dg[c];
    idg[c]         = t * omega;            /* MCSORUpdateIDiag, src/mc_sor.c:114-124 */
    sq[c]          = sqrt(fabs(dg[c]));    /* src/pc_mcgibbs.c:149 */
    sqs[c]         = sq[c] * sc;
  }
  PMG_CALL(pmg_dev_upload((void **)&Lv->st_coef, coef, sizeof(double) * 27 * 27));
  PMG_CALL(pmg_dev_upload((void **)&Lv->st_idiag, idg, sizeof idg));
  PMG_CALL(pmg_dev_upload((void **)&Lv->st_sqrtd, sq, sizeof sq));
  PMG_CALL(pmg_dev_upload((void **)&Lv->st_sqrtd_scaled, sqs, sizeof sqs));
  Lv->st.nx    = Lv->nx;
  Lv->st.ny    = Lv->ny;
  Lv->st.nz    = Lv->nzl;
  Lv->st.kz0   = Lv->kz0;
  Lv->st.nzg   = Lv->nz;
  Lv->st.coef  = Lv->st_coef;
  Lv->st.idiag = Lv->st_idiag;
  Lv->is_st27  = 1;
  return PMG_SUCCESS;
}

static int st27_from_csr(mg_level *Lv, const hcsr *A, double omega, pmg_status *st)
{
  double coef[27 * 27];
  int    have[27];
  *st = PMG_SUCCESS;
  if (!st27_extract(Lv->nx, Lv->ny, Lv->nz, A, coef, have)) return 0;
  *st = st27_install(Lv, coef, have, omega);
  return *st == PMG_SUCCESS;
}

/* ---- hierarchy from class-stencil tables ----------------------------------------------------------------------
   The Galerkin operators of the constant-coefficient grid operator are class stencils whose 27 x 27 tables do not
   depend on the grid size, so they are computed on a small PROXY hierarchy with the same coefficients (same kappa,
   same h2 = 1/(nx-1)^2 of the true grid, as many levels, 2^levels + 1 points per refined direction at most): the
   Galerkin products of the true 10^7..10^8-row matrices never have to be formed, and a z-slab of a multi-device run
   needs nothing but its own planes.  Bit-identical to the tables extracted from the full products (the same entries
   are summed in the same order for every point of a class; tests compare both set-ups). */
static pmg_status stencil_tables_from_proxy(pmg_mgmc h, st27_table *tab /* [nlevels-1], level l < top */, int *ok)
{
  const int top = h->nlevels - 1;
  *ok           = 0;
  int32_t pd[64][3];
  for (int q = 0; q < 3; ++q) {
    const int32_t tn = q == 0 ? h->lv[top].nx : (q == 1 ? h->lv[top].ny : h->lv[top].nz);
    const int64_t cap = ((int64_t)1 << (h->nlevels < 20 ? h->nlevels : 20)) + 1;
    pd[top][q]        = tn == 1 ? 1 : (int32_t)(tn < cap ? tn : cap);
  }
  for (int l = top; l >= 1; --l)
    for (int q = 0; q < 3; ++q) {
      const int32_t tf = q == 0 ? h->lv[l].nx : (q == 1 ? h->lv[l].ny : h->lv[l].nz), tc = q == 0 ? h->lv[l - 1].nx : (q == 1 ? h->lv[l - 1].ny : h->lv[l - 1].nz);
      pd[l - 1][q]     = tf == tc ? pd[l][q] : (pd[l][q] - 1) / 2 + 1; /* coarsened in the true hierarchy <=> coarsened here */
    }
  for (int l = top; l >= 0; --l) /* every position class of the true level must exist on the proxy level */
    for (int q = 0; q < 3; ++q) {
      const int32_t tn = q == 0 ? h->lv[l].nx : (q == 1 ? h->lv[l].ny : h->lv[l].nz);
      if (pd[l][q] != tn && pd[l][q] < 3) return PMG_SUCCESS;
    }
  const mg_level *F = &h->lv[top];
  rowsrc          src;
  memset(&src, 0, sizeof src);
  src.nx    = pd[top][0];
  src.ny    = pd[top][1];
  src.nz    = pd[top][2];
  src.kappa = h->kappa;
  src.h2    = 1. / ((F->nx - 1) * (F->nx - 1)); /* src/problems.c:24, the TRUE grid's spacing */
  for (int nn = 0; nn < 8; ++nn) {
    double dgl = h->kappa * h->kappa;
    for (int q = 0; q < nn; ++q) dgl += src.h2;
    src.diag[nn] = dgl;
  }
  hcsr Aprev;
  memset(&Aprev, 0, sizeof Aprev);
  int good = 1;
  for (int l = top; l >= 1 && good; --l) {
    hcsr P, R, Ac;
    memset(&P, 0, sizeof P);
    memset(&R, 0, sizeof R);
    memset(&Ac, 0, sizeof Ac);
    PMG_CALL(q1_interp(pd[l], pd[l - 1], &P));
    PMG_CALL(hcsr_transpose(&P, &R));
    rowsrc sl = src;
    if (l < top) sl.A = &Aprev;
    PMG_CALL(galerkin_rap(&sl, l == top ? 7 : 64, &P, &R, &Ac));
    good = st27_extract(pd[l - 1][0], pd[l - 1][1], pd[l - 1][2], &Ac, tab[l - 1].coef, tab[l - 1].have);
    hcsr_free(&P);
    hcsr_free(&R);
    hcsr_free(&Aprev);
    Aprev = Ac;
  }
  hcsr_free(&Aprev);
  *ok = good;
  return PMG_SUCCESS;
}

/* assembled CSR of a class-stencil operator on the full nx*ny*nz grid (for the dense coarse factorisation) */
static pmg_status st27_to_csr(int nx, int ny, int nz, const st27_table *t, hcsr *A)
{
  const int32_t n = nx * ny * nz;
  memset(A, 0, sizeof *A);
  A->nr = A->nc = n;
  A->rp         = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
  A->ci         = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 27);
  A->v          = (double *)malloc(sizeof(double) * (size_t)n * 27);
  PMG_CHECK(A->rp && A->ci && A->v, PMG_ERR_MEM, "out of host memory");
  int32_t nnz = 0;
  for (int32_t k = 0; k < nz; ++k)
    for (int32_t j = 0; j < ny; ++j)
      for (int32_t i = 0; i < nx; ++i) {
        const int32_t row = i + nx * (j + ny * k);
        const int     cls = (i == 0 ? 0 : (i == nx - 1 ? 2 : 1)) + 3 * (j == 0 ? 0 : (j == ny - 1 ? 2 : 1)) + 9 * (k == 0 ? 0 : (k == nz - 1 ? 2 : 1));
        A->rp[row]        = nnz;
        int e             = 0;
        for (int dz = -1; dz <= 1; ++dz)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx, ++e)
              if (i + dx >= 0 && i + dx < nx && j + dy >= 0 && j + dy < ny && k + dz >= 0 && k + dz < nz) {
                A->ci[nnz]  = row + dx + nx * (dy + ny * dz);
                A->v[nnz++] = t->coef[27 * cls + e];
              }
      }
  A->rp[n] = nnz;
  return PMG_SUCCESS;
}

/* MATLRC operators of a class-stencil hierarchy (PCGAMGMC_SetUpHierarchy, src/pc_gamgmc.c:157-196): the factor B of the
   finest level is restricted level by level ON THE DEVICE, column by column, with the restriction kernels of the
   V-cycle (B_{l-1} = P_l^T B_l, :177), every level sampler gets A_l + B_l S B_l^T; the coarsest block is returned in
   natural numbering for the dense factorisation (src/pc_chols.c:119-153). */
static pmg_status mg_restrict(pmg_mgmc h, int l, double *r_fine, double *b_coarse, void *stream);

static pmg_status stencil_attach_lowrank(pmg_mgmc h, double **B0_host)
{
  const int  top = h->nlevels - 1, k = h->lrc_k;
  mg_level  *F   = &h->lv[top];
  double    *Bcur = NULL, *tmp = NULL;
  const int32_t nrows = h->dist ? h->n_io : F->n;
  *B0_host            = NULL;
  PMG_CALL(pmg_dev_alloc((void **)&Bcur, sizeof(double) * (size_t)F->ld * k));
  PMG_HIP(hipMemset(Bcur, 0, sizeof(double) * (size_t)F->ld * k));
  PMG_CALL(pmg_dev_alloc((void **)&tmp, sizeof(double) * (size_t)nrows));
  for (int c = 0; c < k; ++c) { /* natural host column (this rank's planes) -> cvec */
    PMG_HIP(hipMemcpy(tmp, h->lrc_B + (size_t)nrows * c, sizeof(double) * (size_t)nrows, hipMemcpyHostToDevice));
    PMG_CALL(pmg_grid_to_cvec(F->g, tmp, Bcur + (size_t)F->ld * c, NULL));
  }
  PMG_HIP(hipDeviceSynchronize());
  pmg_dev_free(tmp);
  if (h->dist) PMG_CALL(pmg_lrc_build_dev(&F->lrc, k, F->ld, Bcur, h->lrc_S, dist_det_sweep, h, F->distributed ? mg_reduce : NULL, h)); /* the slab sweeps run in pmg_dist: the update is applied around them here */
  else PMG_CALL(pmg_grid_set_lowrank_dev(F->g, k, Bcur, h->lrc_S));
  for (int l = top; l >= 1; --l) {
    mg_level *Lv = &h->lv[l], *Cc = &h->lv[l - 1];
    double   *Bnext = NULL;
    PMG_CALL(pmg_dev_alloc((void **)&Bnext, sizeof(double) * (size_t)Cc->ld * k));
    PMG_HIP(hipMemset(Bnext, 0, sizeof(double) * (size_t)Cc->ld * k));
    for (int c = 0; c < k; ++c) PMG_CALL(mg_restrict(h, l, Bcur + (size_t)Lv->ld * c, Bnext + (size_t)Cc->ld * c, NULL)); /* B_{l-1} = P_l^T B_l */
    PMG_HIP(hipDeviceSynchronize());
    pmg_dev_free(Bcur);
    Bcur = Bnext;
    if (Cc->is_st27) { /* a sampled level (not the Cholesky level) */
      st27_det_ctx ctx = {h, Cc};
      PMG_CALL(pmg_lrc_build_dev(&Cc->lrc, k, Cc->ld, Bcur, h->lrc_S, st27_det_sweep, &ctx, Cc->distributed ? mg_reduce : NULL, h));
    }
  }
  /* coarsest block to the host, natural numbering (the padded layout minus its ghost planes) */
  mg_level *C0 = &h->lv[0];
  double   *B0 = (double *)malloc(sizeof(double) * (size_t)C0->n * k);
  PMG_CHECK(B0, PMG_ERR_MEM, "out of host memory");
  for (int c = 0; c < k; ++c) PMG_HIP(hipMemcpy(B0 + (size_t)C0->n * c, Bcur + (size_t)C0->ld * c + C0->off, sizeof(double) * (size_t)C0->n, hipMemcpyDeviceToHost));
  pmg_dev_free(Bcur);
  *B0_host = B0;
  return PMG_SUCCESS;
}

/* set-up from class-stencil tables: every level below the grid level is a class-stencil level in padded natural
   order, transfers are the matrix-free Q1 kernels, the coarsest level is factored from the expanded table */
static pmg_status mgmc_setup_stencil(pmg_mgmc h, const st27_table *tab)
{
  const int top = h->nlevels - 1;
  mg_level *F   = &h->lv[top];
  F->is_grid    = 1;
  if (!F->g) {
    PMG_CALL(pmg_grid_create(F->nx, F->ny, F->nz, 0, F->nz, h->kappa, &F->g));
    h->own_grid = 1;
  }
  PMG_CALL(pmg_grid_set_omega(F->g, h->omega));
  PMG_CALL(pmg_grid_set_sweep_type(F->g, h->sweep_type));
  PMG_CALL(pmg_grid_cvec_len(F->g, &F->ld));
  F->grid_transfer = 1;
  if (h->dist) { /* which levels stay distributed */
    const int nr1 = h->nranks + 1;
    int64_t   cap = 0, rep = (int64_t)1 << 19;
    PMG_CALL(pmg_dist_get_info(h->dist, NULL, NULL, &cap));
    if (getenv("PMG_MG_REPLICATE_BELOW")) rep = atoll(getenv("PMG_MG_REPLICATE_BELOW"));
    F->distributed = h->nranks > 1;
    int replicated = !F->distributed;
    for (int l = top - 1; l >= 0; --l) {
      mg_level      *Lv = &h->lv[l];
      const int32_t *c  = h->cuts + (size_t)l * nr1;
      int            minplanes = 1 << 30;
      for (int r = 0; r < h->nranks; ++r) minplanes = c[r + 1] - c[r] < minplanes ? c[r + 1] - c[r] : minplanes;
      if (!replicated && (Lv->n <= rep || minplanes < 1 || l == 0)) replicated = 1;
      if (replicated) {
        PMG_CHECK(!h->lv[l + 1].distributed || Lv->n <= cap, PMG_ERR_SUP, "level %d (%d unknowns) has to be replicated but exceeds the exchange capacity (%lld): use more levels", l, Lv->n, (long long)cap);
      } else {
        Lv->distributed = 1;
        Lv->kz0         = c[h->rank];
        Lv->nzl         = c[h->rank + 1] - c[h->rank];
      }
    }
  }
  for (int l = top - 1; l >= 0; --l) {
    mg_level *Lv = &h->lv[l];
    level_set_padded(Lv);
    if (l > 0 || h->coarse_type == 1) PMG_CALL(st27_install(Lv, tab[l].coef, tab[l].have, h->omega));
    if (l > 0) Lv->nat_transfer = 1;
  }
  double *B0_host = NULL; /* coarsest block of the low-rank factor, natural numbering, for the dense factorisation */
  if (h->lrc_k) PMG_CALL(stencil_attach_lowrank(h, &B0_host));
  if (h->coarse_type == 0) {
    mg_level *C0 = &h->lv[0];
    hcsr      A0;
    PMG_CHECK(!C0->distributed, PMG_ERR_SUP, "the Cholesky level must not be distributed");
    PMG_CALL(st27_to_csr(C0->nx, C0->ny, C0->nz, &tab[0], &A0));
    pmg_status st = pmg_chol_create_csr_lowrank(C0->n, A0.rp, A0.ci, A0.v, h->lrc_k, B0_host, h->lrc_S, &h->chol);
    hcsr_free(&A0);
    free(B0_host);
    PMG_CALL(st);
  } else {
    free(B0_host);
  }
  for (int l = 0; l <= top; ++l) {
    mg_level *Lv = &h->lv[l];
    PMG_CALL(pmg_dev_alloc((void **)&Lv->b, sizeof(double) * (size_t)Lv->ld));
    PMG_CALL(pmg_dev_alloc((void **)&Lv->x, sizeof(double) * (size_t)Lv->ld));
    PMG_CALL(pmg_dev_alloc((void **)&Lv->r, sizeof(double) * (size_t)Lv->ld));
    PMG_HIP(hipMemset(Lv->b, 0, sizeof(double) * (size_t)Lv->ld));
    PMG_HIP(hipMemset(Lv->x, 0, sizeof(double) * (size_t)Lv->ld));
    PMG_HIP(hipMemset(Lv->r, 0, sizeof(double) * (size_t)Lv->ld));
    if (st27_use_pair(Lv) || st27_use_pair_slab(Lv)) PMG_CALL(pmg_dev_alloc((void **)&Lv->x2, sizeof(double) * (size_t)Lv->ld)); /* zero-filled: the ghost planes stay zero */
    if (l >= 1 && Lv->is_grid && Lv->distributed && Lv->grid_transfer && !Lv->cpos_dev && !h->no_fused && (!h->lrc_k || (Lv->lrc && h->lv[l - 1].lrc)) && !(getenv("PMG_GRID_FUSED_RR_SLAB") && !atoi(getenv("PMG_GRID_FUSED_RR_SLAB")))) {
      /* the fused residual + restriction on a z-slab: every rank needs two planes (it hands its second and second-to-last
         ones to the neighbours) and a coarse plane of its own -- decided from the cuts, identically on every rank -- AND the
         kernel must accept this rank's own slab (limits that depend on the local layout: slabs differ by a plane).  Every
         rank dry-runs the kernel's predicate on its slab and the ranks agree with one all-reduce: a rank that would be
         refused in the cycle (after its peers had entered the next halo) makes all of them keep the two-kernel form */
      const int32_t *fc = h->cuts + (size_t)l * (size_t)(h->nranks + 1), *cc = h->cuts + (size_t)(l - 1) * (size_t)(h->nranks + 1);
      int            ok = 1;
      for (int r = 0; r < h->nranks; ++r) ok = ok && fc[r + 1] - fc[r] >= 2 && cc[r + 1] - cc[r] >= 1;
      if (ok) {
        pmgk_st27_dims CD = level_dims(&h->lv[l - 1]);
        CD.kz0            = cc[h->rank];
        CD.nz             = cc[h->rank + 1] - cc[h->rank];
        double  mine      = pmg_grid_residual_restrict_applies(Lv->g, &CD, 1, 1) ? 1.0 : 0.0, all = 0.0;
        double *flag      = NULL;
        PMG_CALL(pmg_dev_alloc((void **)&flag, sizeof(double)));
        pmg_status st = hipMemcpy(flag, &mine, sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? PMG_SUCCESS : pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "upload failed");
        if (!st) st = pmg_dist_allreduce_sum(h->dist, flag, 1, NULL);
        if (!st && hipMemcpy(&all, flag, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = pmg_set_error(PMG_ERR_GPU, __FILE__, __LINE__, "download failed");
        pmg_dev_free(flag);
        PMG_CALL(st);
        ok = all == (double)h->nranks;
      }
      if (ok) {
        int64_t own, ghost, np;
        PMG_CALL(pmg_grid_halo_plane(Lv->g, 0, 0, &own, &ghost, &np));
        PMG_CALL(pmg_dev_alloc((void **)&Lv->y2lo, sizeof(double) * 2 * (size_t)np));
        PMG_CALL(pmg_dev_alloc((void **)&Lv->y2hi, sizeof(double) * 2 * (size_t)np));
        PMG_HIP(hipMemset(Lv->y2lo, 0, sizeof(double) * 2 * (size_t)np));
        PMG_HIP(hipMemset(Lv->y2hi, 0, sizeof(double) * 2 * (size_t)np));
        Lv->rr_slab = 1;
      }
    }
  }
  PMG_CALL(pmg_dev_alloc((void **)&h->y_lay, sizeof(double) * (size_t)F->ld));
  PMG_CALL(pmg_dev_alloc((void **)&h->b_lay, sizeof(double) * (size_t)F->ld));
  PMG_HIP(hipMemset(h->y_lay, 0, sizeof(double) * (size_t)F->ld));
  PMG_HIP(hipMemset(h->b_lay, 0, sizeof(double) * (size_t)F->ld));
  h->is_setup = 1;
  return PMG_SUCCESS;
}

/* z-neighbour halo of a vector of a distributed level: the boundary planes travel to the neighbours' ghost planes */
static pmg_status halo_level(pmg_mgmc h, mg_level *Lv, double *v, void *stream)
{
  if (!Lv->distributed) return PMG_SUCCESS;
  const double *slo[2], *shi[2];
  double       *rlo[2], *rhi[2];
  int64_t       n[2];
  int           nseg;
  if (Lv->is_grid) { /* cvec: one block per colour */
    nseg = 2;
    for (int c = 0; c < 2; ++c) {
      int64_t own, ghost;
      PMG_CALL(pmg_grid_halo_plane(Lv->g, c, 0, &own, &ghost, &n[c]));
      slo[c] = v + own;
      rlo[c] = v + ghost;
      PMG_CALL(pmg_grid_halo_plane(Lv->g, c, 1, &own, &ghost, &n[c]));
      shi[c] = v + own;
      rhi[c] = v + ghost;
    }
  } else {
    nseg   = 1;
    n[0]   = Lv->off;
    slo[0] = v + Lv->off;
    rlo[0] = v;
    shi[0] = v + Lv->off * Lv->nzl;
    rhi[0] = v + Lv->off * ((int64_t)Lv->nzl + 1);
  }
  return pmg_dist_exchange(h->dist, nseg, slo, n, rlo, n, shi, n, rhi, n, stream);
}

/* `its` samples of the level sampler on a class-stencil level (same draw numbering as pmg_mcsor_sample_layout); on a
   z-slab the two z-parity phases of a sweep are separated by a halo of the boundary planes */
static pmg_status st27_sample(pmg_mgmc h, mg_level *Lv, int its, uint64_t seed, uint64_t *ctr, void *stream)
{
  pmgk_st27 S = Lv->st;
  S.sqrtdiag  = h->scaled ? Lv->st_sqrtd_scaled : Lv->st_sqrtd;
  for (int it = 0; it < its; ++it) {
    const int ndir = h->sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
    for (int d = 0; d < ndir; ++d) {
      const int     backward = ndir == 2 ? d : h->sweep_type == PMG_SOR_BACKWARD_SWEEP;
      const double *rhs      = Lv->b;
      if (Lv->lrc) PMG_CALL(pmg_lrc_rhs(Lv->lrc, Lv->b, seed, *ctr, &rhs, stream)); /* + B (sqrt(S) o eta), src/pc_mcgibbs.c:130-140 */
      if (Lv->distributed && Lv->x2 && st27_use_pair_slab(Lv) && st27_pair_slab_mode() != 2) { /* out of place: x2 <- sweep(x), phase by phase, then the buffers swap */
        PMG_KERNEL(pmgk_st27_sweep_pp_phase(&S, backward, 0, h->omega, 1, seed, *ctr, rhs, Lv->x, Lv->x2, stream));
        PMG_CALL(halo_level(h, Lv, Lv->x2, stream));
        PMG_KERNEL(pmgk_st27_sweep_pp_phase(&S, backward, 1, h->omega, 1, seed, (*ctr)++, rhs, Lv->x, Lv->x2, stream));
        PMG_CALL(halo_level(h, Lv, Lv->x2, stream));
        double *t = Lv->x;
        Lv->x     = Lv->x2;
        Lv->x2    = t;
      } else if (Lv->distributed) {
        PMG_KERNEL(pmgk_st27_sweep_phase(&S, backward, 0, h->omega, 1, seed, *ctr, rhs, Lv->x, stream));
        PMG_CALL(halo_level(h, Lv, Lv->x, stream));
        PMG_KERNEL(pmgk_st27_sweep_phase(&S, backward, 1, h->omega, 1, seed, (*ctr)++, rhs, Lv->x, stream));
        PMG_CALL(halo_level(h, Lv, Lv->x, stream));
      } else {
        PMG_CALL(st27_one_sweep(Lv, &S, backward, h->omega, 1, seed, (*ctr)++, rhs, Lv->x_unset, stream));
        Lv->x_unset = 0;
      }
      if (Lv->lrc) PMG_CALL(pmg_lrc_rhs_done(Lv->lrc, stream));
      if (Lv->lrc) PMG_CALL(pmg_lrc_post(Lv->lrc, backward ? PMG_SOR_BACKWARD_SWEEP : PMG_SOR_FORWARD_SWEEP, Lv->x, stream)); /* src/mc_sor.c:101-112 */
      if (Lv->lrc && Lv->distributed) PMG_CALL(halo_level(h, Lv, Lv->x, stream)); /* the repair changed boundary planes */
    }
  }
  return PMG_SUCCESS;
}

/* per-level noise seed: levels draw from independent streams */
static uint64_t level_seed(uint64_t seed, int level) { return seed + 0x9E3779B97F4A7C15ull * (uint64_t)(level + 1); }

static pmg_status mg_smooth(pmg_mgmc h, int l, uint64_t seed, uint64_t *ctr, void *stream)
{
  mg_level *Lv = &h->lv[l];
  if (Lv->is_grid && h->dist && Lv->lrc) { /* MATLRC on z-slabs: noise term, one slab sweep, repair, per directional sweep */
    for (int it = 0; it < h->nu; ++it) {
      const int ndir = h->sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
      for (int d = 0; d < ndir; ++d) {
        const int     dir = ndir == 2 ? (d == 0 ? PMG_SOR_FORWARD_SWEEP : PMG_SOR_BACKWARD_SWEEP) : h->sweep_type;
        const double *rhs = Lv->b;
        PMG_CALL(pmg_lrc_rhs(Lv->lrc, Lv->b, level_seed(seed, l), *ctr, &rhs, stream));
        PMG_CALL(pmg_dist_sample_cvec(h->dist, rhs, Lv->x, 1, h->scaled, dir, level_seed(seed, l), *ctr, ctr, stream));
        PMG_CALL(pmg_lrc_rhs_done(Lv->lrc, stream));
        PMG_CALL(pmg_lrc_post(Lv->lrc, dir, Lv->x, stream));
        PMG_CALL(halo_level(h, Lv, Lv->x, stream)); /* the repair changed boundary planes */
      }
    }
  } else if (Lv->is_grid && h->dist) PMG_CALL(pmg_dist_sample_cvec(h->dist, Lv->b, Lv->x, h->nu, h->scaled, h->sweep_type, level_seed(seed, l), *ctr, ctr, stream)); /* leaves the ghost planes current */
  else if (Lv->is_grid) PMG_CALL(pmg_grid_sample_cvec(Lv->g, Lv->b, Lv->x, h->nu, h->scaled, level_seed(seed, l), *ctr, ctr, stream));
  else if (Lv->is_st27) PMG_CALL(st27_sample(h, Lv, h->nu, level_seed(seed, l), ctr, stream));
  else if (Lv->dm) PMG_CALL(pmg_distmcsor_sample_layout(Lv->dm, Lv->b, Lv->x, h->nu, h->scaled, h->sweep_type, level_seed(seed, l), *ctr, ctr, stream)); /* row block: refreshes the ghost rows first, leaves them current */
  else PMG_CALL(pmg_mcsor_sample_layout(Lv->mc, Lv->b, Lv->x, h->nu, h->scaled, level_seed(seed, l), *ctr, ctr, stream));
  return PMG_SUCCESS;
}

/* b_coarse = P_l^T r_fine (MatRestrict).  A z-slab restricts into the coarse planes it owns (K with fine plane 2K on
   this rank) and needs r on its ghost planes for that (one halo of r); into a replicated coarse level the owned part is
   followed by an all-gather.  r_fine's ghost planes are overwritten. */
static pmg_status mg_restrict(pmg_mgmc h, int l, double *r_fine, double *b_coarse, void *stream)
{
  mg_level      *Lv = &h->lv[l], *Cc = &h->lv[l - 1];
  pmgk_st27_dims CD   = level_dims(Cc);
  double        *bc   = b_coarse;
  const int      fold = Lv->distributed && !Cc->distributed; /* distributed -> replicated */
  const int32_t *cc   = h->dist ? h->cuts + (size_t)(l - 1) * (size_t)(h->nranks + 1) : NULL;
  if (Lv->distributed) PMG_CALL(halo_level(h, Lv, r_fine, stream));
  if (fold) {
    CD.kz0 = cc[h->rank];
    CD.nz  = cc[h->rank + 1] - cc[h->rank];
    bc     = b_coarse + Cc->off * CD.kz0; /* plane K of the full-size vector = plane K - kz0 of the shifted one */
  }
  if (Lv->grid_transfer) { /* matrix-free */
    pmgk_grid_layout GL;
    PMG_CALL(pmg_grid_get_kernel_layout(Lv->g, &GL));
    PMG_KERNEL(pmgk_q1_restrict(&GL, &CD, Lv->cpos_dev, r_fine, bc, stream));
  } else if (Lv->nat_transfer) {
    const pmgk_st27_dims FD = level_dims(Lv);
    PMG_KERNEL(pmgk_st27_restrict(&FD, &CD, r_fine, bc, stream));
  } else {
    if (Lv->dm) PMG_CALL(pmg_distmcsor_refresh_layout(Lv->dm, r_fine, stream)); /* row block: the rows of P^T read r on other ranks' rows */
    /* inside a cycle on one device the restriction also sets the zero guess of the coarse level: its rows are the rows of P^T, the
       padding of the layout is never written (zero since the allocation) -- one fill kernel less per level (PMG_MG_FUSED_ZERO=0) */
    static int fz = -1;
    if (fz < 0) {
      const char *e = getenv("PMG_MG_FUSED_ZERO");
      fz            = e ? atoi(e) : 1;
    }
    const int needs_zero = l - 1 >= 1 || h->coarse_type != 0;
    double   *zero = fz && needs_zero && !h->dist && !Lv->dm && !Cc->dm && b_coarse == Cc->b && Cc->mc && Lv->R_nrows == Cc->n ? Cc->x : NULL;
    PMG_KERNEL(pmgk_csr_spmv_rows(Lv->R_nrows, Lv->R_rowpos, Lv->R_rowptr, Lv->R_col, Lv->R_val, r_fine, b_coarse, 0, zero, stream));
    if (zero) Cc->x_zeroed = 1;
    if (Lv->dm && l == h->rb_fold) PMG_CALL(rb_fold_allgather(h, b_coarse, stream)); /* the replicated level below: every rank needs the whole right-hand side */
  }
  if (fold) {
    int64_t offs[64], cnts[64];
    PMG_CHECK(h->nranks <= 64, PMG_ERR_ARG_OUTOFRANGE, "too many ranks");
    for (int r = 0; r < h->nranks; ++r) {
      offs[r] = Cc->off * ((int64_t)cc[r] + 1);
      cnts[r] = Cc->off * (int64_t)(cc[r + 1] - cc[r]);
    }
    PMG_CALL(pmg_dist_allgather(h->dist, b_coarse, offs, cnts, stream));
  }
  return PMG_SUCCESS;
}

/* z-slab grid level: b_coarse = P^T (b - A x) in one kernel.  The residual on the two ghost planes needs x two planes deep:
   every rank sends its second and second-to-last planes (both colours, one exchange) into the neighbours' y2 buffers; b is
   current on its ghost planes (pmg_mgmc_sample exchanges them once per call: in-place form only).  The coarse planes
   this rank owns, the fold into a replicated coarse level and the all-gather behind it are those of mg_restrict. */
static pmg_status mg_residual_restrict_slab(pmg_mgmc h, int l, void *stream)
{
  mg_level      *Lv = &h->lv[l], *Cc = &h->lv[l - 1];
  pmgk_st27_dims CD   = level_dims(Cc);
  double        *bc   = Cc->b;
  const int      fold = !Cc->distributed;
  const int32_t *cc   = h->cuts + (size_t)(l - 1) * (size_t)(h->nranks + 1);
  const double  *slo[2], *shi[2];
  double        *rlo[2], *rhi[2];
  int64_t        n[2];
  for (int c = 0; c < 2; ++c) {
    int64_t own, ghost;
    PMG_CALL(pmg_grid_halo_plane(Lv->g, c, 0, &own, &ghost, &n[c]));
    slo[c] = Lv->x + own + n[c]; /* plane 1 */
    rlo[c] = Lv->y2lo + (int64_t)c * n[c];
    PMG_CALL(pmg_grid_halo_plane(Lv->g, c, 1, &own, &ghost, &n[c]));
    shi[c] = Lv->x + own - n[c]; /* plane nz - 2 */
    rhi[c] = Lv->y2hi + (int64_t)c * n[c];
  }
  PMG_CALL(pmg_dist_exchange(h->dist, 2, slo, n, rlo, n, shi, n, rhi, n, stream));
  if (fold) {
    CD.kz0 = cc[h->rank];
    CD.nz  = cc[h->rank + 1] - cc[h->rank];
    bc     = Cc->b + Cc->off * CD.kz0;
  }
  int done = 0;
  PMG_CALL(pmg_grid_residual_restrict(Lv->g, Lv->b, Lv->x, Lv->kz0 > 0 ? Lv->y2lo : NULL, Lv->kz0 + Lv->nzl < Lv->nz ? Lv->y2hi : NULL, &CD, bc, &done, stream));
  PMG_CHECK(done, PMG_ERR_PLIB, "level %d: the fused residual + restriction refused a slab the set-up had accepted", l);
  if (Lv->lrc) PMG_CALL(pmg_lrc_residual_sub_restricted(Lv->lrc, Cc->lrc, Lv->x, Cc->b, stream)); /* - P^T B S B^T x = - B_{l-1} (S B^T x); B^T x summed over the ranks */
  if (fold) {
    int64_t offs[64], cnts[64];
    PMG_CHECK(h->nranks <= 64, PMG_ERR_ARG_OUTOFRANGE, "too many ranks");
    for (int r = 0; r < h->nranks; ++r) {
      offs[r] = Cc->off * ((int64_t)cc[r] + 1);
      cnts[r] = Cc->off * (int64_t)(cc[r + 1] - cc[r]);
    }
    PMG_CALL(pmg_dist_allgather(h->dist, Cc->b, offs, cnts, stream));
  }
  return PMG_SUCCESS;
}

/* x_fine += P_l e_coarse (MatInterpolateAdd).  only_color (grid level): -1 = both colours, else just that one. */
static pmg_status mg_prolong_add(pmg_mgmc h, int l, const double *e_coarse, double *x_fine, int only_color, void *stream)
{
  mg_level *Lv = &h->lv[l], *Cc = &h->lv[l - 1];
  /* a z-slab also interpolates onto its in-domain ghost planes (from its coarse planes + coarse ghost planes): the
     same arithmetic the owner does, so the fine ghost planes stay current without an exchange */
  const int glo = Lv->distributed && Lv->kz0 > 0, ghi = Lv->distributed && Lv->kz0 + Lv->nzl < Lv->nz;
  if (Lv->grid_transfer) { /* matrix-free */
    pmgk_grid_layout     GL;
    const pmgk_st27_dims CD = level_dims(Cc);
    PMG_CALL(pmg_grid_get_kernel_layout(Lv->g, &GL));
    PMG_KERNEL(pmgk_q1_prolong_add(&GL, &CD, Lv->cpos_dev, -glo, Lv->nzl + glo + ghi, only_color, e_coarse, x_fine, stream));
  } else if (Lv->nat_transfer) {
    const pmgk_st27_dims FD = level_dims(Lv), CD = level_dims(Cc);
    PMG_KERNEL(pmgk_st27_prolong_add(&FD, &CD, Lv->kz0 - glo, Lv->nzl + glo + ghi, e_coarse, x_fine, stream));
  } else {
    PMG_KERNEL(pmgk_csr_spmv_rows(Lv->P_nrows, Lv->P_rowpos, Lv->P_rowptr, Lv->P_col, Lv->P_val, e_coarse, x_fine, 1, NULL, stream));
  }
  return PMG_SUCCESS;
}

/* the low-rank update a level's sampler applies (held by the level, or by the grid object of a single-device grid level) */
static pmg_lrc mg_level_lrc(mg_level *Lv) { return Lv->lrc ? Lv->lrc : (Lv->is_grid ? pmg_grid_lrc(Lv->g) : NULL); }

/* The noise terms B (sqrt(S) o eta) of a cycle need one draw of k numbers per directional sweep and level
   (src/pc_mcgibbs.c:130-134) -- a launch of one wavefront in front of every sweep, ~2 us each on the cycle's critical path
   (timing probe without them: 0.732 -> 0.715 ms per 257^3 k = 3 sample).  Which (seed, counter) every sweep of the cycle
   will ask for is known when the cycle starts: they are all drawn by ONE launch here and handed to the levels' updates,
   which take them when the sweep asks with a matching (seed, counter) and draw themselves otherwise.  Same numbers either
   way.  Single device only; PMG_LRC_BATCH=0 switches it off. */
#define MG_ETA_STRIDE 64
static pmg_status mg_draw_lowrank_noise(pmg_mgmc h, uint64_t seed, const uint64_t *ctr, int on, void *stream)
{
  if (!h->eta_batch_mode) {
    const char *e     = getenv("PMG_LRC_BATCH");
    h->eta_batch_mode = (e && !atoi(e)) ? -1 : 1;
  }
  const int env = h->eta_batch_mode > 0;
  const int top = h->nlevels - 1, ndir = h->sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
  uint64_t  seeds[PMGK_NORMAL_BATCH_MAX], ctrs[PMGK_NORMAL_BATCH_MAX];
  int       first[64], count[64], nslots = 0;
  pmg_lrc   any = NULL;
  for (int l = 0; l <= top; ++l) {
    mg_level *Lv = &h->lv[l];
    pmg_lrc   lr = mg_level_lrc(Lv);
    first[l] = count[l] = 0;
    if (!lr) continue;
    pmg_lrc_preset_eta(lr, 0, 0, 0, NULL, 0); /* forget the last cycle's */
    if (!on || !env || h->dist || !pmg_lrc_is_local(lr) || pmg_lrc_rank(lr) > MG_ETA_STRIDE) continue;
    const int n = l >= 1 ? 2 * h->nu * ndir : (h->coarse_type != 0 ? h->coarse_its * ndir : 0); /* pre- and post-smoothing; the sampled coarsest level */
    if (n <= 0 || nslots + n > PMGK_NORMAL_BATCH_MAX) continue;
    first[l] = nslots;
    count[l] = n;
    for (int i = 0; i < n; ++i) {
      seeds[nslots + i] = pmg_lrc_noise_seed(level_seed(seed, l));
      ctrs[nslots + i]  = ctr[l] + (uint64_t)i;
    }
    nslots += n;
    any = lr;
  }
  if (!nslots) return PMG_SUCCESS;
  if (!h->eta_batch) PMG_CALL(pmg_dev_alloc((void **)&h->eta_batch, sizeof(double) * MG_ETA_STRIDE * PMGK_NORMAL_BATCH_MAX));
  PMG_KERNEL(pmgk_fill_normal_batch(nslots, pmg_lrc_rank(any), seeds, ctrs, pmg_lrc_sqrtS(any), h->eta_batch, MG_ETA_STRIDE, stream)); /* S is the same on every level (src/pc_gamgmc.c:170-176) */
  for (int l = 0; l <= top; ++l)
    if (count[l]) pmg_lrc_preset_eta(mg_level_lrc(&h->lv[l]), level_seed(seed, l), ctr[l], count[l], h->eta_batch + (size_t)first[l] * MG_ETA_STRIDE, MG_ETA_STRIDE);
  return PMG_SUCCESS;
}

/* one multiplicative V-cycle on lv[top].b -> lv[top].x; x starts at zero on every level below the top, and on the
   top level too unless top_has_guess */
static pmg_status mg_vcycle(pmg_mgmc h, uint64_t seed, uint64_t sample, int top_has_guess, void *stream)
{
  const int top = h->nlevels - 1;
  uint64_t  ctr[64];
  PMG_CHECK(h->nlevels <= 64, PMG_ERR_ARG_OUTOFRANGE, "too many levels");
  for (int l = 0; l <= top; ++l) ctr[l] = sample * MG_DRAWS_PER_SAMPLE;
  for (int l = 0; l <= top; ++l) h->lv[l].x_zeroed = 0; /* (a cycle that ended in an error may have left one set) */
  if (h->lrc_k > 0) PMG_CALL(mg_draw_lowrank_noise(h, seed, ctr, 1, stream));
  for (int l = top; l >= 1; --l) {
    mg_level *Lv = &h->lv[l], *Cc = &h->lv[l - 1];
    if (l < top || !top_has_guess) {
      /* class-stencil levels with the out-of-place sweep: no memset, the first sweep is told that its input is zero */
      if (Lv->x_zeroed) Lv->x_zeroed = 0;
      else if (Lv->x2 && st27_use_pair(Lv) && h->nu >= 1) Lv->x_unset = 1; /* (also under a low-rank update: the noise term changes b, the repair acts on the swept iterate -- round 3 excluded those levels without need and paid three zero fills and twelve full phase launches per 257^3 sample) */
      else PMG_KERNEL(pmgk_fill_zero(Lv->x, Lv->ld, stream));
    }
    pmg_lrc flrc = Lv->is_grid ? (Lv->lrc ? Lv->lrc : pmg_grid_lrc(Lv->g)) : NULL; /* the grid level's low-rank update (held by the level on a slab hierarchy, by the grid object otherwise) */
    pmg_lrc slrc = Lv->is_grid ? flrc : Lv->lrc;
    pmg_lrc_expect_residual(slrc, 1); /* the last repair of the pre-smoothing also starts the residual's low-rank term */
    const pmg_status sst = mg_smooth(h, l, seed, &ctr[l], stream);
    pmg_lrc_expect_residual(slrc, sst == PMG_SUCCESS); /* (an error: forget) */
    PMG_CALL(sst);
    if (Lv->is_grid && Lv->grid_transfer && !Lv->distributed && !Lv->cpos_dev && !h->no_fused && (!flrc || (pmg_lrc_is_local(flrc) && Cc->is_st27 && pmg_lrc_is_local(Cc->lrc)))) { /* b_{l-1} = P^T (b - A x) in one pass */
      const pmgk_st27_dims CD = level_dims(Cc);
      int                  done = 0;
      PMG_CALL(pmg_grid_residual_restrict(Lv->g, Lv->b, Lv->x, NULL, NULL, &CD, Cc->b, &done, stream));
      if (done) {
        if (flrc) PMG_CALL(pmg_lrc_residual_sub_restricted(flrc, Cc->lrc, Lv->x, Cc->b, stream)); /* - P^T B S B^T x = - B_{l-1} (S B^T x) */
        continue;
      }
    }
    if (Lv->is_grid && Lv->rr_slab && Lv->b == h->b_lay) { /* z-slab, in-place form: b's ghost planes are current */
      PMG_CALL(mg_residual_restrict_slab(h, l, stream));
      continue;
    }
    if (Lv->is_grid) PMG_CALL(pmg_grid_residual_cvec(Lv->g, Lv->b, Lv->x, Lv->r, stream));
    else if (Lv->is_st27) {
      if (st27_use_pair(Lv) || (st27_use_pair_slab(Lv) && st27_pair_slab_mode() != 3)) PMG_KERNEL(pmgk_st27_residual_pair(&Lv->st, Lv->b, Lv->x, Lv->r, stream));
      else PMG_KERNEL(pmgk_st27_residual(&Lv->st, Lv->b, Lv->x, Lv->r, stream));
      if (Lv->lrc) PMG_CALL(pmg_lrc_residual_sub(Lv->lrc, Lv->x, Lv->r, stream)); /* PCMGSetResidual(..., As[l]), src/pc_gamgmc.c:194 */
    }
    else if (Lv->dm) PMG_CALL(pmg_distmcsor_residual_layout(Lv->dm, Lv->b, Lv->x, Lv->r, stream)); /* row block: + the all-reduced low-rank term */
    else PMG_CALL(pmg_mcsor_residual_layout(Lv->mc, Lv->b, Lv->x, Lv->r, stream));
    if (Lv->is_grid && Lv->lrc) PMG_CALL(pmg_lrc_residual_sub(Lv->lrc, Lv->x, Lv->r, stream)); /* z-slabs: the update of the fine level lives here, not in the grid object */
    PMG_CALL(mg_restrict(h, l, Lv->r, Cc->b, stream));
  }
  {
    mg_level *C0 = &h->lv[0];
    if (h->coarse_type == 0) {
      PMG_CALL(pmg_chol_sample(h->chol, C0->b + C0->off, C0->x + C0->off, 1, level_seed(seed, 0), ctr[0], stream));
    } else {
      if (C0->x_zeroed) C0->x_zeroed = 0;
      else PMG_KERNEL(pmgk_fill_zero(C0->x, C0->ld, stream));
      if (C0->is_st27) PMG_CALL(st27_sample(h, C0, h->coarse_its, level_seed(seed, 0), &ctr[0], stream));
      else PMG_CALL(pmg_mcsor_sample_layout(C0->mc, C0->b, C0->x, h->coarse_its, h->scaled, level_seed(seed, 0), ctr[0], &ctr[0], stream));
    }
  }
  for (int l = 1; l <= top; ++l) {
    mg_level *Lv = &h->lv[l];
    /* with omega = 1 a colour sweep never reads the old values of the colour it updates (the (1-omega) x term is
       gone), so the colour the post-smoother visits first needs no correction: it is overwritten unread */
    static int no_skip = -1;
    if (no_skip < 0) no_skip = getenv("PMG_MG_PROLONG_BOTH") != NULL;
    const int first = h->sweep_type == PMG_SOR_BACKWARD_SWEEP ? 1 : 0;
    const int only  = (h->omega == 1.0 && h->nu >= 1 && !no_skip) ? 1 - first : -1;
    PMG_CALL(mg_prolong_add(h, l, h->lv[l - 1].x, Lv->x, only, stream));
    pmg_lrc_expect_residual(mg_level_lrc(Lv), 0); /* no residual behind the post-smoothing */
    PMG_CALL(mg_smooth(h, l, seed, &ctr[l], stream));
  }
  if (h->lrc_k > 0) PMG_CALL(mg_draw_lowrank_noise(h, seed, ctr, 0, stream)); /* nothing outside this cycle takes its noise terms */
  return PMG_SUCCESS;
}

/* ALGORITHMIC bytes of ONE sample as the cycle above is built (what bench.py's V-cycle lines divide by their time):
   every launch of mg_vcycle / pmg_mgmc_sample counted with the operands it must read and write once, per level --
   this rank's owned unknowns N_l; halos, memsets of skipped zero fills and cache re-reads are NOT counted.
     grid sweep (matrix-free 7-point)     24 N per directional sweep (32 N at omega != 1): read y, b, write y
     class-stencil sweep (27 x 27 table)  24 N; 16 N for the zero-guess pre-sweep of the out-of-place kernel (x not read)
     sliced-ELL sweep                     12 nnz + 40 N
     zero fill of a level iterate         8 N (skipped where the zero-guess sweep applies)
     grid residual + restriction fused    16 N + 8 N_c;  unfused: residual 24 N, restriction 8 N + 8 N_c
     class-stencil residual, restriction  24 N, 8 N + 8 N_c;  sliced-ELL: 12 nnz + 28 N, 12 nnz_P + 12 N_c + 8 N
     prolongation                         grid, omega = 1: ONE colour, 8 N + 8 N_c; otherwise 16 N + 8 N_c;
                                          CSR: 12 nnz_P + 20 N + 8 N_c
     exact coarse sample                  8 N_0^2 (two triangles) ; Gibbs coarse: its sweeps
     literal correction form              + outer residual 24 N and update 24 N on the finest level, + its zero fill
     low-rank update per directional sweep on ns support rows: noise term (8 k + 24) ns, repair (16 k + 24) ns;
                                          residual term (8 k + 8) ns + (8 k + 16) ns' (ns' = ns unrestricted, coarse rows restricted)
   per_level (may be NULL): nlevels entries, the smoothing / residual / transfer bytes charged to the level they run on
   (a transfer is charged to its FINE level). */
pmg_status pmg_mgmc_get_algorithmic_bytes(pmg_mgmc h, double *total, double *per_level)
{
  PMG_CHECK(h && total, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mgmc_setup first");
  const int    top  = h->nlevels - 1;
  const int    ndir = h->sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
  const double nsw  = (double)h->nu * ndir; /* directional sweeps per smoothing leg */
  *total            = 0.0;
  for (int l = 0; l <= top; ++l) {
    const mg_level *Lv = &h->lv[l];
    const double    N  = Lv->is_grid || Lv->padded ? (double)Lv->nx * Lv->ny * Lv->nzl : (double)(Lv->rb ? Lv->rb_nowned : Lv->n);
    double          by = 0.0;
    pmg_lrc         lr = Lv->is_grid ? (Lv->lrc ? Lv->lrc : pmg_grid_lrc(Lv->g)) : Lv->lrc;
    int32_t         k  = 0;
    int64_t         ns = 0;
    int             lr_dense = 0;
    /* a rank whose slab or block misses B's support launches none of the low-rank kernels: ns = 0, no bytes (round 3 charged
       such a rank the dense form's (24k + 48) N per sweep and inflated the summed roofline of the multi-GPU low-rank line) */
    if (lr) pmg_lrc_get_sizes(lr, &k, &ns, &lr_dense);
    if (lr_dense) ns = (int64_t)N; /* dense factors: every row of the level */
    const double sweep = Lv->is_grid ? (h->omega == 1.0 ? 24.0 : 32.0) * N : (Lv->is_st27 ? 24.0 * N : 12.0 * (double)Lv->A_nnz + 40.0 * N);
    const double lrsw  = lr ? ((8.0 * k + 24.0) + (16.0 * k + 24.0)) * (double)ns : 0.0;
    if (l == 0) {
      by = h->coarse_type == 0 ? 8.0 * N * N : 8.0 * N + h->coarse_its * ndir * (sweep + lrsw);
    } else {
      const mg_level *Cc = &h->lv[l - 1];
      const double    Nc = Cc->is_grid || Cc->padded ? (double)Cc->nx * Cc->ny * Cc->nzl : (double)(Cc->rb ? Cc->rb_nowned : Cc->n);
      const int has_guess = l == top && !h->correction_form;
      const int unset     = !has_guess && Lv->x2 && st27_use_pair(Lv) && h->nu >= 1; /* zero-guess out-of-place sweep */
      by += 2.0 * nsw * (sweep + lrsw);
      if (!has_guess) by += unset ? -8.0 * N : 8.0 * N; /* the first sweep does not read x / the zero fill */
      const int fused = Lv->is_grid && Lv->grid_transfer && !Lv->cpos_dev && !h->no_fused &&
                        (Lv->distributed ? Lv->rr_slab && !h->correction_form : (!lr || (pmg_lrc_is_local(lr) && Cc->is_st27 && pmg_lrc_is_local(Cc->lrc))));
      if (fused) by += 16.0 * N + 8.0 * Nc;
      else if (Lv->is_grid || Lv->is_st27) by += 24.0 * N + 8.0 * N + 8.0 * Nc;
      else by += 12.0 * (double)Lv->A_nnz + 28.0 * N + 12.0 * (double)Lv->P_nnz + 12.0 * Nc + 8.0 * N;
      if (lr) {
        int32_t kc = 0;
        int64_t nc = 0;
        int     cdense = 0;
        if (Cc->lrc) pmg_lrc_get_sizes(Cc->lrc, &kc, &nc, &cdense);
        if (fused && Cc->lrc && !cdense) by += (8.0 * k + 8.0) * (double)ns + (8.0 * k + 16.0) * (double)nc;
        else by += (8.0 * k + 8.0) * (double)ns + (8.0 * k + 16.0) * (double)ns;
      }
      const int one_colour = Lv->is_grid && h->omega == 1.0 && h->nu >= 1 && !getenv("PMG_MG_PROLONG_BOTH");
      if (Lv->grid_transfer || Lv->nat_transfer) by += (one_colour ? 8.0 : 16.0) * N + 8.0 * Nc;
      else by += 12.0 * (double)Lv->P_nnz + 20.0 * N + 8.0 * Nc;
      if (l == top && h->correction_form) by += 48.0 * N + (lr ? (8.0 * k + 8.0) * (double)ns + (8.0 * k + 16.0) * (double)ns : 0.0);
    }
    if (per_level) per_level[l] = by;
    *total += by;
  }
  return PMG_SUCCESS;
}

static pmg_status lvl_to_layout(mg_level *Lv, const double *nat, double *lay, void *stream)
{
  return Lv->is_grid ? pmg_grid_to_cvec(Lv->g, nat, lay, stream) : pmg_mcsor_to_layout(Lv->mc, nat, lay, stream);
}
static pmg_status lvl_from_layout(mg_level *Lv, const double *lay, double *nat, void *stream)
{
  return Lv->is_grid ? pmg_grid_from_cvec(Lv->g, lay, nat, stream) : pmg_mcsor_from_layout(Lv->mc, lay, nat, stream);
}

pmg_status pmg_mgmc_sample(pmg_mgmc h, const double *b_nat, double *y_nat, int32_t its, int guesszero, uint64_t seed, uint64_t counter0, uint64_t *counter_out, pmg_sample_callback cb, void *cbctx, void *stream)
{
  PMG_CHECK(h && b_nat && y_nat, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mgmc_setup first");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  mg_level    *F     = &h->lv[h->nlevels - 1];
  const size_t bytes = sizeof(double) * (size_t)F->ld;
  PMG_CALL(lvl_to_layout(F, b_nat, h->b_lay, stream));
  PMG_CALL(lvl_to_layout(F, y_nat, h->y_lay, stream));
  if (h->correction_form && F->distributed) PMG_CALL(halo_level(h, F, h->y_lay, stream)); /* the outer residual reads the ghost planes of y */
  if (!h->correction_form && F->rr_slab) PMG_CALL(halo_level(h, F, h->b_lay, stream));     /* the fused residual + restriction reads b on the ghost planes */
  if (h->correction_form && F->dm) PMG_CALL(pmg_distmcsor_refresh_layout(F->dm, h->y_lay, stream)); /* ... the ghost rows of a row block */
  for (int32_t it = 0; it < its; ++it) {
    if (!h->correction_form) {
      /* The cycle run IN PLACE on (b, y): a stationary linear sweep satisfies S(b, y) = y + S(b - A y, 0) with the
         same noise, so "w = b - A y; y += MG(w)" (src/pc_gamgmc.c:253-256) and the V-cycle started from the guess y
         are the same chain up to rounding -- without the outer residual, the axpy and a memset (about 15 % of a
         sample).  pmg_mgmc_set_correction_form(mg, 1) selects the literal form. */
      double *sb = F->b, *sx = F->x;
      F->b       = h->b_lay;
      F->x       = h->y_lay;
      pmg_status st = mg_vcycle(h, seed, counter0 + (uint64_t)it, !(it == 0 && guesszero), stream);
      F->b          = sb;
      F->x          = sx;
      PMG_CALL(st);
    } else if (it == 0 && guesszero) { /* y = MG(b), src/pc_gamgmc.c:243-246 */
      PMG_HIP(hipMemcpyAsync(F->b, h->b_lay, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
      PMG_CALL(mg_vcycle(h, seed, counter0 + (uint64_t)it, 0, stream));
      PMG_HIP(hipMemcpyAsync(h->y_lay, F->x, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    } else { /* w = b - A y; work = MG(w); y += work, src/pc_gamgmc.c:253-256 */
      if (F->is_grid) PMG_CALL(pmg_grid_residual_cvec(F->g, h->b_lay, h->y_lay, F->b, stream));
      else if (F->dm) PMG_CALL(pmg_distmcsor_residual_layout(F->dm, h->b_lay, h->y_lay, F->b, stream));
      else PMG_CALL(pmg_mcsor_residual_layout(F->mc, h->b_lay, h->y_lay, F->b, stream));
      if (F->is_grid && F->lrc) PMG_CALL(pmg_lrc_residual_sub(F->lrc, h->y_lay, F->b, stream)); /* z-slabs */
      PMG_CALL(mg_vcycle(h, seed, counter0 + (uint64_t)it, 0, stream));
      PMG_KERNEL(pmgk_axpy(F->ld, 1.0, F->x, h->y_lay, stream));
    }
    if (cb) { /* pg->scb(it, y, ctx), src/pc_gamgmc.c:258 */
      PMG_CALL(lvl_from_layout(F, h->y_lay, y_nat, stream));
      const int rc = cb(it, y_nat, h->n_io, cbctx);
      PMG_CHECK(rc == 0, rc, "sample callback returned %d", rc);
    }
  }
  PMG_CALL(lvl_from_layout(F, h->y_lay, y_nat, stream));
  if (counter_out) *counter_out = counter0 + (uint64_t)its;
  return PMG_SUCCESS;
}

/* ---- single-kernel entry points of one level (diagnostics: the full-size parity tests run ONE kernel of the V-cycle
   on caller-supplied vectors in the level's own layout and compare sampled rows with the oracle) ---------------------- */
static pmg_status level_checked(pmg_mgmc h, int32_t level, int need_coarser, mg_level **Lv)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(h->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mgmc_setup first");
  PMG_CHECK(!h->dist, PMG_ERR_SUP, "level diagnostics are a single-device feature");
  PMG_CHECK(level >= (need_coarser ? 1 : 0) && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  *Lv = &h->lv[level];
  return PMG_SUCCESS;
}

/* kind: 0 = grid level (colour-partitioned cvec), 1 = class-stencil level, 2 = sliced-ELL level, 3 = dense coarsest level;
   ld = vector length, off = position of natural index 0 for kinds 1 and 3 (plane-padded natural order), else 0 */
pmg_status pmg_mgmc_get_level_layout(pmg_mgmc h, int32_t level, int32_t *kind, int64_t *ld, int64_t *off)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 0, &Lv));
  if (kind) *kind = Lv->is_grid ? 0 : (Lv->is_st27 ? 1 : (Lv->mc ? 2 : 3));
  if (ld) *ld = Lv->ld;
  if (off) *off = Lv->padded ? Lv->off : 0;
  return PMG_SUCCESS;
}

/* the 27 x 27 class table and the per-class noise scale of a class-stencil level, as the kernels use them */
pmg_status pmg_mgmc_get_level_stencil(pmg_mgmc h, int32_t level, double *coef_host, double *sqrtdiag_host)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 0, &Lv));
  PMG_CHECK(Lv->is_st27, PMG_ERR_ARG_WRONGSTATE, "level %d is not a class-stencil level", level);
  if (coef_host) PMG_HIP(hipMemcpy(coef_host, Lv->st_coef, sizeof(double) * 27 * 27, hipMemcpyDeviceToHost));
  if (sqrtdiag_host) PMG_HIP(hipMemcpy(sqrtdiag_host, h->scaled ? Lv->st_sqrtd_scaled : Lv->st_sqrtd, sizeof(double) * 27, hipMemcpyDeviceToHost));
  return PMG_SUCCESS;
}

/* ONE directional sweep of the level sampler (all colours) with the raw (seed, counter) pair */
pmg_status pmg_mgmc_level_sweep(pmg_mgmc h, int32_t level, int backward, int noisy, uint64_t seed, uint64_t counter, const double *b_lvl, double *x_lvl, void *stream)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 0, &Lv));
  PMG_CHECK(b_lvl && x_lvl, PMG_ERR_ARG_NULL, "null vector");
  PMG_CHECK(Lv->is_st27, PMG_ERR_SUP, "level %d: only class-stencil levels (use pmg_grid_* / pmg_mcsor_* for the others)", level);
  pmgk_st27 S = Lv->st;
  S.sqrtdiag  = h->scaled ? Lv->st_sqrtd_scaled : Lv->st_sqrtd;
  if (Lv->x2 && st27_use_pair(Lv)) { /* the production kernel: out of place into the level's second buffer, then copied back */
    PMG_KERNEL(pmgk_st27_sweep_pp(&S, backward != 0, h->omega, noisy != 0, seed, counter, b_lvl, x_lvl, Lv->x2, stream));
    PMG_HIP(hipMemcpyAsync(x_lvl, Lv->x2, sizeof(double) * (size_t)Lv->ld, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PMG_SUCCESS;
  }
  PMG_KERNEL(pmgk_st27_sweep(&S, backward != 0, h->omega, noisy != 0, seed, counter, b_lvl, x_lvl, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_level_residual(pmg_mgmc h, int32_t level, const double *b_lvl, const double *x_lvl, double *r_lvl, void *stream)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 0, &Lv));
  PMG_CHECK(b_lvl && x_lvl && r_lvl, PMG_ERR_ARG_NULL, "null vector");
  if (Lv->is_grid) return pmg_grid_residual_cvec(Lv->g, b_lvl, x_lvl, r_lvl, stream);
  if (Lv->is_st27) {
    if (st27_use_pair(Lv)) PMG_KERNEL(pmgk_st27_residual_pair(&Lv->st, b_lvl, x_lvl, r_lvl, stream));
    else PMG_KERNEL(pmgk_st27_residual(&Lv->st, b_lvl, x_lvl, r_lvl, stream));
    return PMG_SUCCESS;
  }
  PMG_CHECK(Lv->mc, PMG_ERR_SUP, "level %d has no residual kernel", level);
  return pmg_mcsor_residual_layout(Lv->mc, b_lvl, x_lvl, r_lvl, stream);
}

/* b_coarse (level-1) = P^T r_fine (level); x_fine (level) += P e_coarse (level-1), both colours */
pmg_status pmg_mgmc_level_restrict(pmg_mgmc h, int32_t level, double *r_fine, double *b_coarse, void *stream)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 1, &Lv));
  PMG_CHECK(r_fine && b_coarse, PMG_ERR_ARG_NULL, "null vector");
  return mg_restrict(h, level, r_fine, b_coarse, stream);
}

/* the V-cycle's fused step b_coarse = P^T (b - A x) on a grid level; PMG_ERR_SUP where the cycle runs the two steps */
pmg_status pmg_mgmc_level_residual_restrict(pmg_mgmc h, int32_t level, const double *b_lvl, const double *x_lvl, double *b_coarse, void *stream)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 1, &Lv));
  PMG_CHECK(b_lvl && x_lvl && b_coarse, PMG_ERR_ARG_NULL, "null vector");
  int done = 0;
  if (Lv->is_grid && Lv->grid_transfer && !Lv->distributed && !Lv->lrc && !Lv->cpos_dev && !pmg_grid_lrc(Lv->g)) {
    const pmgk_st27_dims CD = level_dims(&h->lv[level - 1]);
    PMG_CALL(pmg_grid_residual_restrict(Lv->g, b_lvl, x_lvl, NULL, NULL, &CD, b_coarse, &done, stream));
  }
  PMG_CHECK(done, PMG_ERR_SUP, "level %d: no fused residual + restriction (z-slab, low-rank update, permuted or semicoarsened coarse level)", level);
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_level_prolong_add(pmg_mgmc h, int32_t level, const double *e_coarse, double *x_fine, void *stream)
{
  mg_level *Lv;
  PMG_CALL(level_checked(h, level, 1, &Lv));
  PMG_CHECK(e_coarse && x_fine, PMG_ERR_ARG_NULL, "null vector");
  return mg_prolong_add(h, level, e_coarse, x_fine, -1, stream);
}

/* the MATLRC update of a level (src/pc_gamgmc.c:157-196), single device: held by the grid object on the grid level, by the
   level on class-stencil levels */
static pmg_status level_lrc(pmg_mgmc h, int32_t level, int need_coarser, mg_level **Lv, pmg_lrc *l)
{
  PMG_CALL(level_checked(h, level, need_coarser, Lv));
  *l = (*Lv)->is_grid ? ((*Lv)->lrc ? (*Lv)->lrc : pmg_grid_lrc((*Lv)->g)) : (*Lv)->lrc;
  PMG_CHECK(*l, PMG_ERR_ARG_WRONGSTATE, "level %d carries no low-rank update", level);
  return PMG_SUCCESS;
}

pmg_status pmg_mgmc_level_lowrank_factors(pmg_mgmc h, int32_t level, int32_t *k, int64_t *ns, int64_t *rows_host, double *B_host, double *Bb_fwd_host, double *Bb_bwd_host)
{
  mg_level *Lv;
  pmg_lrc   l;
  PMG_CALL(level_lrc(h, level, 0, &Lv, &l));
  return pmg_lrc_get_compact(l, k, ns, rows_host, B_host, Bb_fwd_host, Bb_bwd_host);
}

/* y -= Bb (B^T y) with the level's factors, MCSORPostSOR_LRC (src/mc_sor.c:101-112) */
pmg_status pmg_mgmc_level_lowrank_post(pmg_mgmc h, int32_t level, int backward, double *y_lvl, void *stream)
{
  mg_level *Lv;
  pmg_lrc   l;
  PMG_CALL(level_lrc(h, level, 0, &Lv, &l));
  PMG_CHECK(y_lvl, PMG_ERR_ARG_NULL, "null vector");
  return pmg_lrc_post(l, backward ? PMG_SOR_BACKWARD_SWEEP : PMG_SOR_FORWARD_SWEEP, y_lvl, stream);
}

/* the low-rank part of the level residual (PCMGSetResidual on the MATLRC operator, src/pc_gamgmc.c:194):
   restricted = 0: out (this level's layout) -= B_l (S B_l^T x);  restricted = 1: out (the next coarser level's layout)
   -= B_{l-1} (S B_l^T x), the form the cycle uses behind the fused residual + restriction */
pmg_status pmg_mgmc_level_lowrank_residual_sub(pmg_mgmc h, int32_t level, int restricted, const double *x_lvl, double *out, void *stream)
{
  mg_level *Lv, *Cc;
  pmg_lrc   l, lc;
  PMG_CALL(level_lrc(h, level, restricted, &Lv, &l));
  PMG_CHECK(x_lvl && out, PMG_ERR_ARG_NULL, "null vector");
  if (!restricted) return pmg_lrc_residual_sub(l, x_lvl, out, stream);
  PMG_CALL(level_lrc(h, level - 1, 0, &Cc, &lc));
  return pmg_lrc_residual_sub_restricted(l, lc, x_lvl, out, stream);
}

pmg_status pmg_mgmc_destroy(pmg_mgmc *hp)
{
  if (!hp || !*hp) return PMG_SUCCESS;
  pmg_mgmc h = *hp;
  for (int l = 0; l < h->nlevels; ++l) {
    mg_level *Lv = &h->lv[l];
    if (h->own_grid || !Lv->is_grid) pmg_grid_destroy(&Lv->g);
    pmg_distmcsor_destroy(&Lv->dm);
    pmg_mcsor_destroy(&Lv->mc);
    free(Lv->rb_colors), free(Lv->rb_send_ptr), free(Lv->rb_recv_ptr), free(Lv->rb_counts), free(Lv->rb_send_idx), free(Lv->rb_recv_src), free(Lv->rb_recv_idx);
    pmg_dev_free(Lv->b);
    pmg_dev_free(Lv->x);
    pmg_dev_free(Lv->r);
    pmg_dev_free(Lv->x2);
    pmg_dev_free(Lv->y2lo);
    pmg_dev_free(Lv->y2hi);
    pmg_dev_free(Lv->cpos_dev);
    pmg_dev_free(Lv->st_coef);
    pmg_dev_free(Lv->st_idiag);
    pmg_dev_free(Lv->st_sqrtd);
    pmg_dev_free(Lv->st_sqrtd_scaled);
    pmg_lrc_destroy(&Lv->lrc);
    pmg_dev_free(Lv->P_rowpos);
    pmg_dev_free(Lv->P_rowptr);
    pmg_dev_free(Lv->P_col);
    pmg_dev_free(Lv->P_val);
    pmg_dev_free(Lv->R_rowpos);
    pmg_dev_free(Lv->R_rowptr);
    pmg_dev_free(Lv->R_col);
    pmg_dev_free(Lv->R_val);
    hcsr_free(&Lv->A_host);
    hcsr_free(&Lv->P_host);
    free(Lv->A_rp_own);
    free(Lv->A_ci_own);
    free(Lv->P_rp_own);
    free(Lv->P_ci_own);
  }
  pmg_chol_destroy(&h->chol);
  pmg_dev_free(h->y_lay);
  pmg_dev_free(h->b_lay);
  pmg_dev_free(h->eta_batch);
  free(h->lrc_B);
  free(h->lrc_S);
  free(h->cuts);
  free(h->rb_c0_starts);
  pmg_dev_free(h->rb_fold_pos);
  pmg_dev_free(h->rb_fold_iota);
  pmg_dev_free(h->rb_fold_buf);
  free(h->lv);
  free(h);
  *hp = NULL;
  return PMG_SUCCESS;
}
