/* MCSOR on an assembled AIJ matrix -- host side (C11).
 *
 * Mirrors the reference object `MCSOR` (include/parmgmc/mc_sor.h:17-30, src/mc_sor.c): create / set omega /
 * set sweep type / set up / apply / destroy, plus the sample loops built on it (src/pc_mcgibbs.c:155-188,
 * src/pc_sorgibbs.c:76-134).  Set-up does on the host, once, what MCSORSetUp does (src/mc_sor.c:553-605):
 * diagonal pointers (:126-150), colouring (:441-454), idiag (:114-124) -- and then re-lays the matrix out for
 * the GPU: rows renumbered colour by colour, each colour padded to whole 64-row slices, off-diagonal entries
 * stored slice-wise column-major (see kernels_csr.hip).
 */
#include "pmg_internal.h"
#include <math.h>

struct pmg_mcsor_s {
  /* borrowed host CSR (valid until setup) */
  int32_t        n;
  const int32_t *rowptr, *colidx;
  const double  *vals;
  int32_t       *rowptr_own, *colidx_own; /* 32-bit copies of a 64-bit PetscInt matrix (pmg_mcsor_create_csr_idx) */
  double        *vals_own;                /* arrays handed over by pmg_mcsor_adopt_arrays */
  /* options */
  double   omega;
  int      omega_changed;
  int      natural_order;     /* 1: rows ascending inside a colour whatever their locality (levels of a hierarchy: pmg_mcsor_set_natural_order) */
  int      idiag_by_division; /* PCPARSOR's rule: omega / d in one rounding (src/pc_parsor.c:69-81) instead of (1/d) * omega */
  int      type;
  int      rule;
  int32_t *user_colors; /* owned copy */
  /* set-up products */
  int      is_setup;
  int32_t  ncolors;
  int32_t *colors;     /* [n] host */
  int32_t *cslice;     /* [ncolors+1] first slice of each colour */
  int32_t *orig_host;  /* [ld] */
  double  *diag_host;  /* [ld] diagonal in permuted order (1 in pad rows) */
  pmgk_sell S;         /* device arrays */
  double  *idiag_dev, *sqrtd_dev, *sqrtd_scaled_dev;
  double  *b_p, *y_p, *r_p; /* permuted scratch vectors */
  pmg_lrc  lrc;             /* MATLRC: rank-k update B S B^T (src/mc_sor.c:572-595) */
  int64_t  noise_row0;      /* global row of local row 0 (row block of a distributed matrix) */
};

/* --- colouring rules -------------------------------------------------------------------------------- */

/* first-fit in natural row order over structural neighbours */
static pmg_status color_greedy(pmg_mcsor mc)
{
  const int32_t n = mc->n;
  int32_t      *mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
  PMG_CHECK(mark, PMG_ERR_MEM, "out of host memory");
  for (int32_t r = 0; r <= n; ++r) mark[r] = -1;
  for (int32_t r = 0; r < n; ++r) mc->colors[r] = -1;
  int32_t nc = 0;
  for (int32_t r = 0; r < n; ++r) {
    for (int32_t k = mc->rowptr[r]; k < mc->rowptr[r + 1]; ++k) {
      const int32_t c = mc->colidx[k];
      if (c != r && mc->colors[c] >= 0) mark[mc->colors[c]] = r;
    }
    int32_t col = 0;
    while (mark[col] == r) ++col;
    mc->colors[r] = col;
    if (col + 1 > nc) nc = col + 1;
  }
  free(mark);
  mc->ncolors = nc;
  return PMG_SUCCESS;
}

/* First-fit, then first-fit ONCE MORE with the rows visited class by class, the last class first, rows ascending inside a
   class (one round of Culberson's iterated greedy).  Visiting whole classes one after the other can never need more colours
   than there are classes, and turning their order round frees the thin last class of a first-fit colouring: the P1 matrices
   of the refined lshape.msh and their aggregation-Galerkin levels go from 6 classes to 5 (6 033 ... 377 089 rows), i.e. one
   dependent launch fewer per sweep.  Deterministic, O(nnz), a different but equally valid multicolour sweep. */
static pmg_status color_iterated(pmg_mcsor mc)
{
  PMG_CALL(color_greedy(mc));
  const int32_t n = mc->n, nc0 = mc->ncolors;
  if (nc0 <= 2) return PMG_SUCCESS;
  int32_t *first = (int32_t *)calloc((size_t)nc0 + 1, sizeof(int32_t)), *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1)), *mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nc0 + 1));
  int32_t *newc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!first || !order || !mark || !newc) {
    free(first), free(order), free(mark), free(newc);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  /* bucket the rows by (reversed) class, ascending inside a class */
  for (int32_t r = 0; r < n; ++r) ++first[nc0 - 1 - mc->colors[r] + 1];
  for (int32_t c = 0; c < nc0; ++c) first[c + 1] += first[c];
  for (int32_t r = 0; r < n; ++r) order[first[nc0 - 1 - mc->colors[r]]++] = r;
  for (int32_t c = 0; c <= nc0; ++c) mark[c] = -1;
  for (int32_t r = 0; r < n; ++r) newc[r] = -1;
  int32_t nc = 0;
  for (int32_t q = 0; q < n; ++q) {
    const int32_t r = order[q];
    for (int32_t k = mc->rowptr[r]; k < mc->rowptr[r + 1]; ++k) {
      const int32_t c = mc->colidx[k];
      if (c != r && newc[c] >= 0) mark[newc[c]] = r;
    }
    int32_t col = 0;
    while (mark[col] == r) ++col; /* col < nc0: the rows of one old class are not coupled, so the classes visited so far bound it */
    newc[r] = col;
    if (col + 1 > nc) nc = col + 1;
  }
  memcpy(mc->colors, newc, sizeof(int32_t) * (size_t)n);
  mc->ncolors = nc;
  free(first), free(order), free(mark), free(newc);
  return PMG_SUCCESS;
}

/* In which order the rows of a colour are laid out is nobody's business but the gathers': a row's update does not depend on its
   position (the sum runs in the row's own storage order, the noise is keyed on the row's original number), so any order inside a
   colour gives the same bits.  The caller's numbering decides how far apart in memory the y[col] of a row lie -- the node
   numbering uniform mesh refinement leaves (old nodes first, new ones appended level after level) is the worst case: the sliced-
   ELL sweep then moves 1.6 x its algorithmic bytes.  A breadth-first numbering of the matrix graph (Cuthill-McKee without the
   degree sort) puts neighbours next to each other; it is used when it shortens the total index distance sum |pos(r) - pos(c)|
   over the stored entries to less than 0.7 of the natural order's (structured and mesher-ordered matrices keep their own).
   Returns the visiting sequence (malloc'ed) or NULL for the natural order.  PMG_SELL_LOCALITY=0 never, 2 always. */
static int32_t *locality_order(int32_t n, const int32_t *rowptr, const int32_t *colidx)
{
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("PMG_SELL_LOCALITY");
    env           = e ? atoi(e) : 1;
  }
  if (!env || n < 4096) return NULL; /* small matrices live in the L2 whatever their numbering */
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (!order || !rank) {
    free(order), free(rank);
    return NULL;
  }
  for (int32_t r = 0; r < n; ++r) rank[r] = -1;
  int32_t head = 0, tail = 0;
  for (int32_t s = 0; s < n; ++s) { /* every component, from its lowest row */
    if (rank[s] >= 0) continue;
    rank[s]       = tail;
    order[tail++] = s;
    while (head < tail) {
      const int32_t r = order[head++];
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int32_t c = colidx[k];
        if (rank[c] < 0) {
          rank[c]       = tail;
          order[tail++] = c;
        }
      }
    }
  }
  double dn = 0.0, db = 0.0;
  for (int32_t r = 0; r < n; ++r)
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int32_t c = colidx[k];
      dn += (double)(r > c ? r - c : c - r);
      db += (double)(rank[r] > rank[c] ? rank[r] - rank[c] : rank[c] - rank[r]);
    }
  free(rank);
  if (env < 2 && !(db < 0.7 * dn)) {
    free(order);
    return NULL;
  }
  return order;
}

/* level(r) = 1 + max level of the neighbours that precede r: sweeping the levels in ascending order is the
   lexicographic Gauss-Seidel sweep of the reference's serial path (one colour, src/mc_sor.c:397-410) */
static pmg_status color_lexlevels(pmg_mcsor mc)
{
  int32_t nc = 0;
  for (int32_t r = 0; r < mc->n; ++r) {
    int32_t m = -1;
    for (int32_t k = mc->rowptr[r]; k < mc->rowptr[r + 1]; ++k) {
      const int32_t c = mc->colidx[k];
      if (c < r && mc->colors[c] > m) m = mc->colors[c];
    }
    mc->colors[r] = m + 1;
    if (m + 2 > nc) nc = m + 2;
  }
  mc->ncolors = nc;
  return PMG_SUCCESS;
}

static pmg_status color_user(pmg_mcsor mc)
{
  int32_t nc = 0;
  for (int32_t r = 0; r < mc->n; ++r) {
    PMG_CHECK(mc->user_colors[r] >= 0, PMG_ERR_ARG_OUTOFRANGE, "negative colour at row %d", r);
    mc->colors[r] = mc->user_colors[r];
    if (mc->colors[r] + 1 > nc) nc = mc->colors[r] + 1;
  }
  for (int32_t r = 0; r < mc->n; ++r)
    for (int32_t k = mc->rowptr[r]; k < mc->rowptr[r + 1]; ++k) {
      const int32_t c = mc->colidx[k];
      PMG_CHECK(c == r || mc->colors[c] != mc->colors[r], PMG_ERR_ARG_WRONG, "rows %d and %d are coupled but share colour %d: not a distance-1 colouring (use PMG_COLORING_LEXLEVELS for the serial one-colour behaviour)", r, c, mc->colors[r]);
    }
  mc->ncolors = nc;
  return PMG_SUCCESS;
}

/* --- public ------------------------------------------------------------------------------------------ */

pmg_status pmg_mcsor_create_csr(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, pmg_mcsor *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(n >= 0, PMG_ERR_ARG_OUTOFRANGE, "n = %d", n);
  PMG_CHECK(rowptr && (colidx || rowptr[n] == 0) && (vals || rowptr[n] == 0), PMG_ERR_ARG_NULL, "null CSR array");
  pmg_mcsor mc = (pmg_mcsor)calloc(1, sizeof *mc);
  PMG_CHECK(mc, PMG_ERR_MEM, "out of host memory");
  mc->n             = n;
  mc->rowptr        = rowptr;
  mc->colidx        = colidx;
  mc->vals          = vals;
  mc->omega         = 1.0; /* src/mc_sor.c:637 */
  mc->omega_changed = 1;   /* src/mc_sor.c:629 */
  mc->type          = PMG_SOR_FORWARD_SWEEP; /* src/mc_sor.c:636 */
  mc->rule          = PMG_COLORING_GREEDY;
  *out              = mc;
  return PMG_SUCCESS;
}

/* the same for either PetscInt width (include/parmgmc/parmgmc.h:18-24): idx_width = sizeof(PetscInt) * 8 */
pmg_status pmg_mcsor_create_csr_idx(int64_t n, const void *rowptr, const void *colidx, const double *vals, int idx_width, pmg_mcsor *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  const int32_t *rp, *ci;
  int32_t       *rpo, *cio;
  PMG_CALL(pmg_narrow_csr(n, n, rowptr, colidx, idx_width, &rp, &ci, &rpo, &cio));
  pmg_status st = pmg_mcsor_create_csr((int32_t)n, rp, ci, vals, out);
  if (st) {
    free(rpo);
    free(cio);
    return st;
  }
  (*out)->rowptr_own = rpo;
  (*out)->colidx_own = cio;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_set_coloring(pmg_mcsor mc, int rule, const int32_t *user_colors)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  PMG_CHECK(!mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "colouring must be chosen before pmg_mcsor_setup");
  PMG_CHECK(rule == PMG_COLORING_GREEDY || rule == PMG_COLORING_LEXLEVELS || rule == PMG_COLORING_USER || rule == PMG_COLORING_ITERATED, PMG_ERR_ARG_OUTOFRANGE, "unknown colouring rule %d", rule);
  if (rule == PMG_COLORING_USER) {
    PMG_CHECK(user_colors || mc->n == 0, PMG_ERR_ARG_NULL, "user colouring without colour array");
    free(mc->user_colors);
    mc->user_colors = (int32_t *)malloc(sizeof(int32_t) * (size_t)(mc->n > 0 ? mc->n : 1));
    PMG_CHECK(mc->user_colors, PMG_ERR_MEM, "out of host memory");
    if (mc->n) memcpy(mc->user_colors, user_colors, sizeof(int32_t) * (size_t)mc->n);
  }
  mc->rule = rule;
  return PMG_SUCCESS;
}

static void mcsor_free_setup(pmg_mcsor mc)
{
  free(mc->colors);
  free(mc->cslice);
  free(mc->orig_host);
  free(mc->diag_host);
  pmg_dev_free((void *)mc->S.soff);
  pmg_dev_free((void *)mc->S.swidth);
  pmg_dev_free((void *)mc->S.vals);
  pmg_dev_free((void *)mc->S.cols);
  pmg_dev_free((void *)mc->S.diag);
  pmg_dev_free((void *)mc->S.orig);
  pmg_dev_free(mc->idiag_dev);
  pmg_dev_free(mc->sqrtd_dev);
  pmg_dev_free(mc->sqrtd_scaled_dev);
  pmg_dev_free(mc->b_p);
  pmg_dev_free(mc->y_p);
  pmg_dev_free(mc->r_p);
  mc->colors = mc->cslice = mc->orig_host = NULL;
  mc->diag_host = NULL;
  memset(&mc->S, 0, sizeof mc->S);
  mc->idiag_dev = mc->sqrtd_dev = mc->sqrtd_scaled_dev = mc->b_p = mc->y_p = mc->r_p = NULL;
  mc->is_setup = 0;
}

/* idiag = (1/d)*omega (src/mc_sor.c:114-124); sqrtdiag = sqrt|d| [* sqrt((2-omega)/omega)] (src/pc_mcgibbs.c:142-153) */
static pmg_status mcsor_update_idiag(pmg_mcsor mc)
{
  const int32_t ld = mc->S.ld;
  double       *h  = (double *)malloc(sizeof(double) * 3 * (size_t)(ld > 0 ? ld : 1));
  PMG_CHECK(h, PMG_ERR_MEM, "out of host memory");
  double      *id = h, *sd = h + ld, *ss = h + 2 * (size_t)ld;
  const double s  = sqrt((2 - mc->omega) / mc->omega);
  for (int32_t r = 0; r < ld; ++r) {
    if (mc->orig_host[r] < 0) {
      id[r] = sd[r] = ss[r] = 0.0;
      continue;
    }
    const double t = 1.0 / mc->diag_host[r];
    id[r]          = mc->idiag_by_division ? mc->omega / mc->diag_host[r] : t * mc->omega;
    sd[r]          = sqrt(fabs(mc->diag_host[r]));
    ss[r]          = sd[r] * s;
  }
  if (ld) {
    PMG_HIP(hipMemcpy(mc->idiag_dev, id, sizeof(double) * (size_t)ld, hipMemcpyHostToDevice));
    PMG_HIP(hipMemcpy(mc->sqrtd_dev, sd, sizeof(double) * (size_t)ld, hipMemcpyHostToDevice));
    PMG_HIP(hipMemcpy(mc->sqrtd_scaled_dev, ss, sizeof(double) * (size_t)ld, hipMemcpyHostToDevice));
  }
  free(h);
  mc->omega_changed = 0;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_setup(pmg_mcsor mc)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  if (mc->is_setup) return PMG_SUCCESS; /* the borrowed CSR was released after the first set-up */
  const int32_t n = mc->n;
  /* structural checks + diagonal pointers (MatGetDiagonalPointers, src/mc_sor.c:126-150) */
  int32_t *diagptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  PMG_CHECK(diagptr, PMG_ERR_MEM, "out of host memory");
  for (int32_t r = 0; r < n; ++r) {
    diagptr[r] = -1;
    if (mc->rowptr[r + 1] < mc->rowptr[r]) {
      free(diagptr);
      PMG_FAIL(PMG_ERR_ARG_WRONG, "rowptr not monotone at row %d", r);
    }
    for (int32_t k = mc->rowptr[r]; k < mc->rowptr[r + 1]; ++k) {
      const int32_t c = mc->colidx[k];
      if (c < 0 || c >= n) {
        free(diagptr);
        PMG_FAIL(PMG_ERR_ARG_OUTOFRANGE, "column %d out of range in row %d", c, r);
      }
      if (c == r) diagptr[r] = k;
    }
    if (diagptr[r] < 0) {
      free(diagptr);
      PMG_FAIL(PMG_ERR_ARG_WRONG, "row %d has no stored diagonal entry", r);
    }
  }
  mc->colors = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!mc->colors) {
    free(diagptr);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  pmg_status st = mc->rule == PMG_COLORING_GREEDY ? color_greedy(mc) : mc->rule == PMG_COLORING_ITERATED ? color_iterated(mc) : mc->rule == PMG_COLORING_LEXLEVELS ? color_lexlevels(mc) : color_user(mc);
  if (st) {
    free(diagptr);
    return st;
  }
  const int32_t nc = mc->ncolors;

  /* colour-partitioned numbering: colour c occupies slices [cslice[c], cslice[c+1]) */
  int32_t *count = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
  mc->cslice     = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
  for (int32_t r = 0; r < n; ++r) count[mc->colors[r]]++;
  for (int32_t c = 0; c < nc; ++c) mc->cslice[c + 1] = mc->cslice[c] + (count[c] + 63) / 64;
  const int32_t nslices = mc->cslice[nc];
  const int32_t ld      = nslices * 64;
  mc->orig_host         = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ld > 0 ? ld : 1));
  mc->diag_host         = (double *)malloc(sizeof(double) * (size_t)(ld > 0 ? ld : 1));
  int32_t *newidx       = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t *fill         = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
  for (int32_t r = 0; r < ld; ++r) {
    mc->orig_host[r] = -1;
    mc->diag_host[r] = 1.0;
  }
  int32_t *visit = mc->rule == PMG_COLORING_LEXLEVELS || mc->natural_order ? NULL : locality_order(n, mc->rowptr, mc->colidx); /* NULL: rows ascending inside a colour, as ISColoringGetIS lists them */
  for (int32_t q = 0; q < n; ++q) {
    const int32_t r  = visit ? visit[q] : q;
    const int32_t c  = mc->colors[r];
    const int32_t nr = mc->cslice[c] * 64 + fill[c]++;
    newidx[r]        = nr;
    mc->orig_host[nr] = r;
    mc->diag_host[nr] = mc->vals[diagptr[r]];
  }
  /* slice widths and offsets */
  int32_t *swidth = (int32_t *)calloc((size_t)(nslices > 0 ? nslices : 1), sizeof(int32_t));
  int64_t *soff   = (int64_t *)calloc((size_t)nslices + 1, sizeof(int64_t));
  for (int32_t s = 0; s < nslices; ++s) {
    int32_t w = 0;
    for (int l = 0; l < 64; ++l) {
      const int32_t o = mc->orig_host[s * 64 + l];
      if (o >= 0) {
        const int32_t len = mc->rowptr[o + 1] - mc->rowptr[o] - 1;
        if (len > w) w = len;
      }
    }
    swidth[s]   = w;
    soff[s + 1] = soff[s] + (int64_t)w * 64;
  }
  const int64_t tot = soff[nslices];
  double       *sv  = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
  int32_t      *sc  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
  for (int32_t s = 0; s < nslices; ++s)
    for (int l = 0; l < 64; ++l) {
      const int32_t row = s * 64 + l;
      const int32_t o   = mc->orig_host[row];
      int32_t       j   = 0;
      if (o >= 0)
        for (int32_t k = mc->rowptr[o]; k < mc->rowptr[o + 1]; ++k) {
          if (k == diagptr[o]) continue; /* strictly lower part, then strictly upper part: src/mc_sor.c:264-265 */
          sv[soff[s] + (int64_t)j * 64 + l] = mc->vals[k];
          sc[soff[s] + (int64_t)j * 64 + l] = newidx[mc->colidx[k]];
          ++j;
        }
      for (; j < swidth[s]; ++j) {
        sv[soff[s] + (int64_t)j * 64 + l] = 0.0;
        sc[soff[s] + (int64_t)j * 64 + l] = row;
      }
    }
  free(diagptr);
  free(newidx);
  free(fill);
  free(count);
  free(visit);

  mc->S.n       = n;
  mc->S.ld      = ld;
  mc->S.noise_row0 = mc->noise_row0;
  mc->S.nslices = nslices;
  st            = pmg_dev_upload((void **)&mc->S.soff, soff, sizeof(int64_t) * ((size_t)nslices + 1));
  if (!st) st = pmg_dev_upload((void **)&mc->S.swidth, swidth, sizeof(int32_t) * (size_t)nslices);
  if (!st) st = pmg_dev_upload((void **)&mc->S.vals, sv, sizeof(double) * (size_t)tot);
  if (!st) st = pmg_dev_upload((void **)&mc->S.cols, sc, sizeof(int32_t) * (size_t)tot);
  if (!st) st = pmg_dev_upload((void **)&mc->S.diag, mc->diag_host, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_upload((void **)&mc->S.orig, mc->orig_host, sizeof(int32_t) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&mc->idiag_dev, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&mc->sqrtd_dev, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&mc->sqrtd_scaled_dev, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&mc->b_p, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&mc->y_p, sizeof(double) * (size_t)ld);
  if (!st) st = pmg_dev_alloc((void **)&mc->r_p, sizeof(double) * (size_t)ld);
  free(swidth);
  free(soff);
  free(sv);
  free(sc);
  if (st) {
    mcsor_free_setup(mc);
    return st;
  }
  mc->S.idiag       = mc->idiag_dev;
  mc->omega_changed = 1;
  mc->is_setup      = 1;
  /* the borrowed CSR is no longer needed */
  mc->rowptr = mc->colidx = NULL;
  mc->vals            = NULL;
  return PMG_SUCCESS;
}

/* internal: keep the caller's row order inside the colours (no locality renumbering).  The levels of a hierarchy: their
   transfers read and write the level vectors through the same layout, and on the bench's aggregation hierarchy the breadth-
   first layout that speeds the stand-alone sweep up by 5 % made the whole MGMC sample 1.7 % slower (0.2324 -> 0.2365 ms). */
pmg_status pmg_mcsor_set_natural_order(pmg_mcsor mc, int on)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  PMG_CHECK(!mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "before pmg_mcsor_setup");
  mc->natural_order = on;
  return PMG_SUCCESS;
}

/* internal: LocalMatInvertDiagonalForSOR's idiag = omega / d (1 / d for omega = 1; src/pc_parsor.c:69-81) */
pmg_status pmg_mcsor_set_idiag_by_division(pmg_mcsor mc, int on)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  mc->idiag_by_division = on;
  mc->omega_changed     = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_set_omega(pmg_mcsor mc, double omega)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  mc->omega         = omega;
  mc->omega_changed = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_set_sweep_type(pmg_mcsor mc, int type)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  PMG_CHECK(pmg_sweep_type_ok(type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported"); /* src/mc_sor.c:427 */
  mc->type = type;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_get_sweep_type(pmg_mcsor mc, int *type)
{
  PMG_CHECK(mc && type, PMG_ERR_ARG_NULL, "null argument");
  *type = mc->type;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_get_num_colors(pmg_mcsor mc, int32_t *ncolors)
{
  PMG_CHECK(mc && ncolors, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  *ncolors = mc->ncolors;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_get_coloring(pmg_mcsor mc, int32_t *colors)
{
  PMG_CHECK(mc && colors, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  if (mc->n) memcpy(colors, mc->colors, sizeof(int32_t) * (size_t)mc->n);
  return PMG_SUCCESS;
}

/* one direction over all colours on permuted vectors (src/mc_sor.c:256-289: colours ascending / descending) */
static pmg_status mcsor_one_sweep(pmg_mcsor mc, int dir, int noisy, int scaled, uint64_t seed, uint64_t sweep, const double *b_p, double *y_p, void *stream)
{
  pmgk_sell S = mc->S;
  S.sqrtdiag  = scaled ? mc->sqrtd_scaled_dev : mc->sqrtd_dev;
  int rc      = 0;
  pmg_trace_begin(PMG_EVENT_MULTICOL_SOR); /* PetscLogEventBegin(MULTICOL_SOR), src/mc_sor.c:221 */
  if (dir == PMG_SOR_FORWARD_SWEEP) {
    for (int32_t c = 0; c < mc->ncolors && !rc; ++c) rc = pmgk_sell_color_sweep(&S, mc->cslice[c], mc->cslice[c + 1] - mc->cslice[c], mc->omega, noisy, seed, sweep, b_p, y_p, stream);
  } else {
    for (int32_t c = mc->ncolors - 1; c >= 0 && !rc; --c) rc = pmgk_sell_color_sweep(&S, mc->cslice[c], mc->cslice[c + 1] - mc->cslice[c], mc->omega, noisy, seed, sweep, b_p, y_p, stream);
  }
  pmg_trace_end();
  PMG_KERNEL(rc);
  return PMG_SUCCESS;
}

/* one directional sweep + the low-rank repair: ctx->sor(...) then ctx->postsor(...) of src/mc_sor.c:223-236, with
   the extra noise term of PrepareRHS_LRC (src/pc_mcgibbs.c:130-140) when the sweep is a noisy one */
static pmg_status mcsor_sweep_lrc(pmg_mcsor mc, int dir, int noisy, int scaled, uint64_t seed, uint64_t sweep, const double *b_p, double *y_p, void *stream)
{
  const double *rhs = b_p;
  if (mc->lrc && noisy) PMG_CALL(pmg_lrc_rhs(mc->lrc, b_p, seed, sweep, &rhs, stream));
  PMG_CALL(mcsor_one_sweep(mc, dir, noisy, scaled, seed, sweep, rhs, y_p, stream));
  if (mc->lrc && noisy) PMG_CALL(pmg_lrc_rhs_done(mc->lrc, stream));
  if (mc->lrc) PMG_CALL(pmg_lrc_post(mc->lrc, dir, y_p, stream));
  return PMG_SUCCESS;
}

static pmg_status mcsor_ready(pmg_mcsor mc)
{
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  if (mc->omega_changed) PMG_CALL(mcsor_update_idiag(mc)); /* src/mc_sor.c:222 */
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_apply(pmg_mcsor mc, const double *b, double *y, void *stream)
{
  PMG_CHECK(mc && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(mcsor_ready(mc));
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, b, mc->b_p, stream));
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, y, mc->y_p, stream));
  if (mc->type == PMG_SOR_SYMMETRIC_SWEEP) { /* src/mc_sor.c:223-232 */
    PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_FORWARD_SWEEP, 0, 0, 0, 0, mc->b_p, mc->y_p, stream));
    PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_BACKWARD_SWEEP, 0, 0, 0, 0, mc->b_p, mc->y_p, stream));
  } else {
    PMG_CALL(mcsor_sweep_lrc(mc, mc->type, 0, 0, 0, 0, mc->b_p, mc->y_p, stream));
  }
  PMG_KERNEL(pmgk_permute_out(mc->S.ld, mc->S.orig, mc->y_p, y, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_sample(pmg_mcsor mc, const double *b, double *y, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(mc && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CHECK(scaled || mc->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  PMG_CALL(mcsor_ready(mc));
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, b, mc->b_p, stream));
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, y, mc->y_p, stream));
  uint64_t ctr = counter0;
  for (int it = 0; it < its; ++it) {
    if (mc->type == PMG_SOR_SYMMETRIC_SWEEP) { /* src/pc_mcgibbs.c:172-181 */
      PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_FORWARD_SWEEP, 1, scaled, seed, ctr++, mc->b_p, mc->y_p, stream));
      PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_BACKWARD_SWEEP, 1, scaled, seed, ctr++, mc->b_p, mc->y_p, stream));
    } else {
      PMG_CALL(mcsor_sweep_lrc(mc, mc->type, 1, scaled, seed, ctr++, mc->b_p, mc->y_p, stream));
    }
  }
  PMG_KERNEL(pmgk_permute_out(mc->S.ld, mc->S.orig, mc->y_p, y, stream));
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}

/* ---- building blocks of the row-block distributed sampler (MCSORApply_MPIAIJ, src/mc_sor.c:298-381: per colour, ghost
   update then the colour's rows) ---- */
pmg_status pmg_mcsor_set_noise_row_offset(pmg_mcsor mc, int64_t row0)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  PMG_CHECK(row0 >= 0, PMG_ERR_ARG_OUTOFRANGE, "row offset %lld", (long long)row0);
  mc->S.noise_row0 = row0;
  mc->noise_row0   = row0;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_sweep_color_layout(pmg_mcsor mc, int32_t color, int noisy, int scaled, uint64_t seed, uint64_t sweep, const double *b_lay, double *y_lay, void *stream)
{
  PMG_CHECK(mc && b_lay && y_lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(mcsor_ready(mc));
  PMG_CHECK(color >= 0 && color < mc->ncolors, PMG_ERR_ARG_OUTOFRANGE, "colour %d of %d", color, mc->ncolors);
  PMG_CHECK(!mc->lrc, PMG_ERR_SUP, "per-colour sweeps do not carry the low-rank repair");
  PMG_CHECK(!noisy || scaled || mc->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  pmgk_sell S = mc->S;
  S.sqrtdiag  = scaled ? mc->sqrtd_scaled_dev : mc->sqrtd_dev;
  PMG_KERNEL(pmgk_sell_color_sweep(&S, mc->cslice[color], mc->cslice[color + 1] - mc->cslice[color], mc->omega, noisy, seed, sweep, b_lay, y_lay, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_residual(pmg_mcsor mc, const double *b, const double *y, double *r, void *stream)
{
  PMG_CHECK(mc && b && y && r, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(mcsor_ready(mc));
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, b, mc->b_p, stream));
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, y, mc->y_p, stream));
  PMG_KERNEL(pmgk_sell_residual(&mc->S, mc->b_p, mc->y_p, mc->r_p, stream));
  if (mc->lrc) PMG_CALL(pmg_lrc_residual_sub(mc->lrc, mc->y_p, mc->r_p, stream)); /* MatMult of the MATLRC operator */
  PMG_KERNEL(pmgk_permute_out(mc->S.ld, mc->S.orig, mc->r_p, r, stream));
  return PMG_SUCCESS;
}

/* ---- entry points on vectors that already live in the colour-partitioned numbering ("layout") ---------- */

pmg_status pmg_mcsor_layout_len(pmg_mcsor mc, int32_t *ld)
{
  PMG_CHECK(mc && ld, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  *ld = mc->S.ld;
  return PMG_SUCCESS;
}

/* number of rows of the operator */
pmg_status pmg_mcsor_get_size(pmg_mcsor mc, int32_t *n)
{
  PMG_CHECK(mc && n, PMG_ERR_ARG_NULL, "null argument");
  *n = mc->n;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_get_layout(pmg_mcsor mc, int32_t *pos_of_row)
{
  PMG_CHECK(mc && pos_of_row, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  for (int32_t p = 0; p < mc->S.ld; ++p)
    if (mc->orig_host[p] >= 0) pos_of_row[mc->orig_host[p]] = p;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_to_layout(pmg_mcsor mc, const double *nat, double *lay, void *stream)
{
  PMG_CHECK(mc && nat && lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  PMG_KERNEL(pmgk_permute_in(mc->S.ld, mc->S.orig, nat, lay, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_from_layout(pmg_mcsor mc, const double *lay, double *nat, void *stream)
{
  PMG_CHECK(mc && nat && lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(mc->is_setup, PMG_ERR_ARG_WRONGSTATE, "call pmg_mcsor_setup first");
  PMG_KERNEL(pmgk_permute_out(mc->S.ld, mc->S.orig, lay, nat, stream));
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_apply_layout(pmg_mcsor mc, const double *b_lay, double *y_lay, void *stream)
{
  PMG_CHECK(mc && b_lay && y_lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(mcsor_ready(mc));
  if (mc->type == PMG_SOR_SYMMETRIC_SWEEP) {
    PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_FORWARD_SWEEP, 0, 0, 0, 0, b_lay, y_lay, stream));
    PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_BACKWARD_SWEEP, 0, 0, 0, 0, b_lay, y_lay, stream));
  } else {
    PMG_CALL(mcsor_sweep_lrc(mc, mc->type, 0, 0, 0, 0, b_lay, y_lay, stream));
  }
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_sample_layout(pmg_mcsor mc, const double *b_lay, double *y_lay, int32_t its, int scaled, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(mc && b_lay && y_lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CHECK(scaled || mc->omega == 1.0, PMG_ERR_SUP, "the unscaled (sorgibbs) noise requires omega = 1 (src/pc_sorgibbs.c:94)");
  PMG_CALL(mcsor_ready(mc));
  uint64_t ctr = counter0;
  for (int it = 0; it < its; ++it) {
    if (mc->type == PMG_SOR_SYMMETRIC_SWEEP) {
      PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_FORWARD_SWEEP, 1, scaled, seed, ctr++, b_lay, y_lay, stream));
      PMG_CALL(mcsor_sweep_lrc(mc, PMG_SOR_BACKWARD_SWEEP, 1, scaled, seed, ctr++, b_lay, y_lay, stream));
    } else {
      PMG_CALL(mcsor_sweep_lrc(mc, mc->type, 1, scaled, seed, ctr++, b_lay, y_lay, stream));
    }
  }
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}

pmg_status pmg_mcsor_residual_layout(pmg_mcsor mc, const double *b_lay, const double *y_lay, double *r_lay, void *stream)
{
  PMG_CHECK(mc && b_lay && y_lay && r_lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(mcsor_ready(mc));
  PMG_KERNEL(pmgk_sell_residual(&mc->S, b_lay, y_lay, r_lay, stream));
  if (mc->lrc) PMG_CALL(pmg_lrc_residual_sub(mc->lrc, y_lay, r_lay, stream));
  return PMG_SUCCESS;
}

static pmg_status mcsor_det_sweep(void *ctx, int dir, const double *b_lay, double *y_lay, void *stream)
{
  return mcsor_one_sweep((pmg_mcsor)ctx, dir, 0, 0, 0, 0, b_lay, y_lay, stream);
}

/* MCSORSetUp's MATLRC branch (src/mc_sor.c:572-595): B is n x k column-major in the matrix's row numbering,
   S the k diagonal entries of Sigma^-1.  k = 0 removes the update. */
pmg_status pmg_mcsor_set_lowrank(pmg_mcsor mc, int32_t k, const double *B_host, const double *S_host)
{
  PMG_CHECK(mc, PMG_ERR_ARG_NULL, "null MCSOR");
  PMG_CALL(mcsor_ready(mc));
  pmg_lrc_destroy(&mc->lrc);
  if (k == 0) return PMG_SUCCESS;
  int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(mc->n > 0 ? mc->n : 1));
  PMG_CHECK(pos, PMG_ERR_MEM, "out of host memory");
  for (int32_t p = 0; p < mc->S.ld; ++p)
    if (mc->orig_host[p] >= 0) pos[mc->orig_host[p]] = p;
  pmg_status st = pmg_lrc_build(&mc->lrc, k, mc->S.ld, mc->n, B_host, pos, S_host, mcsor_det_sweep, mc);
  free(pos);
  return st;
}

/* the CSR arrays pmg_mcsor_create_csr borrowed become the object's (malloc'd by the caller, freed with the object) */
void pmg_mcsor_adopt_arrays(pmg_mcsor mc, int32_t *rowptr, int32_t *colidx, double *vals)
{
  mc->rowptr_own = rowptr;
  mc->colidx_own = colidx;
  mc->vals_own   = vals;
}

pmg_status pmg_mcsor_destroy(pmg_mcsor *mc)
{
  if (!mc || !*mc) return PMG_SUCCESS;
  pmg_lrc_destroy(&(*mc)->lrc);
  mcsor_free_setup(*mc);
  free((*mc)->user_colors);
  free((*mc)->rowptr_own);
  free((*mc)->colidx_own);
  free((*mc)->vals_own);
  free(*mc);
  *mc = NULL;
  return PMG_SUCCESS;
}
