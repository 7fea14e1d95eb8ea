/* Row-block distribution of MATAIJ operators and hierarchies -- the host-side set-up a caller with an MPI communicator
 * needs to reach the multi-GPU samplers, in C (C11, no GPU call in the plan builders).
 *
 * The reference runs on MATMPIAIJ matrices: every rank owns a contiguous block of rows (MatGetOwnershipRange), stored as
 * a diagonal block, an off-diagonal block and the global column of every compact off-diagonal column
 * (MatMPIAIJGetSeqAIJ(A, &Ad, &Ao, &garray), src/mc_sor.c:308).  Before the rows of colour c are swept the ghost values
 * of that colour travel (MatCreateScatters, src/mc_sor.c:152-214; used at :318-319), and PCGAMGMC walks a PCMG hierarchy
 * whose levels are all distributed this way (src/pc_gamgmc.c:157-223).  What is built here from exactly those inputs:
 *
 *   pmg_rowblock_merge_mpiaij   (Ad, Ao, garray) -> the rank's rows with GLOBAL columns
 *   pmg_rowblock_color_greedy   the library's first-fit colouring of the GLOBAL matrix, computed rank after rank
 *   pmg_rowblock_plan_*         ghost set + per-colour ghost-update plan (MatCreateScatters, de-duplicated per ghost row,
 *                               one all-gather per colour)
 *   pmg_rbh_*                   the same for every level of a hierarchy: local operators, owned rows of P, owned rows of
 *                               P^T, small levels replicated -- what pmg_mgmc_set_level_rowblock & co. take
 *   pmg_rowblock_sampler_create the stand-alone multicolour sampler on a row block (device set-up included)
 *   pmg_dist_create_comm        the ipc / RCCL transport bootstrapped through the caller's byte all-gather
 *
 * Everything collective goes through ONE callback, a fixed-size byte all-gather (pmg_host_comm): MPI_Allgather in a PETSc
 * adapter, torch.distributed in the Python tests, pipes in examples/pmg_bench.c.  The plans are pinned index for index to
 * the Python builders of round 2 (parmgmc_amd/dist.py: rowblock_plan, rowblock_hierarchy) by tests/test_rowblock_c.py. */
#include "pmg_internal.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------------- */
/* host communication helpers                                                                            */
/* ---------------------------------------------------------------------------------------------------- */
static pmg_status hc_check(const pmg_host_comm *c)
{
  PMG_CHECK(c, PMG_ERR_ARG_NULL, "null host communicator");
  PMG_CHECK(c->nranks >= 1 && c->rank >= 0 && c->rank < c->nranks, PMG_ERR_ARG_OUTOFRANGE, "rank %d of %d", c->rank, c->nranks);
  PMG_CHECK(c->nranks == 1 || c->allgather, PMG_ERR_ARG_NULL, "host communicator without an all-gather callback");
  return PMG_SUCCESS;
}

/* recv: nranks * nbytes */
static pmg_status hc_allgather(const pmg_host_comm *c, const void *send, int64_t nbytes, void *recv)
{
  if (c->nranks == 1) {
    if (nbytes) memcpy(recv, send, (size_t)nbytes);
    return PMG_SUCCESS;
  }
  const int rc = c->allgather(c->ctx, send, nbytes, recv);
  PMG_CHECK(rc == 0, PMG_ERR_LIB, "the caller's all-gather callback returned %d", rc);
  return PMG_SUCCESS;
}

/* blocks of different length: *recv (malloc'd) holds the blocks in rank order, offs[0..nranks] their byte offsets */
static pmg_status hc_allgatherv(const pmg_host_comm *c, const void *send, int64_t nbytes, void **recv, int64_t *offs)
{
  const int np  = c->nranks;
  int64_t  *len = (int64_t *)malloc(sizeof(int64_t) * (size_t)np);
  PMG_CHECK(len, PMG_ERR_MEM, "out of host memory");
  pmg_status st = hc_allgather(c, &nbytes, sizeof(int64_t), len);
  int64_t    mx = 0;
  offs[0]       = 0;
  for (int r = 0; r < np && !st; ++r) {
    if (len[r] < 0) st = pmg_set_error(PMG_ERR_LIB, __FILE__, __LINE__, "negative block length from rank %d", r);
    if (len[r] > mx) mx = len[r];
    offs[r + 1] = offs[r] + len[r];
  }
  char *out = NULL, *pad = NULL, *all = NULL;
  if (!st) {
    out = (char *)malloc((size_t)(offs[np] > 0 ? offs[np] : 1));
    pad = (char *)calloc((size_t)(mx > 0 ? mx : 1), 1);
    all = (char *)malloc((size_t)(mx > 0 ? mx : 1) * (size_t)np);
    if (!out || !pad || !all) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  }
  if (!st && mx > 0) {
    if (nbytes) memcpy(pad, send, (size_t)nbytes);
    st = hc_allgather(c, pad, mx, all);
    for (int r = 0; r < np && !st; ++r)
      if (len[r]) memcpy(out + offs[r], all + (size_t)r * (size_t)mx, (size_t)len[r]);
  }
  free(len);
  free(pad);
  free(all);
  if (st) {
    free(out);
    return st;
  }
  *recv = out;
  return PMG_SUCCESS;
}

/* every rank reaches this with its local status; all return the same verdict (a failure anywhere fails everywhere) */
static pmg_status hc_agree(const pmg_host_comm *c, pmg_status mine, const char *what)
{
  int32_t  v   = (int32_t)mine;
  int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->nranks);
  if (!all) return pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  pmg_status st = hc_allgather(c, &v, sizeof v, all);
  for (int r = 0; r < c->nranks && !st; ++r)
    if (all[r] != 0 && !mine) st = pmg_set_error(all[r], __FILE__, __LINE__, "%s failed on rank %d", what, r);
  free(all);
  return mine ? mine : st;
}

static int cmp_i64(const void *a, const void *b)
{
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return (x > y) - (x < y);
}

/* sort + unique in place; returns the new length */
static int64_t sort_unique(int64_t *v, int64_t n)
{
  if (n <= 0) return 0;
  qsort(v, (size_t)n, sizeof(int64_t), cmp_i64);
  int64_t m = 1;
  for (int64_t i = 1; i < n; ++i)
    if (v[i] != v[m - 1]) v[m++] = v[i];
  return m;
}

/* index of x in the sorted array v, or -1 */
static int64_t bsearch_i64(const int64_t *v, int64_t n, int64_t x)
{
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1;
    else hi = mid;
  }
  return lo < n && v[lo] == x ? lo : -1;
}

/* rank that owns global row g: starts[r] <= g < starts[r+1] */
static int owner_of(const int64_t *starts, int np, int64_t g)
{
  int lo = 0, hi = np; /* starts[lo] <= g < starts[hi] */
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= g) lo = mid;
    else hi = mid;
  }
  return lo;
}

static int64_t idx_at(const void *a, int64_t i, int w) { return w == 64 ? ((const int64_t *)a)[i] : (int64_t)((const int32_t *)a)[i]; }

/* ---------------------------------------------------------------------------------------------------- */
/* MatMPIAIJGetSeqAIJ blocks -> rows with global columns                                                  */
/* ---------------------------------------------------------------------------------------------------- */
/* Ad: nloc x nloc with LOCAL columns (global = col + cstart), Ao: nloc x ncompact with compact columns (global =
   garray[col]), both CSR with PetscInt indices of idx_width bits.  order = PMG_ROWBLOCK_ORDER_GLOBAL: the entries of a row
   by ascending global column -- the row of the sequential AIJ matrix, so that the distributed chain equals the
   single-process chain bit for bit; PMG_ROWBLOCK_ORDER_MPIAIJ: the diagonal block's entries, then the off-diagonal
   block's -- the order in which MCSORApply_MPIAIJ visits them (src/mc_sor.c:331-333).  rp_out: nloc + 1, ci_out / v_out:
   nnz(Ad) + nnz(Ao) entries. */
pmg_status pmg_rowblock_merge_mpiaij(int32_t nloc, int64_t cstart, const void *ad_rp, const void *ad_ci, const double *ad_v, const void *ao_rp, const void *ao_ci, const double *ao_v, const void *garray, int idx_width, int order, int64_t *rp_out, int64_t *ci_out, double *v_out)
{
  PMG_CHECK(idx_width == 32 || idx_width == 64, PMG_ERR_ARG_OUTOFRANGE, "idx_width = %d (32 | 64)", idx_width);
  PMG_CHECK(order == PMG_ROWBLOCK_ORDER_GLOBAL || order == PMG_ROWBLOCK_ORDER_MPIAIJ, PMG_ERR_ARG_OUTOFRANGE, "entry order %d", order);
  PMG_CHECK(nloc >= 0 && rp_out && (nloc == 0 || (ad_rp && ao_rp)), PMG_ERR_ARG_NULL, "null argument");
  int64_t q = 0;
  rp_out[0] = 0;
  for (int32_t r = 0; r < nloc; ++r) {
    int64_t a = idx_at(ad_rp, r, idx_width), a1 = idx_at(ad_rp, r + 1, idx_width);
    int64_t o = idx_at(ao_rp, r, idx_width), o1 = idx_at(ao_rp, r + 1, idx_width);
    if (order == PMG_ROWBLOCK_ORDER_MPIAIJ) {
      for (; a < a1; ++a, ++q) ci_out[q] = idx_at(ad_ci, a, idx_width) + cstart, v_out[q] = ad_v[a];
      for (; o < o1; ++o, ++q) ci_out[q] = idx_at(garray, idx_at(ao_ci, o, idx_width), idx_width), v_out[q] = ao_v[o];
    } else { /* both blocks are sorted by column (PETSc keeps AIJ rows sorted): off-diagonal columns below cstart, the diagonal block, the rest */
      for (; o < o1 && idx_at(garray, idx_at(ao_ci, o, idx_width), idx_width) < cstart; ++o, ++q) ci_out[q] = idx_at(garray, idx_at(ao_ci, o, idx_width), idx_width), v_out[q] = ao_v[o];
      for (; a < a1; ++a, ++q) ci_out[q] = idx_at(ad_ci, a, idx_width) + cstart, v_out[q] = ad_v[a];
      for (; o < o1; ++o, ++q) ci_out[q] = idx_at(garray, idx_at(ao_ci, o, idx_width), idx_width), v_out[q] = ao_v[o];
    }
    rp_out[r + 1] = q;
  }
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* ghost-update plan of one row block                                                                     */
/* ---------------------------------------------------------------------------------------------------- */
struct pmg_rowblock_plan_s {
  int32_t  rank, nranks, ncolors, nloc, nghost;
  int64_t  row0;
  int64_t *ghosts;    /* nghost sorted global rows: local row nloc + q */
  int64_t *send_ptr;  /* ncolors + 1 */
  int32_t *send_rows; /* local OWNED rows this rank contributes per colour, by ascending global row */
  int64_t *counts;    /* [ncolors][nranks]: length of every rank's list of a colour */
  int64_t *recv_ptr;  /* ncolors + 1 */
  int32_t *recv_src;  /* index in the colour's gather buffer (rank blocks in rank order) */
  int32_t *recv_rows; /* local ghost row it refreshes */
};

void pmg_rowblock_plan_destroy(pmg_rowblock_plan *pp)
{
  if (!pp || !*pp) return;
  pmg_rowblock_plan p = *pp;
  free(p->ghosts), free(p->send_ptr), free(p->send_rows), free(p->counts), free(p->recv_ptr), free(p->recv_src), free(p->recv_rows);
  free(p);
  *pp = NULL;
}

/* row_starts[0..nranks]: first global row of every rank (row_starts[nranks] = number of rows); cols: every global column
   index of this rank's rows (any order, duplicates welcome -- only those outside the block matter); extra: further global
   rows of other ranks whose values this rank reads (columns of its restriction / of the finer interpolation);
   colors_owned: this rank's entries of a globally valid distance-1 colouring with ncolors colours (the same ncolors on
   every rank).  Collective. */
pmg_status pmg_rowblock_plan_create(const pmg_host_comm *comm, const int64_t *row_starts, int64_t ncols, const int64_t *cols, int64_t nextra, const int64_t *extra, int32_t ncolors, const int32_t *colors_owned, pmg_rowblock_plan *out)
{
  PMG_CALL(hc_check(comm));
  PMG_CHECK(out && row_starts && (ncols == 0 || cols) && (nextra == 0 || extra), PMG_ERR_ARG_NULL, "null argument");
  *out             = NULL;
  const int     np = comm->nranks, me = comm->rank;
  const int64_t r0 = row_starts[me], r1 = row_starts[me + 1], n = row_starts[np];
  const int64_t nloc = r1 - r0;
  PMG_CHECK(nloc >= 0 && nloc < INT32_MAX && ncolors >= 1 && (nloc == 0 || colors_owned), PMG_ERR_ARG_OUTOFRANGE, "row block [%lld, %lld), %d colours", (long long)r0, (long long)r1, ncolors);
  pmg_rowblock_plan p = (pmg_rowblock_plan)calloc(1, sizeof *p);
  PMG_CHECK(p, PMG_ERR_MEM, "out of host memory");
  p->rank = me, p->nranks = np, p->ncolors = ncolors, p->nloc = (int32_t)nloc, p->row0 = r0;
  pmg_status st = PMG_SUCCESS;
  int64_t   *tmp = NULL, *wantmsg = NULL, *offs = NULL, *needed = NULL, *sendmsg = NULL, *soffs = NULL;
  void      *asked_all = NULL, *lists_all = NULL;
  /* ---- the ghost set ---- */
  int64_t ng = 0;
  for (int64_t i = 0; i < ncols; ++i) ng += cols[i] < r0 || cols[i] >= r1;
  for (int64_t i = 0; i < nextra; ++i) ng += extra[i] < r0 || extra[i] >= r1;
  tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ng > 0 ? ng : 1));
  if (!tmp) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    int64_t q = 0;
    for (int64_t i = 0; i < ncols; ++i)
      if (cols[i] < r0 || cols[i] >= r1) tmp[q++] = cols[i];
    for (int64_t i = 0; i < nextra; ++i)
      if (extra[i] < r0 || extra[i] >= r1) tmp[q++] = extra[i];
    ng = sort_unique(tmp, ng);
    if (ng && (tmp[0] < 0 || tmp[ng - 1] >= n)) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "a column index lies outside [0, %lld)", (long long)n);
    if (!st && ng >= INT32_MAX - nloc) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "too many local rows");
  }
  st = hc_agree(comm, st, "collecting the ghost rows");
  if (st) goto done;
  p->ghosts = tmp, tmp = NULL, p->nghost = (int32_t)ng;
  /* ---- what I want from every owner: message = [np counts][ghosts] (the ghosts are sorted, so they are grouped by owner) ---- */
  wantmsg = (int64_t *)calloc((size_t)(np + ng), sizeof(int64_t));
  offs    = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  soffs   = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  if (!wantmsg || !offs || !soffs) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    for (int64_t q = 0; q < ng; ++q) wantmsg[owner_of(row_starts, np, p->ghosts[q])]++;
    memcpy(wantmsg + np, p->ghosts, sizeof(int64_t) * (size_t)ng);
  }
  st = hc_agree(comm, st, "allocating the plan");
  if (st) goto done;
  st = hc_allgatherv(comm, wantmsg, (int64_t)sizeof(int64_t) * (np + ng), &asked_all, offs);
  st = hc_agree(comm, st, "exchanging the ghost requests");
  if (st) goto done;
  /* ---- my rows that another rank reads, by colour (ascending global row) ---- */
  int64_t nneed = 0;
  for (int q = 0; q < np; ++q)
    if (q != me) nneed += ((const int64_t *)((const char *)asked_all + offs[q]))[me];
  needed = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nneed > 0 ? nneed : 1));
  if (!needed) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    int64_t w = 0;
    for (int q = 0; q < np; ++q) {
      if (q == me) continue;
      const int64_t *m = (const int64_t *)((const char *)asked_all + offs[q]);
      int64_t        skip = 0;
      for (int r = 0; r < me; ++r) skip += m[r];
      memcpy(needed + w, m + np + skip, sizeof(int64_t) * (size_t)m[me]);
      w += m[me];
    }
    nneed = sort_unique(needed, nneed);
    if (nneed && (needed[0] < r0 || needed[nneed - 1] >= r1)) st = pmg_set_error(PMG_ERR_PLIB, __FILE__, __LINE__, "a rank asked for a row outside this block");
  }
  if (!st) {
    p->send_ptr  = (int64_t *)calloc((size_t)ncolors + 1, sizeof(int64_t));
    p->send_rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nneed > 0 ? nneed : 1));
    sendmsg      = (int64_t *)calloc((size_t)(ncolors + nneed), sizeof(int64_t));
    if (!p->send_ptr || !p->send_rows || !sendmsg) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  }
  if (!st) {
    for (int64_t i = 0; i < nneed && !st; ++i) {
      const int32_t c = colors_owned[needed[i] - r0];
      if (c < 0 || c >= ncolors) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "colour %d of row %lld outside [0, %d)", c, (long long)needed[i], ncolors);
      else sendmsg[c]++;
    }
    for (int c = 0; c < ncolors && !st; ++c) p->send_ptr[c + 1] = p->send_ptr[c] + sendmsg[c];
    if (!st) {
      int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncolors);
      if (!fill) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
      else {
        memcpy(fill, p->send_ptr, sizeof(int64_t) * (size_t)ncolors);
        for (int64_t i = 0; i < nneed; ++i) {
          const int64_t at       = fill[colors_owned[needed[i] - r0]]++;
          p->send_rows[at]       = (int32_t)(needed[i] - r0);
          sendmsg[ncolors + at]  = needed[i];
        }
        free(fill);
      }
    }
  }
  st = hc_agree(comm, st, "sorting the boundary rows by colour");
  if (st) goto done;
  st = hc_allgatherv(comm, sendmsg, (int64_t)sizeof(int64_t) * (ncolors + nneed), &lists_all, soffs);
  st = hc_agree(comm, st, "exchanging the boundary rows");
  if (st) goto done;
  /* ---- counts, receive lists ---- */
  p->counts    = (int64_t *)calloc((size_t)ncolors * (size_t)np, sizeof(int64_t));
  p->recv_ptr  = (int64_t *)calloc((size_t)ncolors + 1, sizeof(int64_t));
  p->recv_src  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ng > 0 ? ng : 1));
  p->recv_rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ng > 0 ? ng : 1));
  if (!p->counts || !p->recv_ptr || !p->recv_src || !p->recv_rows) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    for (int r = 0; r < np; ++r) {
      const int64_t *m = (const int64_t *)((const char *)lists_all + soffs[r]);
      for (int c = 0; c < ncolors; ++c) p->counts[(size_t)c * (size_t)np + r] = m[c];
    }
    int64_t w = 0;
    for (int c = 0; c < ncolors && !st; ++c) {
      int64_t off = 0, gq = 0; /* offset of rank r's block in the colour's buffer; first ghost owned by r */
      for (int r = 0; r < np && !st; ++r) {
        const int64_t  mine_from_r = wantmsg[r]; /* my ghosts owned by r: ghosts[gq .. gq + mine_from_r) */
        const int64_t *m = (const int64_t *)((const char *)lists_all + soffs[r]);
        const int64_t  len = m[c];
        if (r != me && mine_from_r && len) {
          int64_t skip = 0;
          for (int cc = 0; cc < c; ++cc) skip += m[cc];
          const int64_t *lst = m + ncolors + skip;
          for (int64_t q = gq; q < gq + mine_from_r; ++q) {
            const int64_t k = bsearch_i64(lst, len, p->ghosts[q]);
            if (k < 0) continue;
            if (w >= ng || off + k > INT32_MAX) {
              st = pmg_set_error(PMG_ERR_PLIB, __FILE__, __LINE__, "a ghost row has two colours or the gather buffer is too long");
              break;
            }
            p->recv_src[w]  = (int32_t)(off + k);
            p->recv_rows[w] = (int32_t)(nloc + q);
            ++w;
          }
        }
        off += len;
        gq += mine_from_r;
      }
      p->recv_ptr[c + 1] = w;
    }
    if (!st && w != ng) st = pmg_set_error(PMG_ERR_PLIB, __FILE__, __LINE__, "%lld of %lld ghost rows are refreshed: every ghost row must be owned by one rank and carry one colour", (long long)w, (long long)ng);
  }
  st = hc_agree(comm, st, "building the receive lists");
done:
  free(tmp), free(wantmsg), free(offs), free(soffs), free(needed), free(sendmsg), free(asked_all), free(lists_all);
  if (st) {
    pmg_rowblock_plan_destroy(&p);
    return st;
  }
  *out = p;
  return PMG_SUCCESS;
}

/* borrowed views (valid until the plan is destroyed); any pointer may be NULL */
pmg_status pmg_rowblock_plan_get(pmg_rowblock_plan p, int32_t *nghost, const int64_t **ghosts, const int64_t **send_ptr, const int32_t **send_rows, const int64_t **counts, const int64_t **recv_ptr, const int32_t **recv_src, const int32_t **recv_rows)
{
  PMG_CHECK(p, PMG_ERR_ARG_NULL, "null plan");
  if (nghost) *nghost = p->nghost;
  if (ghosts) *ghosts = p->ghosts;
  if (send_ptr) *send_ptr = p->send_ptr;
  if (send_rows) *send_rows = p->send_rows;
  if (counts) *counts = p->counts;
  if (recv_ptr) *recv_ptr = p->recv_ptr;
  if (recv_src) *recv_src = p->recv_src;
  if (recv_rows) *recv_rows = p->recv_rows;
  return PMG_SUCCESS;
}

/* A colouring is only usable if no owned row shares its colour with one of its columns -- owned or ghost (the ghost rows'
   colours are in the plan: a ghost row is received in the colour in which it changes).  Rows swept together would otherwise
   read each other's half-updated values and the chain would be wrong without any other symptom, so every constructor that
   takes rows + a plan checks it (cheap: one pass over the rows).  rp / ci: the owned rows with GLOBAL columns.  Local. */
static pmg_status rowblock_check_coloring(const pmg_rowblock_plan p, const int64_t *rp, const int64_t *ci, const int32_t *colors_owned)
{
  const int64_t nloc = p->nloc, r0 = p->row0;
  int32_t      *gcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->nghost > 0 ? p->nghost : 1));
  PMG_CHECK(gcol, PMG_ERR_MEM, "out of host memory");
  for (int32_t q = 0; q < p->nghost; ++q) gcol[q] = -1;
  for (int32_t c = 0; c < p->ncolors; ++c)
    for (int64_t w = p->recv_ptr[c]; w < p->recv_ptr[c + 1]; ++w) gcol[p->recv_rows[w] - nloc] = c;
  pmg_status st = PMG_SUCCESS;
  for (int64_t i = 0; i < nloc && !st; ++i)
    for (int64_t k = rp[i]; k < rp[i + 1] && !st; ++k) {
      const int64_t c = ci[k];
      if (c == r0 + i) continue;
      int32_t cc;
      if (c >= r0 && c < r0 + nloc) cc = colors_owned[c - r0];
      else {
        const int64_t q = bsearch_i64(p->ghosts, p->nghost, c);
        cc              = q >= 0 ? gcol[q] : -1;
      }
      if (cc == colors_owned[i]) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "not a distance-1 colouring: rows %lld and %lld are coupled and both have colour %d", (long long)(r0 + i), (long long)c, (int)cc);
    }
  free(gcol);
  return st;
}

/* the same check for a caller that holds rows and a plan (collective: a conflict on one rank is returned on every rank) */
pmg_status pmg_rowblock_check_coloring(const pmg_host_comm *comm, pmg_rowblock_plan plan, const int64_t *rowptr, const int64_t *colidx_global, const int32_t *colors_owned)
{
  PMG_CALL(hc_check(comm));
  PMG_CHECK(plan && rowptr && (plan->nloc == 0 || (colidx_global && colors_owned)), PMG_ERR_ARG_NULL, "null argument");
  return hc_agree(comm, rowblock_check_coloring(plan, rowptr, colidx_global, colors_owned), "checking the colouring");
}

/* ---------------------------------------------------------------------------------------------------- */
/* first-fit colouring of the global matrix, rank after rank                                              */
/* ---------------------------------------------------------------------------------------------------- */
/* The library's greedy rule (pmg_mcsor.c color_greedy: rows in ascending order, the smallest colour no already coloured
   neighbour carries) applied to the GLOBAL matrix: rank r colours its rows once the ranks below it have -- np rounds with
   one all-gather of a rank's colours each -- so the result is the colouring a single process computes, and with it the
   distributed chain is the single-process chain.  rp / ci: this rank's rows with global columns.  Collective. */
pmg_status pmg_rowblock_color_greedy(const pmg_host_comm *comm, const int64_t *row_starts, const int64_t *rp, const int64_t *ci, int32_t *colors_owned, int32_t *ncolors)
{
  PMG_CALL(hc_check(comm));
  PMG_CHECK(row_starts && rp && colors_owned && ncolors, PMG_ERR_ARG_NULL, "null argument");
  const int     np = comm->nranks, me = comm->rank;
  const int64_t r0 = row_starts[me], r1 = row_starts[me + 1], nloc = r1 - r0;
  pmg_status    st = PMG_SUCCESS;
  /* colours of the lower ranks' rows that my rows touch */
  int64_t nlow = 0;
  for (int64_t k = 0; k < rp[nloc]; ++k) nlow += ci[k] < r0;
  int64_t *low  = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nlow > 0 ? nlow : 1));
  int32_t *lcol = NULL, *mark = NULL;
  int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  if (!low || !offs) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    int64_t q = 0;
    for (int64_t k = 0; k < rp[nloc]; ++k)
      if (ci[k] < r0) low[q++] = ci[k];
    nlow = sort_unique(low, nlow);
    lcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nlow > 0 ? nlow : 1));
    if (!lcol) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  }
  st = hc_agree(comm, st, "preparing the colouring");
  int32_t nc = 0;
  for (int r = 0; r < np && !st; ++r) {
    if (r == me) {
      /* First fit gives a row a colour that is at most the number of its neighbours, whatever colours those carry: the
         mark array is sized by the largest row, NOT by the number of local rows (round 3 sized it nloc + 2 and dropped
         neighbour colours above nloc -- a block of one or two rows beside a clique of a lower rank could then repeat a
         neighbour's colour: advisor finding, reproduced by tests/test_rowblock_c.py::test_tiny_block_beside_a_clique) */
      int64_t maxdeg = 0;
      for (int64_t i = 0; i < nloc; ++i)
        if (rp[i + 1] - rp[i] > maxdeg) maxdeg = rp[i + 1] - rp[i];
      mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)(maxdeg + 2));
      if (!mark) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
      for (int64_t i = 0; i <= maxdeg + 1 && !st; ++i) mark[i] = -1;
      for (int64_t i = 0; i < nloc && !st; ++i) {
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
          const int64_t c = ci[k];
          int32_t       cc = -1;
          if (c < r0) cc = lcol[bsearch_i64(low, nlow, c)];
          else if (c < r0 + i) cc = colors_owned[c - r0];
          if (cc >= 0 && cc <= maxdeg) mark[cc] = (int32_t)i; /* a colour above the row's degree cannot be the smallest free one */
        }
        int32_t col = 0;
        while (col <= maxdeg && mark[col] == (int32_t)i) ++col;
        colors_owned[i] = col;
        if (col + 1 > nc) nc = col + 1;
      }
      free(mark);
      mark = NULL;
    }
    void *all = NULL;
    if (!st) st = hc_allgatherv(comm, colors_owned, r == me ? (int64_t)sizeof(int32_t) * nloc : 0, &all, offs);
    st = hc_agree(comm, st, "colouring a row block");
    if (!st && r < me) { /* pick the colours of my lower neighbours owned by r */
      const int32_t *cr = (const int32_t *)((const char *)all + offs[r]);
      for (int64_t q = 0; q < nlow; ++q)
        if (low[q] >= row_starts[r] && low[q] < row_starts[r + 1]) lcol[q] = cr[low[q] - row_starts[r]];
    }
    free(all);
  }
  if (!st) { /* number of colours: the maximum over the ranks */
    int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)np);
    if (!all) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st) st = hc_allgather(comm, &nc, sizeof nc, all);
    for (int r = 0; r < np && !st; ++r)
      if (all[r] > nc) nc = all[r];
    free(all);
  }
  free(low), free(lcol), free(offs);
  PMG_CALL(st);
  *ncolors = nc;
  return PMG_SUCCESS;
}

/* PMG_COLORING_ITERATED (pmg_mcsor.c color_iterated) on the GLOBAL matrix: first-fit as above, then first-fit once more with
   the rows visited class by class, the last class first.  The rows of one first-fit class are not coupled, so every rank colours
   its rows of the class at the same time; between two classes the ranks exchange what they assigned (one all-gather of the owned
   colour arrays per class) -- the colouring one process computes, whatever the number of ranks.  Collective. */
pmg_status pmg_rowblock_color_iterated(const pmg_host_comm *comm, const int64_t *row_starts, const int64_t *rp, const int64_t *ci, int32_t *colors_owned, int32_t *ncolors)
{
  PMG_CALL(pmg_rowblock_color_greedy(comm, row_starts, rp, ci, colors_owned, ncolors));
  const int     np = comm->nranks, me = comm->rank, nc0 = *ncolors;
  const int64_t r0 = row_starts[me], r1 = row_starts[me + 1], nloc = r1 - r0;
  if (nc0 <= 2) return PMG_SUCCESS;
  pmg_status st = PMG_SUCCESS;
  int64_t    ngh = 0, maxdeg = 0;
  for (int64_t k = 0; k < rp[nloc]; ++k) ngh += ci[k] < r0 || ci[k] >= r1;
  for (int64_t i = 0; i < nloc; ++i)
    if (rp[i + 1] - rp[i] > maxdeg) maxdeg = rp[i + 1] - rp[i];
  int64_t *gh   = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ngh > 0 ? ngh : 1));
  int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  int32_t *newc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nloc > 0 ? nloc : 1));
  int32_t *mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)(maxdeg + 2));
  int32_t *gcol = NULL, *gown = NULL;
  if (!gh || !offs || !newc || !mark) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    int64_t q = 0;
    for (int64_t k = 0; k < rp[nloc]; ++k)
      if (ci[k] < r0 || ci[k] >= r1) gh[q++] = ci[k];
    ngh  = sort_unique(gh, ngh);
    gcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ngh > 0 ? ngh : 1));
    gown = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ngh > 0 ? ngh : 1));
    if (!gcol || !gown) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    for (int64_t g = 0, r = 0; g < ngh && !st; ++g) { /* the ghosts are sorted: their owners ascend */
      while (r + 1 < np && gh[g] >= row_starts[r + 1]) ++r;
      gown[g] = (int32_t)r;
      gcol[g] = -1;
    }
    for (int64_t i = 0; i < nloc; ++i) newc[i] = -1;
    for (int64_t i = 0; i <= maxdeg + 1; ++i) mark[i] = -1;
  }
  st = hc_agree(comm, st, "preparing the second colouring round");
  int32_t nc = 0;
  for (int32_t cls = nc0 - 1; cls >= 0 && !st; --cls) {
    for (int64_t i = 0; i < nloc; ++i) {
      if (colors_owned[i] != cls) continue;
      for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
        const int64_t c  = ci[k];
        const int32_t cc = c == r0 + i ? -1 : (c >= r0 && c < r1 ? newc[c - r0] : gcol[bsearch_i64(gh, ngh, c)]);
        if (cc >= 0 && cc <= maxdeg) mark[cc] = (int32_t)i; /* a colour above the row's degree cannot be the smallest free one */
      }
      int32_t col = 0;
      while (col <= maxdeg && mark[col] == (int32_t)i) ++col;
      newc[i] = col;
      if (col + 1 > nc) nc = col + 1;
    }
    void *all = NULL;
    st        = hc_allgatherv(comm, newc, (int64_t)sizeof(int32_t) * nloc, &all, offs);
    st        = hc_agree(comm, st, "exchanging a colour class");
    for (int64_t g = 0; g < ngh && !st; ++g) gcol[g] = ((const int32_t *)((const char *)all + offs[gown[g]]))[gh[g] - row_starts[gown[g]]];
    free(all);
  }
  if (!st) {
    memcpy(colors_owned, newc, sizeof(int32_t) * (size_t)nloc);
    int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)np);
    if (!all) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st) st = hc_allgather(comm, &nc, sizeof nc, all);
    for (int r = 0; r < np && !st; ++r)
      if (all[r] > nc) nc = all[r];
    free(all);
  }
  free(gh), free(offs), free(newc), free(mark), free(gcol), free(gown);
  PMG_CALL(st);
  *ncolors = nc;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* hierarchy distributed by row blocks                                                                    */
/* ---------------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t  nr;
  int64_t *rp, *ci;
  double  *v;
} gcsr; /* rows with global (64-bit) columns, owned */

static void gcsr_free(gcsr *m)
{
  free(m->rp), free(m->ci), free(m->v);
  memset(m, 0, sizeof *m);
}

typedef struct {
  int64_t   n, row0, nloc;       /* global rows, this rank's block */
  int64_t  *starts;              /* nranks + 1 */
  gcsr      A, P;                /* the caller's rows (copied, widened to 64-bit columns) */
  int       have_A, have_P;
  int32_t   ncolors;
  int32_t  *colors;              /* owned rows */
  /* built */
  pmg_rowblock_plan plan;
  int32_t  *l_rp, *l_ci;         /* local operator: owned rows + identity ghost rows */
  double   *l_v;
  int32_t  *P_rp, *P_ci, *R_rp, *R_ci;
  double   *P_v, *R_v;
  int32_t   R_nrows, ncoarse_local, nlocal;
  /* replicated level: the whole matrices */
  int32_t  *g_rp, *g_ci, *gP_rp, *gP_ci;
  double   *g_v, *gP_v;
} rbh_level;

struct pmg_rbh_s {
  pmg_host_comm comm;
  int32_t       nlevels, fold, built;
  int           coloring; /* rule of the levels without a caller's colouring: PMG_COLORING_GREEDY (0) or PMG_COLORING_ITERATED */
  int64_t       replicate_below;
  rbh_level    *lv;
};

pmg_status pmg_rbh_create(const pmg_host_comm *comm, int32_t nlevels, int64_t replicate_below, pmg_rbh *out)
{
  PMG_CALL(hc_check(comm));
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  PMG_CHECK(nlevels >= 2 && nlevels <= 64, PMG_ERR_ARG_OUTOFRANGE, "levels = %d (2..64)", nlevels);
  pmg_rbh h = (pmg_rbh)calloc(1, sizeof *h);
  PMG_CHECK(h, PMG_ERR_MEM, "out of host memory");
  h->comm            = *comm;
  h->nlevels         = nlevels;
  h->replicate_below = replicate_below >= 0 ? replicate_below : 50000;
  h->lv              = (rbh_level *)calloc((size_t)nlevels, sizeof(rbh_level));
  if (!h->lv) {
    free(h);
    PMG_FAIL(PMG_ERR_MEM, "out of host memory");
  }
  *out = h;
  return PMG_SUCCESS;
}

void pmg_rbh_destroy(pmg_rbh *hp)
{
  if (!hp || !*hp) return;
  pmg_rbh h = *hp;
  for (int l = 0; l < h->nlevels; ++l) {
    rbh_level *L = &h->lv[l];
    gcsr_free(&L->A), gcsr_free(&L->P);
    pmg_rowblock_plan_destroy(&L->plan);
    free(L->starts), free(L->colors), free(L->l_rp), free(L->l_ci), free(L->l_v), free(L->P_rp), free(L->P_ci), free(L->P_v), free(L->R_rp), free(L->R_ci), free(L->R_v);
    free(L->g_rp), free(L->g_ci), free(L->g_v), free(L->gP_rp), free(L->gP_ci), free(L->gP_v);
  }
  free(h->lv);
  free(h);
  *hp = NULL;
}

static pmg_status gcsr_copy(gcsr *m, int64_t nr, const void *rp, const void *ci, const double *v, int w)
{
  gcsr_free(m);
  PMG_CHECK(w == 32 || w == 64, PMG_ERR_ARG_OUTOFRANGE, "idx_width = %d (32 | 64)", w);
  PMG_CHECK(nr >= 0 && (nr == 0 || (rp && ci && v)), PMG_ERR_ARG_NULL, "null CSR array");
  const int64_t nnz = nr ? idx_at(rp, nr, w) : 0;
  m->nr = nr;
  m->rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nr + 1));
  m->ci = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
  m->v  = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  PMG_CHECK(m->rp && m->ci && m->v, PMG_ERR_MEM, "out of host memory");
  m->rp[0] = 0;
  for (int64_t r = 0; r < nr; ++r) m->rp[r + 1] = idx_at(rp, r + 1, w);
  for (int64_t k = 0; k < nnz; ++k) m->ci[k] = idx_at(ci, k, w), m->v[k] = v[k];
  return PMG_SUCCESS;
}

/* this rank's rows [row0, row0 + nloc) of the level's operator (n_global rows in all) with GLOBAL column indices, entries
   in the order the chain shall visit them (pmg_rowblock_merge_mpiaij); copied */
pmg_status pmg_rbh_set_level_operator(pmg_rbh h, int32_t level, int64_t n_global, int64_t row0, int64_t nloc, const void *rowptr, const void *colidx_global, const double *vals, int idx_width)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->built, PMG_ERR_ARG_WRONGSTATE, "the hierarchy is already built");
  PMG_CHECK(level >= 0 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  PMG_CHECK(n_global >= 1 && row0 >= 0 && nloc >= 0 && row0 + nloc <= n_global && nloc < INT32_MAX, PMG_ERR_ARG_OUTOFRANGE, "rows [%lld, %lld) of %lld", (long long)row0, (long long)(row0 + nloc), (long long)n_global);
  rbh_level *L = &h->lv[level];
  PMG_CALL(gcsr_copy(&L->A, nloc, rowptr, colidx_global, vals, idx_width));
  L->n = n_global, L->row0 = row0, L->nloc = nloc, L->have_A = 1;
  return PMG_SUCCESS;
}

/* this rank's rows of P_level (the rows it owns on `level`), columns = global rows of level - 1; copied */
pmg_status pmg_rbh_set_level_interpolation(pmg_rbh h, int32_t level, int64_t nloc_rows, const void *rowptr, const void *colidx_global, const double *vals, int idx_width)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->built, PMG_ERR_ARG_WRONGSTATE, "the hierarchy is already built");
  PMG_CHECK(level >= 1 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  rbh_level *L = &h->lv[level];
  PMG_CALL(gcsr_copy(&L->P, nloc_rows, rowptr, colidx_global, vals, idx_width));
  L->have_P = 1;
  return PMG_SUCCESS;
}

/* optional: the caller's colouring of its rows of `level` (e.g. PETSc's MatColoring as the reference applies it,
   src/mc_sor.c:383-395); must be a valid distance-1 colouring of the global matrix with the same ncolors on every rank.
   Default: the library's first-fit rule on the global matrix (pmg_rowblock_color_greedy). */
/* rule of every level that gets no colouring of the caller's: PMG_COLORING_GREEDY (default) or PMG_COLORING_ITERATED -- row-block
   levels through pmg_rowblock_color_greedy / _iterated, the replicated levels below the fold through pmg_mgmc_set_coloring */
pmg_status pmg_rbh_set_coloring(pmg_rbh h, int rule)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->built, PMG_ERR_ARG_WRONGSTATE, "the hierarchy is already built");
  PMG_CHECK(rule == PMG_COLORING_GREEDY || rule == PMG_COLORING_ITERATED, PMG_ERR_ARG_OUTOFRANGE, "colouring rule %d: greedy or iterated expected", rule);
  h->coloring = rule;
  return PMG_SUCCESS;
}

pmg_status pmg_rbh_set_level_coloring(pmg_rbh h, int32_t level, int32_t ncolors, const int32_t *colors_owned)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->built, PMG_ERR_ARG_WRONGSTATE, "the hierarchy is already built");
  PMG_CHECK(level >= 1 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  rbh_level *L = &h->lv[level];
  PMG_CHECK(L->have_A, PMG_ERR_ARG_WRONGSTATE, "set the operator of level %d first", level);
  PMG_CHECK(ncolors >= 1 && (L->nloc == 0 || colors_owned), PMG_ERR_ARG_NULL, "null colouring");
  free(L->colors);
  L->colors = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L->nloc > 0 ? L->nloc : 1));
  PMG_CHECK(L->colors, PMG_ERR_MEM, "out of host memory");
  if (L->nloc) memcpy(L->colors, colors_owned, sizeof(int32_t) * (size_t)L->nloc);
  L->ncolors = ncolors;
  return PMG_SUCCESS;
}

/* the whole matrix on every rank from its row blocks (rows keep the caller's entry order); 32-bit output */
static pmg_status gather_rows(const pmg_host_comm *c, const gcsr *m, int64_t ncols_max, int32_t **rp_out, int32_t **ci_out, double **v_out, int64_t *nrows_out)
{
  const int  np   = c->nranks;
  int64_t   *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  void      *lens = NULL, *cols = NULL, *vals = NULL;
  pmg_status st   = offs ? PMG_SUCCESS : pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  int64_t   *rl   = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m->nr > 0 ? m->nr : 1));
  if (!st && !rl) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  for (int64_t r = 0; r < m->nr && !st; ++r) rl[r] = m->rp[r + 1] - m->rp[r];
  const int64_t nnz = m->nr ? m->rp[m->nr] : 0;
  int64_t       nrows = 0, tot = 0;
  if (!st) st = hc_allgatherv(c, rl, (int64_t)sizeof(int64_t) * m->nr, &lens, offs);
  if (!st) nrows = offs[np] / (int64_t)sizeof(int64_t);
  if (!st) st = hc_allgatherv(c, m->ci, (int64_t)sizeof(int64_t) * nnz, &cols, offs);
  if (!st) tot = offs[np] / (int64_t)sizeof(int64_t);
  if (!st) st = hc_allgatherv(c, m->v, (int64_t)sizeof(double) * nnz, &vals, offs);
  st = hc_agree(c, st, "gathering a replicated level");
  int32_t *rp = NULL, *ci = NULL;
  if (!st && (nrows >= INT32_MAX || tot >= INT32_MAX)) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "a replicated level must fit 32-bit indices");
  if (!st) {
    rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nrows + 1));
    ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
    if (!rp || !ci) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  }
  if (!st) {
    rp[0] = 0;
    for (int64_t r = 0; r < nrows; ++r) rp[r + 1] = rp[r] + (int32_t)((const int64_t *)lens)[r];
    for (int64_t k = 0; k < tot && !st; ++k) {
      const int64_t cc = ((const int64_t *)cols)[k];
      if (cc < 0 || cc >= ncols_max) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "column %lld outside [0, %lld)", (long long)cc, (long long)ncols_max);
      ci[k] = (int32_t)cc;
    }
  }
  free(offs), free(rl), free(lens), free(cols);
  if (st) {
    free(rp), free(ci), free(vals);
    return st;
  }
  *rp_out = rp, *ci_out = ci, *v_out = (double *)vals, *nrows_out = nrows;
  return PMG_SUCCESS;
}

/* sort the entries of every row by column (level 0: the dense factorisation takes a canonical matrix) */
typedef struct {
  int32_t c;
  double  v;
} cv_pair;
static int cmp_cv(const void *a, const void *b) { return ((const cv_pair *)a)->c - ((const cv_pair *)b)->c; }
static pmg_status sort_rows(int64_t nr, const int32_t *rp, int32_t *ci, double *v)
{
  int32_t w = 0;
  for (int64_t r = 0; r < nr; ++r)
    if (rp[r + 1] - rp[r] > w) w = rp[r + 1] - rp[r];
  cv_pair *t = (cv_pair *)malloc(sizeof(cv_pair) * (size_t)(w > 0 ? w : 1));
  PMG_CHECK(t, PMG_ERR_MEM, "out of host memory");
  for (int64_t r = 0; r < nr; ++r) {
    const int32_t len = rp[r + 1] - rp[r];
    for (int32_t k = 0; k < len; ++k) t[k].c = ci[rp[r] + k], t[k].v = v[rp[r] + k];
    qsort(t, (size_t)len, sizeof(cv_pair), cmp_cv);
    for (int32_t k = 0; k < len; ++k) ci[rp[r] + k] = t[k].c, v[rp[r] + k] = t[k].v;
  }
  free(t);
  return PMG_SUCCESS;
}

/* the rows of P_l^T this rank owns on level l - 1 (block [c0, c1) of THAT level), columns = global fine rows, entries by
   ascending fine row: every rank contributes the triples of its rows of P_l whose column falls into another rank's block */
typedef struct {
  int64_t row, col;
  double  v;
} triple;

static pmg_status build_restriction(const pmg_host_comm *c, const gcsr *P, int64_t fine_row0, const int64_t *cstarts, gcsr *R)
{
  const int     np = c->nranks, me = c->rank;
  const int64_t c0 = cstarts[me], c1 = cstarts[me + 1], ncoarse = cstarts[np], nnz = P->nr ? P->rp[P->nr] : 0;
  pmg_status    st = PMG_SUCCESS;
  int64_t       noff = 0;
  for (int64_t k = 0; k < nnz && !st; ++k) {
    if (P->ci[k] < 0 || P->ci[k] >= ncoarse) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "interpolation column %lld outside [0, %lld)", (long long)P->ci[k], (long long)ncoarse);
    noff += P->ci[k] < c0 || P->ci[k] >= c1;
  }
  triple  *mine = (triple *)malloc(sizeof(triple) * (size_t)(noff > 0 ? noff : 1));
  int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  void    *all  = NULL;
  if (!st && (!mine || !offs)) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  if (!st) {
    int64_t q = 0;
    for (int64_t r = 0; r < P->nr; ++r)
      for (int64_t k = P->rp[r]; k < P->rp[r + 1]; ++k)
        if (P->ci[k] < c0 || P->ci[k] >= c1) mine[q].row = fine_row0 + r, mine[q].col = P->ci[k], mine[q].v = P->v[k], ++q;
  }
  st = hc_agree(c, st, "collecting the interpolation entries of other ranks' columns");
  if (!st) st = hc_allgatherv(c, mine, (int64_t)sizeof(triple) * noff, &all, offs);
  st = hc_agree(c, st, "exchanging the interpolation entries");
  free(mine);
  if (st) {
    free(offs), free(all);
    return st;
  }
  /* count per coarse row, then fill in rank order: ranks own ascending blocks of fine rows and list them in row order */
  const int64_t nr = c1 - c0;
  R->nr = nr;
  R->rp = (int64_t *)calloc((size_t)nr + 1, sizeof(int64_t));
  if (!R->rp) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  for (int pass = 0; pass < 2 && !st; ++pass) {
    int64_t *fill = NULL;
    if (pass == 1) {
      for (int64_t r = 0; r < nr; ++r) R->rp[r + 1] += R->rp[r];
      const int64_t tot = R->rp[nr];
      R->ci = (int64_t *)malloc(sizeof(int64_t) * (size_t)(tot > 0 ? tot : 1));
      R->v  = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
      fill  = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nr > 0 ? nr : 1));
      if (!R->ci || !R->v || !fill) {
        free(fill);
        st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
        break;
      }
      memcpy(fill, R->rp, sizeof(int64_t) * (size_t)nr);
    }
    for (int r = 0; r < np; ++r) {
      if (r == me) { /* my own rows of P: the entries whose column I own */
        for (int64_t i = 0; i < P->nr; ++i)
          for (int64_t k = P->rp[i]; k < P->rp[i + 1]; ++k) {
            const int64_t cc = P->ci[k];
            if (cc < c0 || cc >= c1) continue;
            if (pass == 0) R->rp[cc - c0 + 1]++;
            else {
              const int64_t at = fill[cc - c0]++;
              R->ci[at] = fine_row0 + i, R->v[at] = P->v[k];
            }
          }
        continue;
      }
      const triple *t  = (const triple *)((const char *)all + offs[r]);
      const int64_t nt = (offs[r + 1] - offs[r]) / (int64_t)sizeof(triple);
      for (int64_t q = 0; q < nt; ++q) {
        if (t[q].col < c0 || t[q].col >= c1) continue;
        if (pass == 0) R->rp[t[q].col - c0 + 1]++;
        else {
          const int64_t at = fill[t[q].col - c0]++;
          R->ci[at] = t[q].row, R->v[at] = t[q].v;
        }
      }
    }
    free(fill);
  }
  free(offs), free(all);
  if (st) gcsr_free(R);
  return st;
}

/* global row -> local row of this rank on a row-block level (owned: g - row0; ghost: nloc + its rank among the ghosts) */
static int64_t local_of(const rbh_level *L, const int64_t *ghosts, int32_t nghost, int64_t g)
{
  if (g >= L->row0 && g < L->row0 + L->nloc) return g - L->row0;
  const int64_t q = bsearch_i64(ghosts, nghost, g);
  return q < 0 ? -1 : L->nloc + q;
}

/* Collective: row blocks of every level, the fold (levels with at most replicate_below rows -- at least the coarsest -- are
   replicated: gathered whole on every rank), colourings, restriction rows, ghost plans, local matrices. */
pmg_status pmg_rbh_build(pmg_rbh h)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(!h->built, PMG_ERR_ARG_WRONGSTATE, "the hierarchy is already built");
  const pmg_host_comm *c  = &h->comm;
  const int            np = c->nranks, me = c->rank, L = h->nlevels;
  pmg_status           st = PMG_SUCCESS;
  for (int l = 0; l < L && !st; ++l)
    if (!h->lv[l].have_A || (l >= 1 && !h->lv[l].have_P)) st = pmg_set_error(PMG_ERR_ARG_WRONGSTATE, __FILE__, __LINE__, "operator or interpolation of level %d missing", l);
  for (int l = 1; l < L && !st; ++l)
    if (h->lv[l].P.nr != h->lv[l].nloc) st = pmg_set_error(PMG_ERR_ARG_SIZ, __FILE__, __LINE__, "level %d: %lld rows of P for %lld owned rows", l, (long long)h->lv[l].P.nr, (long long)h->lv[l].nloc);
  st = hc_agree(c, st, "checking the levels");
  /* ---- row blocks of every level ---- */
  for (int l = 0; l < L && !st; ++l) {
    rbh_level *Lv = &h->lv[l];
    Lv->starts    = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
    if (!Lv->starts) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st) st = hc_allgather(c, &Lv->row0, sizeof(int64_t), Lv->starts);
    if (!st) {
      Lv->starts[np] = Lv->n;
      if (Lv->starts[0] != 0 || Lv->starts[me + 1] != Lv->row0 + Lv->nloc) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "level %d: the row blocks must tile the rows in rank order", l);
    }
    st = hc_agree(c, st, "exchanging the ownership ranges");
  }
  if (st) return st;
  int fold = 1;
  while (fold < L - 1 && h->lv[fold].n <= h->replicate_below) ++fold;
  h->fold = fold;
  /* ---- replicated levels: the whole operator (and interpolation) on every rank ---- */
  for (int l = 0; l < fold && !st; ++l) {
    rbh_level *Lv = &h->lv[l];
    int64_t    nr = 0;
    st = gather_rows(c, &Lv->A, Lv->n, &Lv->g_rp, &Lv->g_ci, &Lv->g_v, &nr);
    if (!st && nr != Lv->n) st = pmg_set_error(PMG_ERR_ARG_SIZ, __FILE__, __LINE__, "level %d: %lld rows gathered, %lld expected", l, (long long)nr, (long long)Lv->n);
    if (!st && l == 0) st = sort_rows(nr, Lv->g_rp, Lv->g_ci, Lv->g_v);
    if (!st && l >= 1) {
      st = gather_rows(c, &Lv->P, h->lv[l - 1].n, &Lv->gP_rp, &Lv->gP_ci, &Lv->gP_v, &nr);
      if (!st && nr != Lv->n) st = pmg_set_error(PMG_ERR_ARG_SIZ, __FILE__, __LINE__, "level %d: %lld rows of P gathered, %lld expected", l, (long long)nr, (long long)Lv->n);
    }
    st = hc_agree(c, st, "replicating a small level");
  }
  /* ---- row-block levels ---- */
  gcsr *R = (gcsr *)calloc((size_t)L, sizeof(gcsr)); /* R[l]: my rows of P_l^T (block of level l-1), columns = global rows of level l */
  if (!st && !R) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  for (int l = fold; l < L && !st; ++l) {
    rbh_level *Lv = &h->lv[l];
    if (!Lv->colors) {
      Lv->colors = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Lv->nloc > 0 ? Lv->nloc : 1));
      if (!Lv->colors) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
      st = hc_agree(c, st, "allocating the colouring");
      if (!st) st = (h->coloring == PMG_COLORING_ITERATED ? pmg_rowblock_color_iterated : pmg_rowblock_color_greedy)(c, Lv->starts, Lv->A.rp, Lv->A.ci, Lv->colors, &Lv->ncolors);
    } else { /* the caller's: every rank must name the same number of colours */
      int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)np);
      if (!all) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
      if (!st) st = hc_allgather(c, &Lv->ncolors, sizeof(int32_t), all);
      for (int r = 0; r < np && !st; ++r)
        if (all[r] != Lv->ncolors) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "level %d: rank %d names %d colours, this rank %d", l, r, all[r], Lv->ncolors);
      free(all);
      st = hc_agree(c, st, "checking the caller's colouring");
    }
    if (!st) st = build_restriction(c, &Lv->P, Lv->row0, h->lv[l - 1].starts, &R[l]);
  }
  for (int l = fold; l < L && !st; ++l) {
    rbh_level    *Lv = &h->lv[l];
    const int64_t nx = R[l].rp[R[l].nr] + (l + 1 < L ? h->lv[l + 1].P.rp[h->lv[l + 1].P.nr] : 0);
    int64_t      *extra = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nx > 0 ? nx : 1));
    if (!extra) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st) {
      memcpy(extra, R[l].ci, sizeof(int64_t) * (size_t)R[l].rp[R[l].nr]);
      if (l + 1 < L) memcpy(extra + R[l].rp[R[l].nr], h->lv[l + 1].P.ci, sizeof(int64_t) * (size_t)h->lv[l + 1].P.rp[h->lv[l + 1].P.nr]);
    }
    st = hc_agree(c, st, "collecting the rows the transfers read");
    if (!st) st = pmg_rowblock_plan_create(c, Lv->starts, Lv->A.rp[Lv->nloc], Lv->A.ci, nx, extra, Lv->ncolors, Lv->colors, &Lv->plan);
    if (!st) st = hc_agree(c, rowblock_check_coloring(Lv->plan, Lv->A.rp, Lv->A.ci, Lv->colors), "checking the colouring of a level");
    free(extra);
  }
  /* ---- local matrices ---- */
  for (int l = fold; l < L && !st; ++l) {
    rbh_level     *Lv = &h->lv[l];
    const int64_t *gh = Lv->plan->ghosts;
    const int32_t  ng = Lv->plan->nghost;
    const int64_t  nnz = Lv->A.rp[Lv->nloc];
    Lv->nlocal         = (int32_t)(Lv->nloc + ng);
    if (nnz + ng >= INT32_MAX) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "level %d: too many entries for 32-bit indices", l);
    if (!st) {
      Lv->l_rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Lv->nlocal + 1));
      Lv->l_ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz + ng > 0 ? nnz + ng : 1));
      Lv->l_v  = (double *)malloc(sizeof(double) * (size_t)(nnz + ng > 0 ? nnz + ng : 1));
      if (!Lv->l_rp || !Lv->l_ci || !Lv->l_v) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    }
    if (!st) {
      for (int64_t r = 0; r <= Lv->nloc; ++r) Lv->l_rp[r] = (int32_t)Lv->A.rp[r];
      for (int64_t k = 0; k < nnz && !st; ++k) {
        const int64_t lo = local_of(Lv, gh, ng, Lv->A.ci[k]);
        if (lo < 0) st = pmg_set_error(PMG_ERR_PLIB, __FILE__, __LINE__, "level %d: column %lld is neither owned nor a ghost", l, (long long)Lv->A.ci[k]);
        Lv->l_ci[k] = (int32_t)lo, Lv->l_v[k] = Lv->A.v[k];
      }
      for (int32_t q = 0; q < ng; ++q) Lv->l_rp[Lv->nloc + q + 1] = (int32_t)(nnz + q + 1), Lv->l_ci[nnz + q] = (int32_t)(Lv->nloc + q), Lv->l_v[nnz + q] = 1.0;
    }
    /* my rows of P_l, columns -> local numbering of level l-1 (global where that level is replicated) */
    const rbh_level *Cc  = &h->lv[l - 1];
    const int        rep = l == fold;
    const int64_t    pnz = Lv->P.rp[Lv->P.nr], rnz = R[l].rp[R[l].nr];
    if (!st && (pnz >= INT32_MAX || rnz >= INT32_MAX)) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "level %d: too many transfer entries for 32-bit indices", l);
    if (!st) {
      Lv->P_rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Lv->P.nr + 1));
      Lv->P_ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pnz > 0 ? pnz : 1));
      Lv->P_v  = (double *)malloc(sizeof(double) * (size_t)(pnz > 0 ? pnz : 1));
      Lv->R_rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R[l].nr + 1));
      Lv->R_ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rnz > 0 ? rnz : 1));
      Lv->R_v  = (double *)malloc(sizeof(double) * (size_t)(rnz > 0 ? rnz : 1));
      if (!Lv->P_rp || !Lv->P_ci || !Lv->P_v || !Lv->R_rp || !Lv->R_ci || !Lv->R_v) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    }
    if (!st) {
      for (int64_t r = 0; r <= Lv->P.nr; ++r) Lv->P_rp[r] = (int32_t)Lv->P.rp[r];
      for (int64_t k = 0; k < pnz && !st; ++k) {
        const int64_t lo = rep ? Lv->P.ci[k] : local_of(Cc, Cc->plan->ghosts, Cc->plan->nghost, Lv->P.ci[k]);
        if (lo < 0) st = pmg_set_error(PMG_ERR_PLIB, __FILE__, __LINE__, "level %d: the interpolation reads coarse row %lld, neither owned nor a ghost", l, (long long)Lv->P.ci[k]);
        Lv->P_ci[k] = (int32_t)lo, Lv->P_v[k] = Lv->P.v[k];
      }
      Lv->R_nrows = (int32_t)R[l].nr;
      for (int64_t r = 0; r <= R[l].nr; ++r) Lv->R_rp[r] = (int32_t)R[l].rp[r];
      for (int64_t k = 0; k < rnz && !st; ++k) {
        const int64_t lo = local_of(Lv, gh, ng, R[l].ci[k]);
        if (lo < 0) st = pmg_set_error(PMG_ERR_PLIB, __FILE__, __LINE__, "level %d: the restriction reads fine row %lld, neither owned nor a ghost", l, (long long)R[l].ci[k]);
        Lv->R_ci[k] = (int32_t)lo, Lv->R_v[k] = R[l].v[k];
      }
      Lv->ncoarse_local = rep ? (int32_t)Cc->n : Cc->nlocal;
    }
    st = hc_agree(c, st, "building the local matrices");
  }
  if (R)
    for (int l = 0; l < L; ++l) gcsr_free(&R[l]);
  free(R);
  PMG_CALL(st);
  h->built = 1;
  return PMG_SUCCESS;
}

pmg_status pmg_rbh_get_info(pmg_rbh h, int32_t *nlevels, int32_t *fold)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  PMG_CHECK(h->built, PMG_ERR_ARG_WRONGSTATE, "call pmg_rbh_build first");
  if (nlevels) *nlevels = h->nlevels;
  if (fold) *fold = h->fold;
  return PMG_SUCCESS;
}

/* borrowed views of one level (valid until the handle is destroyed) */
pmg_status pmg_rbh_get_level(pmg_rbh h, int32_t level, pmg_rbh_level_view *v)
{
  PMG_CHECK(h && v, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->built, PMG_ERR_ARG_WRONGSTATE, "call pmg_rbh_build first");
  PMG_CHECK(level >= 0 && level < h->nlevels, PMG_ERR_ARG_OUTOFRANGE, "level %d", level);
  const rbh_level *L = &h->lv[level];
  memset(v, 0, sizeof *v);
  v->n_global = L->n, v->row0 = L->row0, v->nowned = (int32_t)L->nloc, v->starts = L->starts;
  v->replicated = level < h->fold;
  if (v->replicated) {
    v->nlocal = (int32_t)L->n;
    v->rp = L->g_rp, v->ci = L->g_ci, v->v = L->g_v;
    v->P_rp = L->gP_rp, v->P_ci = L->gP_ci, v->P_v = L->gP_v;
    v->P_nrows = level >= 1 ? (int32_t)L->n : 0;
    v->ncoarse_local = level >= 1 ? (int32_t)h->lv[level - 1].n : 0;
    return PMG_SUCCESS;
  }
  v->nlocal = L->nlocal, v->rp = L->l_rp, v->ci = L->l_ci, v->v = L->l_v;
  v->ncolors = L->ncolors, v->colors = L->colors;
  v->P_nrows = (int32_t)L->nloc, v->P_rp = L->P_rp, v->P_ci = L->P_ci, v->P_v = L->P_v;
  v->R_nrows = L->R_nrows, v->R_rp = L->R_rp, v->R_ci = L->R_ci, v->R_v = L->R_v;
  v->ncoarse_local = L->ncoarse_local;
  return pmg_rowblock_plan_get(L->plan, &v->nghost, &v->ghosts, &v->send_ptr, &v->send_rows, &v->counts, &v->recv_ptr, &v->recv_src, &v->recv_rows);
}

/* PCGAMGMC on the hierarchy: pmg_mgmc_create_hierarchy + every pmg_mgmc_set_level_* call (what parmgmc_amd.dist.DistAIJMGMC
   does from Python).  `transport`: any pmg_dist object of the same ranks (pmg_dist_create_comm).  The arrays stay borrowed
   from `h` until pmg_mgmc_setup: destroy `h` after it. */
pmg_status pmg_rbh_create_mgmc(pmg_rbh h, pmg_dist transport, pmg_mgmc *out)
{
  PMG_CHECK(h && transport && out, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(h->built, PMG_ERR_ARG_WRONGSTATE, "call pmg_rbh_build first");
  *out        = NULL;
  pmg_mgmc mg = NULL;
  PMG_CALL(pmg_mgmc_create_hierarchy(h->nlevels, &mg));
  PMG_CALL(pmg_mgmc_set_coloring(mg, h->coloring)); /* the replicated levels below the fold follow the hierarchy's rule */
  pmg_status st = PMG_SUCCESS;
  for (int l = 0; l < h->nlevels && !st; ++l) {
    pmg_rbh_level_view v;
    st = pmg_rbh_get_level(h, l, &v);
    if (!st) st = pmg_mgmc_set_level_operator(mg, l, v.nlocal, v.rp, v.ci, v.v);
    if (!st && l == 0) st = pmg_mgmc_set_rowblock_transport(mg, transport, h->lv[h->fold - 1].starts);
    if (!st && l >= 1) st = pmg_mgmc_set_level_interpolation(mg, l, v.P_nrows, v.ncoarse_local, v.P_rp, v.P_ci, v.P_v);
    if (!st && !v.replicated) {
      st = pmg_mgmc_set_level_rowblock(mg, l, v.row0, v.nowned, v.ncolors, v.colors, v.send_ptr, v.send_rows, v.counts, v.recv_ptr, v.recv_src, v.recv_rows);
      if (!st) st = pmg_mgmc_set_level_restriction(mg, l, v.R_nrows, v.nlocal, v.R_rp, v.R_ci, v.R_v);
    }
  }
  if (st) {
    pmg_mgmc_destroy(&mg);
    return st;
  }
  *out = mg;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* stand-alone sampler on one row block (MCSORApply_MPIAIJ, src/mc_sor.c:298-381)                          */
/* ---------------------------------------------------------------------------------------------------- */
/* This rank's rows [row_starts[rank], row_starts[rank+1]) with GLOBAL columns (pmg_rowblock_merge_mpiaij), a colouring of
   them (colors_owned NULL: the library's first-fit rule on the global matrix), omega; builds the local operator (ghost
   columns as identity rows of a colour that is never swept), its device form with noise keyed on the global row, the
   ghost-update plan and the C sample loop on `transport`.  *mc_out is owned by the caller too (destroy the distmcsor
   first).  Collective; needs a GPU. */
pmg_status pmg_rowblock_sampler_create(const pmg_host_comm *comm, pmg_dist transport, const int64_t *row_starts, const void *rowptr, const void *colidx_global, const double *vals, int idx_width, int32_t ncolors, const int32_t *colors_owned, double omega, pmg_mcsor *mc_out, pmg_distmcsor *out)
{
  PMG_CALL(hc_check(comm));
  PMG_CHECK(transport && row_starts && mc_out && out, PMG_ERR_ARG_NULL, "null argument");
  *mc_out = NULL, *out = NULL;
  const int     me = comm->rank;
  const int64_t r0 = row_starts[me], nloc = row_starts[me + 1] - r0;
  gcsr          A;
  memset(&A, 0, sizeof A);
  pmg_rowblock_plan plan = NULL;
  int32_t          *cols = NULL, *rp = NULL, *ci = NULL, *lcol = NULL, *pos = NULL, *spos = NULL, *rpos = NULL;
  double           *v = NULL;
  pmg_mcsor         mc = NULL;
  pmg_status        st = gcsr_copy(&A, nloc, rowptr, colidx_global, vals, idx_width);
  st = hc_agree(comm, st, "copying the row block");
  if (!st && !colors_owned) {
    cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nloc > 0 ? nloc : 1));
    if (!cols) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    st = hc_agree(comm, st, "allocating the colouring");
    if (!st) st = pmg_rowblock_color_greedy(comm, row_starts, A.rp, A.ci, cols, &ncolors);
    colors_owned = cols;
  }
  if (!st) st = pmg_rowblock_plan_create(comm, row_starts, A.rp[nloc], A.ci, 0, NULL, ncolors, colors_owned, &plan);
  if (!st) st = hc_agree(comm, rowblock_check_coloring(plan, A.rp, A.ci, colors_owned), "checking the colouring");
  if (!st) { /* local operator */
    const int32_t ng = plan->nghost;
    const int64_t nnz = A.rp[nloc], nl = nloc + ng;
    rp   = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl + 1));
    ci   = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz + ng > 0 ? nnz + ng : 1));
    v    = (double *)malloc(sizeof(double) * (size_t)(nnz + ng > 0 ? nnz + ng : 1));
    lcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1));
    pos  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1));
    if (!rp || !ci || !v || !lcol || !pos) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st && nnz + ng >= INT32_MAX) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "too many entries for 32-bit indices");
    if (!st) {
      rbh_level tmp;
      memset(&tmp, 0, sizeof tmp);
      tmp.row0 = r0, tmp.nloc = nloc;
      for (int64_t r = 0; r <= nloc; ++r) rp[r] = (int32_t)A.rp[r];
      for (int64_t k = 0; k < nnz; ++k) ci[k] = (int32_t)local_of(&tmp, plan->ghosts, ng, A.ci[k]), v[k] = A.v[k];
      for (int32_t q = 0; q < ng; ++q) rp[nloc + q + 1] = (int32_t)(nnz + q + 1), ci[nnz + q] = (int32_t)(nloc + q), v[nnz + q] = 1.0;
      for (int64_t r = 0; r < nloc; ++r) lcol[r] = colors_owned[r];
      for (int32_t q = 0; q < ng; ++q) lcol[nloc + q] = ncolors;
    }
    if (!st) st = pmg_mcsor_create_csr((int32_t)nl, rp, ci, v, &mc);
    if (!st) st = pmg_mcsor_set_coloring(mc, PMG_COLORING_USER, lcol);
    if (!st) st = pmg_mcsor_set_omega(mc, omega);
    if (!st) st = pmg_mcsor_setup(mc);
    if (!st) st = pmg_mcsor_set_noise_row_offset(mc, r0);
    if (!st) st = pmg_mcsor_get_layout(mc, pos);
    if (!st) {
      const int64_t ns = plan->send_ptr[ncolors], nr = plan->recv_ptr[ncolors];
      spos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ns > 0 ? ns : 1));
      rpos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1));
      if (!spos || !rpos) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
      for (int64_t i = 0; i < ns && !st; ++i) spos[i] = pos[plan->send_rows[i]];
      for (int64_t i = 0; i < nr && !st; ++i) rpos[i] = pos[plan->recv_rows[i]];
    }
  }
  st = hc_agree(comm, st, "setting up the local operator");
  if (!st) st = pmg_distmcsor_create(mc, transport, ncolors, plan->send_ptr, spos, plan->counts, plan->recv_ptr, plan->recv_src, rpos, out);
  st = hc_agree(comm, st, "creating the row-block sampler");
  gcsr_free(&A);
  pmg_rowblock_plan_destroy(&plan);
  free(cols), free(lcol), free(pos), free(spos), free(rpos);
  if (st) {
    pmg_distmcsor_destroy(out);
    pmg_mcsor_destroy(&mc);
    free(rp), free(ci), free(v);
    return st;
  }
  /* pmg_mcsor_create_csr borrows its CSR arrays: hand them over */
  pmg_mcsor_adopt_arrays(mc, rp, ci, v);
  *mc_out = mc;
  return PMG_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- */
/* transports bootstrapped through the caller's all-gather                                                */
/* ---------------------------------------------------------------------------------------------------- */
/* kind = "ipc" (hipIpc peer stores + flag words) or "rccl" (ncclSend/ncclRecv; rccl_path = the librccl.so to dlopen, NULL:
   the default search).  g: this rank's slab for the grid samplers, or NULL for a pure transport (row blocks).  Every step
   that can fail on one rank alone is followed by an agreement, so no rank is left inside a collective.  Collective. */
pmg_status pmg_dist_create_comm(const pmg_host_comm *comm, const char *kind, pmg_grid g, const char *rccl_path, pmg_dist *out)
{
  PMG_CALL(hc_check(comm));
  PMG_CHECK(kind && out, PMG_ERR_ARG_NULL, "null argument");
  *out           = NULL;
  const int  np  = comm->nranks, me = comm->rank;
  pmg_dist   d   = NULL;
  pmg_status st  = PMG_SUCCESS;
  if (!strcmp(kind, "rccl")) {
    char id[128], *all = (char *)malloc((size_t)128 * (size_t)np);
    memset(id, 0, sizeof id);
    if (!all) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st && me == 0 && np > 1) st = pmg_dist_get_unique_id(rccl_path, id);
    st = hc_agree(comm, st, "creating the RCCL unique id");
    if (!st) st = hc_allgather(comm, id, 128, all);
    if (!st) st = pmg_dist_create(g, me, np, np > 1 ? all : NULL, rccl_path, 0, &d); /* rank 0's id */
    free(all);
    st = hc_agree(comm, st, "ncclCommInitRank");
  } else if (!strcmp(kind, "ipc")) {
    int32_t nb = 0;
    char   *blob = NULL, *all = NULL;
    st = pmg_dist_create_ipc(g, me, np, &d);
    if (!st) st = pmg_dist_ipc_blob_bytes(&nb);
    if (!st) {
      blob = (char *)calloc((size_t)nb, 1);
      all  = (char *)malloc((size_t)nb * (size_t)np);
      if (!blob || !all) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    }
    if (!st) st = pmg_dist_ipc_export(d, blob);
    st = hc_agree(comm, st, "ipc: allocating / exporting the receive block");
    if (!st) st = hc_allgather(comm, blob, nb, all);
    if (!st && np > 1) {
      st = pmg_dist_ipc_connect(d, me > 0 ? all + (size_t)nb * (size_t)(me - 1) : NULL, me < np - 1 ? all + (size_t)nb * (size_t)(me + 1) : NULL);
      if (!st && np > 2) {
        const void **bl = (const void **)malloc(sizeof(void *) * (size_t)np);
        if (!bl) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
        for (int r = 0; r < np && !st; ++r) bl[r] = all + (size_t)nb * (size_t)r;
        if (!st) st = pmg_dist_ipc_connect_all(d, bl);
        free(bl);
      }
    }
    free(blob), free(all);
    st = hc_agree(comm, st, "ipc: opening the peers' memory handles"); /* also the barrier in front of the first message */
  } else st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "transport '%s' (ipc | rccl)", kind);
  if (st) {
    /* Every rank gets here together (each branch ends in an agreement), but NOT every rank holds an object: the creation
       may have failed on some ranks only.  The barrier all-gather between "peers' blocks unmapped" and "own block freed"
       is therefore entered by every rank, with or without an object (round 3 guarded it by `if (d)`: the ranks whose
       creation had failed skipped a collective the others entered -- advisor finding); only the calls on the object are
       conditional.  An unknown `kind` fails identically everywhere before any collective and needs none. */
    if (strcmp(kind, "ipc") == 0 || strcmp(kind, "rccl") == 0) {
      int32_t z = 0, small[64], *all = np <= 64 ? small : (int32_t *)malloc(sizeof(int32_t) * (size_t)np);
      if (d) pmg_dist_ipc_disconnect(d);
      if (all) hc_allgather(comm, &z, sizeof z, all);
      if (all != small) free(all);
    }
    if (d) pmg_dist_destroy(&d);
    return st;
  }
  *out = d;
  return PMG_SUCCESS;
}

/* collective tear-down: unmap the peers' blocks, barrier (one all-gather), free the own block */
pmg_status pmg_dist_destroy_comm(const pmg_host_comm *comm, pmg_dist *d)
{
  PMG_CALL(hc_check(comm));
  if (d && *d) pmg_dist_ipc_disconnect(*d);
  PMG_HIP(hipDeviceSynchronize());
  int32_t z = 0, *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)comm->nranks);
  PMG_CHECK(all, PMG_ERR_MEM, "out of host memory");
  pmg_status st = hc_allgather(comm, &z, sizeof z, all);
  free(all);
  if (d) pmg_dist_destroy(d);
  return st;
}
