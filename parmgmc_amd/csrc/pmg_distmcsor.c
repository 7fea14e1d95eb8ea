/* MCSOR on a general AIJ matrix distributed by ROW BLOCKS, one rank per device -- host side (C11).
 *
 * Replaces MCSORApply_MPIAIJ (reference src/mc_sor.c:298-381): for every colour the ghost values are updated, then the
 * colour's rows are swept.  The reference builds one VecScatter per colour (MatCreateScatters, :152-214) whose ghost
 * buffer holds one entry per off-process NONZERO in row-visit order (:197-198, :332); here a rank's off-process COLUMNS
 * are ghost rows of its local sliced-ELL operator (identity rows of an extra, never swept colour), every value travels
 * once, and the whole loop -- colour sweeps and ghost updates -- runs in C on the caller's stream:
 *
 *   update of colour c:  gather  my rows of colour c that any other rank reads     (one kernel, index list)
 *                        all-gather over the halo transport of pmg_dist.c          (ipc: one push into every rank's
 *                                                                                    gather area + flag words; rccl: one
 *                                                                                    group of ncclSend / ncclRecv)
 *                        scatter what I read from the others into my ghost rows     (one kernel, index lists)
 *
 * The blocks of a colour tile one buffer in rank order, so the all-gather is the single-step form.  Every rank receives
 * every boundary value of the colour, not only its own ghosts: with the rank counts of one node (<= 8) and boundary
 * sets of a few thousand rows that is cheaper than a sparse all-to-all and reuses the transport as it stands.
 * Noise is keyed on the global row (pmg_mcsor_set_noise_row_offset) and the entries of a row keep the order of the
 * global CSR row, so the chain is the single-process chain bit for bit.
 */
#include "pmg_internal.h"

struct pmg_distmcsor_s {
  pmg_mcsor mc;   /* borrowed: local rows + ghost rows, set up */
  pmg_dist  dist; /* borrowed: transport (any pmg_dist object; its grid is not used) */
  int32_t   ncolors, nranks, rank;
  /* per colour c: my send list [send_ptr[c], send_ptr[c+1]) of layout positions; the gather buffer of the colour holds
     the ranks' blocks in rank order (offsets / counts [c*nranks + r]); my read list [recv_ptr[c], recv_ptr[c+1]) of
     (index in the colour's gather buffer, layout position of the ghost row) */
  int64_t *send_ptr, *recv_ptr, *goff, *gcnt, *gtot;
  int32_t *send_pos_dev, *recv_src_dev, *recv_pos_dev;
  double  *gbuf;
  int64_t  gcap;
  /* all colours at once (a refresh of every ghost row): rank r's block = its colour blocks one after the other, so the
     send list is the concatenation that is already stored; all_src[q] = index of ghost q's value in that buffer.
     all_tot = -1: the combined block exceeds the transport's capacity, refresh colour by colour */
  int64_t *all_off, *all_cnt, all_tot;
  int32_t *all_src_dev;
  pmg_lrc  lrc; /* MATLRC update A + B S B^T of the distributed operator (pmg_distmcsor_set_lowrank) */
  /* scratch of the natural-order entry points (pmg_distmcsor_sample / _apply): local vectors (owned rows, then ghosts) and their layout forms */
  double  *nat_b, *nat_y, *lay_b, *lay_y;
  int32_t  nlocal, ld;
};

pmg_status pmg_distmcsor_destroy(pmg_distmcsor *hp)
{
  if (!hp || !*hp) return PMG_SUCCESS;
  pmg_distmcsor h = *hp;
  pmg_lrc_destroy(&h->lrc);
  free(h->send_ptr);
  free(h->recv_ptr);
  free(h->goff);
  free(h->gcnt);
  free(h->gtot);
  free(h->all_off);
  free(h->all_cnt);
  pmg_dev_free(h->all_src_dev);
  pmg_dev_free(h->send_pos_dev);
  pmg_dev_free(h->recv_src_dev);
  pmg_dev_free(h->recv_pos_dev);
  pmg_dev_free(h->gbuf);
  pmg_dev_free(h->nat_b), pmg_dev_free(h->nat_y), pmg_dev_free(h->lay_b), pmg_dev_free(h->lay_y);
  free(h);
  *hp = NULL;
  return PMG_SUCCESS;
}

/* mc: the local operator (my rows with the ghost columns appended as identity rows of colour `ncolors`), set up.
   send_ptr[ncolors+1] / send_pos: layout positions of my rows of colour c that another rank reads, in the order of the
   colour's block; counts[c*nranks + r]: length of rank r's block of colour c (identical on every rank);
   recv_ptr[ncolors+1] / recv_src / recv_pos: for every ghost row that changes in colour c, its index in the colour's
   gather buffer (blocks in rank order) and its layout position.  All arrays are host arrays and are copied. */
pmg_status pmg_distmcsor_create(pmg_mcsor mc, pmg_dist dist, int32_t ncolors, const int64_t *send_ptr, const int32_t *send_pos, const int64_t *counts, const int64_t *recv_ptr, const int32_t *recv_src, const int32_t *recv_pos, pmg_distmcsor *out)
{
  PMG_CHECK(out, PMG_ERR_ARG_NULL, "null output handle");
  *out = NULL;
  PMG_CHECK(mc && dist && send_ptr && counts && recv_ptr, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(ncolors >= 1, PMG_ERR_ARG_OUTOFRANGE, "ncolors = %d", ncolors);
  int32_t mcols = 0, ld = 0;
  PMG_CALL(pmg_mcsor_get_num_colors(mc, &mcols));
  PMG_CALL(pmg_mcsor_layout_len(mc, &ld));
  PMG_CHECK(mcols == ncolors || mcols == ncolors + 1, PMG_ERR_ARG_SIZ, "the local operator has %d colours; expected %d swept colours (+ 1 for ghost rows)", mcols, ncolors);
  pmg_distmcsor h = (pmg_distmcsor)calloc(1, sizeof *h);
  PMG_CHECK(h, PMG_ERR_MEM, "out of host memory");
  h->mc      = mc;
  h->dist    = dist;
  h->ncolors = ncolors;
  pmg_status st = pmg_dist_get_info(dist, &h->rank, &h->nranks, &h->gcap);
  const size_t nc1 = (size_t)ncolors + 1, ncr = (size_t)ncolors * (size_t)(h->nranks > 0 ? h->nranks : 1);
  if (!st) {
    h->send_ptr = (int64_t *)malloc(sizeof(int64_t) * nc1);
    h->recv_ptr = (int64_t *)malloc(sizeof(int64_t) * nc1);
    h->goff     = (int64_t *)malloc(sizeof(int64_t) * ncr);
    h->gcnt     = (int64_t *)malloc(sizeof(int64_t) * ncr);
    h->gtot     = (int64_t *)malloc(sizeof(int64_t) * nc1);
    if (!h->send_ptr || !h->recv_ptr || !h->goff || !h->gcnt || !h->gtot) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  }
  int64_t maxtot = 0;
  if (!st) {
    memcpy(h->send_ptr, send_ptr, sizeof(int64_t) * nc1);
    memcpy(h->recv_ptr, recv_ptr, sizeof(int64_t) * nc1);
    memcpy(h->gcnt, counts, sizeof(int64_t) * ncr);
    for (int32_t c = 0; c < ncolors && !st; ++c) {
      int64_t off = 0;
      for (int32_t r = 0; r < h->nranks; ++r) {
        if (counts[(size_t)c * h->nranks + r] < 0) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "negative block length");
        h->goff[(size_t)c * h->nranks + r] = off;
        off += counts[(size_t)c * h->nranks + r];
      }
      h->gtot[c] = off;
      if (off > maxtot) maxtot = off;
      if (!st && send_ptr[c + 1] - send_ptr[c] != counts[(size_t)c * h->nranks + h->rank]) st = pmg_set_error(PMG_ERR_ARG_SIZ, __FILE__, __LINE__, "colour %d: my send list has %lld entries but my block %lld", c, (long long)(send_ptr[c + 1] - send_ptr[c]), (long long)counts[(size_t)c * h->nranks + h->rank]);
      if (!st && (send_ptr[c + 1] < send_ptr[c] || recv_ptr[c + 1] < recv_ptr[c])) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "list pointers must be monotone");
    }
    if (!st && (send_ptr[0] != 0 || recv_ptr[0] != 0)) st = pmg_set_error(PMG_ERR_ARG_WRONG, __FILE__, __LINE__, "list pointers must start at 0");
  }
  if (!st && maxtot > h->gcap) st = pmg_set_error(PMG_ERR_ARG_SIZ, __FILE__, __LINE__, "%lld boundary values of one colour exceed the exchange capacity of the transport (%lld)", (long long)maxtot, (long long)h->gcap);
  const int64_t ns = st ? 0 : send_ptr[ncolors], nr = st ? 0 : recv_ptr[ncolors];
  if (!st && ((ns > 0 && !send_pos) || (nr > 0 && (!recv_src || !recv_pos)))) st = pmg_set_error(PMG_ERR_ARG_NULL, __FILE__, __LINE__, "null index list");
  for (int64_t q = 0; q < ns && !st; ++q)
    if (send_pos[q] < 0 || send_pos[q] >= ld) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "send position %d outside the layout (%d)", send_pos[q], ld);
  for (int32_t c = 0; c < ncolors && !st; ++c)
    for (int64_t q = recv_ptr[c]; q < recv_ptr[c + 1] && !st; ++q)
      if (recv_pos[q] < 0 || recv_pos[q] >= ld || recv_src[q] < 0 || recv_src[q] >= h->gtot[c]) st = pmg_set_error(PMG_ERR_ARG_OUTOFRANGE, __FILE__, __LINE__, "colour %d: read entry %lld (source %d of %lld, position %d of %d) out of range", c, (long long)q, recv_src[q], (long long)h->gtot[c], recv_pos[q], ld);
  if (!st) st = pmg_dev_upload((void **)&h->send_pos_dev, send_pos, sizeof(int32_t) * (size_t)ns);
  if (!st) st = pmg_dev_upload((void **)&h->recv_src_dev, recv_src, sizeof(int32_t) * (size_t)nr);
  if (!st) st = pmg_dev_upload((void **)&h->recv_pos_dev, recv_pos, sizeof(int32_t) * (size_t)nr);
  /* the combined plan */
  int32_t *all_src = NULL;
  if (!st) {
    const size_t nrk = (size_t)(h->nranks > 0 ? h->nranks : 1);
    h->all_off = (int64_t *)calloc(nrk, sizeof(int64_t));
    h->all_cnt = (int64_t *)calloc(nrk, sizeof(int64_t));
    int64_t *pre = (int64_t *)calloc(ncr ? ncr : 1, sizeof(int64_t)); /* pre[c][r]: rank r's values of the colours before c */
    all_src      = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1));
    if (!h->all_off || !h->all_cnt || !pre || !all_src) st = pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
    if (!st) {
      for (int32_t c = 0; c < ncolors; ++c)
        for (int32_t r = 0; r < h->nranks; ++r) {
          pre[(size_t)c * h->nranks + r] = h->all_cnt[r];
          h->all_cnt[r] += counts[(size_t)c * h->nranks + r];
        }
      int64_t off = 0;
      for (int32_t r = 0; r < h->nranks; ++r) {
        h->all_off[r] = off;
        off += h->all_cnt[r];
      }
      h->all_tot = off <= h->gcap && off < ((int64_t)1 << 31) ? off : -1;
      for (int32_t c = 0; c < ncolors && h->all_tot >= 0; ++c)
        for (int64_t q = recv_ptr[c]; q < recv_ptr[c + 1]; ++q) {
          int32_t r = 0; /* the rank whose block of colour c holds the value */
          while (r + 1 < h->nranks && recv_src[q] >= h->goff[(size_t)c * h->nranks + r + 1]) ++r;
          all_src[q] = (int32_t)(h->all_off[r] + pre[(size_t)c * h->nranks + r] + (recv_src[q] - h->goff[(size_t)c * h->nranks + r]));
        }
    }
    free(pre);
  }
  if (!st && h->all_tot >= 0) st = pmg_dev_upload((void **)&h->all_src_dev, all_src, sizeof(int32_t) * (size_t)nr);
  free(all_src);
  if (!st && h->all_tot > maxtot) maxtot = h->all_tot;
  if (!st) st = pmg_dev_alloc((void **)&h->gbuf, sizeof(double) * (size_t)(maxtot > 0 ? maxtot : 1));
  if (st) {
    pmg_distmcsor_destroy(&h);
    return st;
  }
  *out = h;
  return PMG_SUCCESS;
}

/* ghost update for the rows of colour c: VecScatterBegin/End of src/mc_sor.c:318-319 */
static pmg_status distmcsor_update(pmg_distmcsor h, int32_t c, double *y, void *stream)
{
  if (h->gtot[c] == 0 || h->nranks == 1) return PMG_SUCCESS; /* identical on every rank */
  const int64_t *off = h->goff + (size_t)c * h->nranks, *cnt = h->gcnt + (size_t)c * h->nranks;
  const int64_t  ns = h->send_ptr[c + 1] - h->send_ptr[c], nr = h->recv_ptr[c + 1] - h->recv_ptr[c];
  if (ns > 0) PMG_KERNEL(pmgk_gather_idx(ns, h->send_pos_dev + h->send_ptr[c], y, h->gbuf + off[h->rank], stream));
  PMG_CALL(pmg_dist_allgather(h->dist, h->gbuf, off, cnt, stream));
  if (nr > 0) PMG_KERNEL(pmgk_scatter_idx(nr, h->recv_src_dev + h->recv_ptr[c], h->recv_pos_dev + h->recv_ptr[c], h->gbuf, y, stream));
  return PMG_SUCCESS;
}

/* every ghost row: ONE all-gather of all colours' boundary values where the transport can carry them, else colour by colour */
static pmg_status distmcsor_refresh(pmg_distmcsor h, double *y, void *stream)
{
  if (h->nranks == 1) return PMG_SUCCESS;
  static int by_colour = -1; /* PMG_DISTMCSOR_REFRESH_BY_COLOUR=1: the colour-by-colour form (same values) */
  if (by_colour < 0) by_colour = getenv("PMG_DISTMCSOR_REFRESH_BY_COLOUR") != NULL;
  if (h->all_tot < 0 || by_colour) {
    for (int32_t c = 0; c < h->ncolors; ++c) PMG_CALL(distmcsor_update(h, c, y, stream));
    return PMG_SUCCESS;
  }
  if (h->all_tot == 0) return PMG_SUCCESS;
  const int64_t ns = h->send_ptr[h->ncolors], nr = h->recv_ptr[h->ncolors];
  if (ns > 0) PMG_KERNEL(pmgk_gather_idx(ns, h->send_pos_dev, y, h->gbuf + h->all_off[h->rank], stream));
  PMG_CALL(pmg_dist_allgather(h->dist, h->gbuf, h->all_off, h->all_cnt, stream));
  if (nr > 0) PMG_KERNEL(pmgk_scatter_idx(nr, h->all_src_dev, h->recv_pos_dev, h->gbuf, y, stream));
  return PMG_SUCCESS;
}

/* with_lrc = 0: the sweeps of A alone (what MCSORBuildLRCCorrection applies to the columns of B) */
static pmg_status distmcsor_sweeps_x(pmg_distmcsor h, const double *b, double *y, int32_t its, int noisy, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, int with_lrc, void *stream)
{
  PMG_CHECK(h && b && y, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(its >= 0, PMG_ERR_ARG_OUTOFRANGE, "its = %d", its);
  PMG_CHECK(pmg_sweep_type_ok(sweep_type), PMG_ERR_SUP, "Only forward, backward and symmetric sweep supported");
  pmg_lrc lrc = with_lrc ? h->lrc : NULL;
  PMG_CALL(distmcsor_refresh(h, y, stream)); /* the caller's y has no ghost values yet */
  uint64_t ctr = counter0;
  for (int32_t it = 0; it < its; ++it) {
    const int ndir = sweep_type == PMG_SOR_SYMMETRIC_SWEEP ? 2 : 1;
    for (int q = 0; q < ndir; ++q) {
      const int     dir = ndir == 2 ? (q == 0 ? PMG_SOR_FORWARD_SWEEP : PMG_SOR_BACKWARD_SWEEP) : sweep_type;
      const double *rhs = b;
      if (lrc && noisy) PMG_CALL(pmg_lrc_rhs(lrc, b, seed, ctr, &rhs, stream)); /* + B (sqrt(S) o eta), src/pc_mcgibbs.c:130-140: eta is keyed on (seed, counter), the same on every rank */
      for (int32_t cc = 0; cc < h->ncolors; ++cc) {
        const int32_t c = dir == PMG_SOR_FORWARD_SWEEP ? cc : h->ncolors - 1 - cc; /* src/mc_sor.c:317, :344 */
        PMG_CALL(pmg_mcsor_sweep_color_layout(h->mc, c, noisy, scaled, seed, ctr, rhs, y, stream));
        PMG_CALL(distmcsor_update(h, c, y, stream));
      }
      if (lrc && noisy) PMG_CALL(pmg_lrc_rhs_done(lrc, stream));
      /* y -= Bb (B^T y), src/mc_sor.c:101-112: B has zeros on the ghost rows (every row counts once in the all-reduced
         k-vector), Bb carries the owners' values there, so the ghost rows receive their owners' update without an exchange */
      if (lrc) PMG_CALL(pmg_lrc_post(lrc, dir, y, stream));
      ++ctr; /* a symmetric sweep draws twice per sample, src/pc_mcgibbs.c:172-181 */
    }
  }
  if (counter_out) *counter_out = ctr;
  return PMG_SUCCESS;
}

static pmg_status distmcsor_sweeps(pmg_distmcsor h, const double *b, double *y, int32_t its, int noisy, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  return distmcsor_sweeps_x(h, b, y, its, noisy, scaled, sweep_type, seed, counter0, counter_out, 1, stream);
}

static pmg_status distmcsor_det_sweep(void *ctx, int dir, const double *b, double *y, void *stream)
{
  return distmcsor_sweeps_x((pmg_distmcsor)ctx, b, y, 1, 0, 0, dir, 0, 0, NULL, 0, stream);
}
static pmg_status distmcsor_reduce(void *ctx, double *vals_dev, int count, void *stream)
{
  return pmg_dist_allreduce_sum(((pmg_distmcsor)ctx)->dist, vals_dev, count, stream);
}

/* MATLRC operator A + B S B^T on row blocks (MCSORSetUp's LRC branch, src/mc_sor.c:572-595, on a MATMPIAIJ base): B_lay_dev
   is ld x k column-major in the LAYOUT of the local operator on the device, this rank's rows filled and ZERO on the ghost
   rows; S the k diagonal entries.  The correction Bb = C (S^-1 + B^T C)^-1, C = M^-1 B (:480-544) is built with distributed
   deterministic sweeps and rank-ordered all-reduces of the k x k products.  Collective.  k = 0 removes the update. */
pmg_status pmg_distmcsor_set_lowrank_dev(pmg_distmcsor h, int32_t k, const double *B_lay_dev, const double *S_host)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  pmg_lrc_destroy(&h->lrc);
  if (k == 0) return PMG_SUCCESS;
  PMG_CHECK(B_lay_dev && S_host, PMG_ERR_ARG_NULL, "null low-rank factor");
  PMG_CHECK(k > 0 && k <= 64, PMG_ERR_ARG_OUTOFRANGE, "rank k = %d (1..64 supported)", k);
  int32_t ld = 0;
  PMG_CALL(pmg_mcsor_layout_len(h->mc, &ld));
  return pmg_lrc_build_dev(&h->lrc, k, ld, B_lay_dev, S_host, distmcsor_det_sweep, h, h->nranks > 1 ? distmcsor_reduce : NULL, h);
}

/* the same from the host: B_local is nlocal x k column-major in the local row numbering (nlocal = rows of the local
   operator, owned rows first; the ghost rows' entries are ignored) */
pmg_status pmg_distmcsor_set_lowrank(pmg_distmcsor h, int32_t k, int32_t nlocal, int32_t nowned, const double *B_local_host, const double *S_host)
{
  PMG_CHECK(h, PMG_ERR_ARG_NULL, "null handle");
  if (k == 0) return pmg_distmcsor_set_lowrank_dev(h, 0, NULL, NULL);
  PMG_CHECK(B_local_host && S_host && k > 0 && k <= 64, PMG_ERR_ARG_WRONG, "low-rank factor: k = %d", k);
  int32_t       ld = 0;
  const int32_t n  = nlocal; /* rows of the local operator (owned + ghost), the leading dimension of B_local */
  PMG_CALL(pmg_mcsor_layout_len(h->mc, &ld));
  PMG_CHECK(nowned >= 0 && nowned <= n && n <= ld, PMG_ERR_ARG_OUTOFRANGE, "%d owned rows of %d local rows (layout %d)", nowned, n, ld);
  int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  double  *Bl  = (double *)calloc((size_t)ld * (size_t)k, sizeof(double));
  pmg_status st = (pos && Bl) ? pmg_mcsor_get_layout(h->mc, pos) : pmg_set_error(PMG_ERR_MEM, __FILE__, __LINE__, "out of host memory");
  double *Bd = NULL;
  if (!st) {
    for (int32_t c = 0; c < k; ++c)
      for (int32_t r = 0; r < nowned; ++r) Bl[(size_t)ld * c + pos[r]] = B_local_host[(size_t)n * c + r];
    st = pmg_dev_upload((void **)&Bd, Bl, sizeof(double) * (size_t)ld * (size_t)k);
  }
  free(pos);
  free(Bl);
  if (!st) st = pmg_distmcsor_set_lowrank_dev(h, k, Bd, S_host);
  pmg_dev_free(Bd);
  return st;
}

/* r = b - (A + B S B^T) y on the owned rows of layout vectors whose ghost rows are current */
pmg_status pmg_distmcsor_residual_layout(pmg_distmcsor h, const double *b_lay, const double *y_lay, double *r_lay, void *stream)
{
  PMG_CHECK(h && b_lay && y_lay && r_lay, PMG_ERR_ARG_NULL, "null argument");
  PMG_CALL(pmg_mcsor_residual_layout(h->mc, b_lay, y_lay, r_lay, stream));
  if (h->lrc) PMG_CALL(pmg_lrc_residual_sub(h->lrc, y_lay, r_lay, stream));
  return PMG_SUCCESS;
}

/* every ghost row of a layout vector from its owner (all colours' updates): the vector may be any level vector -- a
   residual whose restriction reads other ranks' rows, an iterate after an interpolation.  Collective. */
pmg_status pmg_distmcsor_refresh_layout(pmg_distmcsor h, double *v_lay, void *stream)
{
  PMG_CHECK(h && v_lay, PMG_ERR_ARG_NULL, "null argument");
  return distmcsor_refresh(h, v_lay, stream);
}

/* `its` samples of the mcgibbs / sorgibbs chain on layout vectors (b, y: my rows filled; the ghost entries of y are
   refreshed here).  Collective: every rank makes the same call. */
pmg_status pmg_distmcsor_sample_layout(pmg_distmcsor h, const double *b_lay, double *y_lay, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  return distmcsor_sweeps(h, b_lay, y_lay, its, 1, scaled, sweep_type, seed, counter0, counter_out, stream);
}

/* The same on NATURAL-order vectors of this rank's owned rows (what a Vec of a MATMPIAIJ holds): MCSORApply / the sample
   loops of src/pc_mcgibbs.c:155-188 for callers that do not keep their vectors in the layout.  nowned = number of owned
   rows (the first rows of the local operator).  y's owned rows are the chain's state: in and out. */
static pmg_status distmcsor_natural(pmg_distmcsor h, int32_t nowned, const double *b_owned, double *y_owned, int32_t its, int noisy, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  PMG_CHECK(h && (nowned == 0 || (b_owned && y_owned)), PMG_ERR_ARG_NULL, "null argument");
  if (!h->lay_b) {
    PMG_CALL(pmg_mcsor_layout_len(h->mc, &h->ld));
    PMG_CALL(pmg_mcsor_get_size(h->mc, &h->nlocal));
    const size_t nb = sizeof(double) * (size_t)(h->nlocal > 0 ? h->nlocal : 1), lb = sizeof(double) * (size_t)(h->ld > 0 ? h->ld : 1);
    PMG_CALL(pmg_dev_alloc((void **)&h->nat_b, nb));
    PMG_CALL(pmg_dev_alloc((void **)&h->nat_y, nb));
    PMG_CALL(pmg_dev_alloc((void **)&h->lay_b, lb));
    PMG_CALL(pmg_dev_alloc((void **)&h->lay_y, lb));
  }
  PMG_CHECK(nowned >= 0 && nowned <= h->nlocal, PMG_ERR_ARG_OUTOFRANGE, "%d owned rows of %d local rows", nowned, h->nlocal);
  hipStream_t s = (hipStream_t)stream;
  if (nowned) {
    PMG_HIP(hipMemcpyAsync(h->nat_b, b_owned, sizeof(double) * (size_t)nowned, hipMemcpyDeviceToDevice, s));
    PMG_HIP(hipMemcpyAsync(h->nat_y, y_owned, sizeof(double) * (size_t)nowned, hipMemcpyDeviceToDevice, s));
  }
  PMG_CALL(pmg_mcsor_to_layout(h->mc, h->nat_b, h->lay_b, stream));
  PMG_CALL(pmg_mcsor_to_layout(h->mc, h->nat_y, h->lay_y, stream));
  PMG_CALL(distmcsor_sweeps(h, h->lay_b, h->lay_y, its, noisy, scaled, sweep_type, seed, counter0, counter_out, stream));
  PMG_CALL(pmg_mcsor_from_layout(h->mc, h->lay_y, h->nat_y, stream));
  if (nowned) PMG_HIP(hipMemcpyAsync(y_owned, h->nat_y, sizeof(double) * (size_t)nowned, hipMemcpyDeviceToDevice, s));
  return PMG_SUCCESS;
}

pmg_status pmg_distmcsor_sample(pmg_distmcsor h, int32_t nowned, const double *b_owned_dev, double *y_owned_dev, int32_t its, int scaled, int sweep_type, uint64_t seed, uint64_t counter0, uint64_t *counter_out, void *stream)
{
  return distmcsor_natural(h, nowned, b_owned_dev, y_owned_dev, its, 1, scaled, sweep_type, seed, counter0, counter_out, stream);
}

pmg_status pmg_distmcsor_apply(pmg_distmcsor h, int32_t nowned, const double *b_owned_dev, double *y_owned_dev, int sweep_type, void *stream)
{
  return distmcsor_natural(h, nowned, b_owned_dev, y_owned_dev, 1, 0, 0, sweep_type, 0, 0, NULL, stream);
}

/* MCSORApply: one deterministic sweep of the given type */
pmg_status pmg_distmcsor_apply_layout(pmg_distmcsor h, const double *b_lay, double *y_lay, int sweep_type, void *stream)
{
  return distmcsor_sweeps(h, b_lay, y_lay, 1, 0, 0, sweep_type, 0, 0, NULL, stream);
}
