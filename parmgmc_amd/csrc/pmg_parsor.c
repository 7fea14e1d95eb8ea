/* PCPARSOR's multi-rank sweep as a data-flow graph (host side, C11).
 *
 * ParallelSORApply (reference src/pc_parsor.c:703-878) lets every MPI rank sweep its rows in the order
 * TOP, INT1, MID, INT2, BOT (ParallelSORPartitionNodes, :272-592) and fixes, by the points at which ghost values
 * travel, whether a row sees the value of an off-rank neighbour from BEFORE the iteration or the one computed IN it:
 *   - a row and a neighbour on the same rank: the neighbour's new value iff the neighbour comes earlier in that order
 *     (rows ascending inside a phase; adjacent MID rows wait for each other in ascending order, :547-557);
 *   - a TOP row (off-rank neighbours on lower-coloured ranks only) reads their old values (`topsct`, :720-723);
 *   - a MID row reads new values from higher-coloured ranks (their TOP rows through `botsct` :739,745, their MID rows
 *     through the messages it waits for :451-456,826-846) and old values from lower-coloured ranks;
 *   - a BOT row (neighbours on higher-coloured ranks only) reads the new value of a TOP neighbour and, of a MID
 *     neighbour, the new value iff some MID row of its own rank references that neighbour too -- only then does the
 *     message overwrite the ghost slot (:441-456,841-846) -- and the old one otherwise.
 * That is all one needs to reproduce the result on one device, whatever the number of ranks emulated: rows become
 * nodes of a DAG ("reads the new value of"), its longest-path levels become the colours of a multicolour sweep
 * (pmg_mcsor with a USER colouring; rows of a level are independent by construction), and an "old" read of a
 * neighbour that the level order would already have overwritten is redirected to a snapshot of x taken at the start
 * of the iteration -- the matrix handed to pmg_mcsor has 2n rows, rows n..2n-1 are identity rows in an extra colour that
 * is never swept and hold that snapshot.  Row sums run over the entries of the rank's diagonal block, then over its
 * off-diagonal block, as SORLocalForwardSweepIS does (:666-701).
 *
 * The reference colours the ranks with PETSc's randomised JP colouring (:240-243), which no reference test pins;
 * here the caller's colouring or first-fit in rank order.  The pattern must be structurally symmetric (the
 * reference's MID hand-shake assumes it: a sender's row must be known to the receiver, :841-842).
 */
#include "pmg_internal.h"
#include <math.h>

enum { PH_TOP = 0, PH_INT1 = 1, PH_MID = 2, PH_INT2 = 3, PH_BOT = 4 };
enum { CL_INT = 0, CL_TOP = 1, CL_MID = 2, CL_BOT = 3 };

typedef struct {
  int32_t        n, nparts;
  const int32_t *rowptr, *colidx;
  int32_t       *owner, *cls, *phase, *pcol;
  int64_t       *midkeys; /* sorted keys rank * n + column: ghost columns that receive MID messages on that rank */
  int64_t        nmidkeys;
} flow;

static int cmp_i64(const void *a, const void *b)
{
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : x > y;
}

/* does row i read the value row j gets in THIS iteration? */
static int reads_new(const flow *F, int32_t i, int32_t j)
{
  const int32_t p = F->owner[i], q = F->owner[j];
  if (p == q) return F->phase[j] < F->phase[i] || (F->phase[j] == F->phase[i] && j < i);
  switch (F->cls[i]) {
  case CL_TOP: return 0;
  case CL_MID: return F->pcol[q] > F->pcol[p] && (F->cls[j] == CL_TOP || F->cls[j] == CL_MID);
  case CL_BOT:
    if (F->cls[j] == CL_TOP) return 1;
    if (F->cls[j] == CL_MID) {
      const int64_t key = (int64_t)p * F->n + j;
      return bsearch(&key, F->midkeys, (size_t)F->nmidkeys, sizeof(int64_t), cmp_i64) != NULL;
    }
    return 0;
  default: return 0;
  }
}

static void flow_free(flow *F)
{
  free(F->owner);
  free(F->cls);
  free(F->phase);
  free(F->pcol);
  free(F->midkeys);
}

pmg_status pmg_parsor_build_dataflow(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int32_t nparts, const int32_t *row_starts, const int32_t *proccols_in, int32_t **e_rowptr, int32_t **e_colidx, double **e_vals, int32_t **e_colors, int32_t *nlevels_out, int32_t *proccols_out, int32_t *classes_out)
{
  PMG_CHECK(n > 0 && rowptr && colidx && vals && row_starts, PMG_ERR_ARG_NULL, "null argument");
  PMG_CHECK(nparts >= 1 && nparts <= 4096, PMG_ERR_ARG_OUTOFRANGE, "nparts = %d", nparts);
  PMG_CHECK(row_starts[0] == 0 && row_starts[nparts] == n, PMG_ERR_ARG_WRONG, "row_starts must run from 0 to n");
  for (int32_t p = 0; p < nparts; ++p) PMG_CHECK(row_starts[p + 1] > row_starts[p], PMG_ERR_ARG_WRONG, "rank %d owns no rows", p);
  flow F;
  memset(&F, 0, sizeof F);
  F.n = n, F.nparts = nparts, F.rowptr = rowptr, F.colidx = colidx;
  F.owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  F.cls   = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  F.phase = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  F.pcol  = (int32_t *)malloc(sizeof(int32_t) * (size_t)nparts);
  unsigned char *padj = (unsigned char *)calloc((size_t)nparts * (size_t)nparts, 1);
  int32_t       *level = (int32_t *)calloc((size_t)n, sizeof(int32_t)), *indeg = (int32_t *)calloc((size_t)n, sizeof(int32_t)), *queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  pmg_status     st = PMG_SUCCESS;
#define PARSOR_FAIL(code, ...) \
  do { \
    st = pmg_set_error((code), __FILE__, __LINE__, __VA_ARGS__); \
    goto done; \
  } while (0)
  if (!F.owner || !F.cls || !F.phase || !F.pcol || !padj || !level || !indeg || !queue) PARSOR_FAIL(PMG_ERR_MEM, "out of host memory");
  for (int32_t p = 0; p < nparts; ++p)
    for (int32_t r = row_starts[p]; r < row_starts[p + 1]; ++r) F.owner[r] = p;
  /* structural symmetry + the rank graph of ColorProcessors (:207-217) */
  for (int32_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t j = colidx[k];
      if (j < 0 || j >= n) PARSOR_FAIL(PMG_ERR_ARG_OUTOFRANGE, "column %d out of range in row %d", j, i);
      if (j == i) continue;
      int found = 0;
      for (int32_t k2 = rowptr[j]; k2 < rowptr[j + 1] && !found; ++k2) found = colidx[k2] == i;
      if (!found) PARSOR_FAIL(PMG_ERR_ARG_WRONG, "the pattern is not symmetric: (%d,%d) stored, (%d,%d) not", i, j, j, i);
      if (F.owner[i] != F.owner[j]) padj[(size_t)F.owner[i] * nparts + F.owner[j]] = 1;
    }
  if (proccols_in) {
    for (int32_t p = 0; p < nparts; ++p) F.pcol[p] = proccols_in[p];
  } else { /* first fit in rank order */
    for (int32_t p = 0; p < nparts; ++p) {
      int32_t c = 0;
      for (int again = 1; again;) {
        again = 0;
        for (int32_t q = 0; q < p; ++q)
          if (padj[(size_t)p * nparts + q] && F.pcol[q] == c) {
            ++c;
            again = 1;
          }
      }
      F.pcol[p] = c;
    }
  }
  for (int32_t p = 0; p < nparts; ++p)
    for (int32_t q = 0; q < nparts; ++q)
      if (padj[(size_t)p * nparts + q] && F.pcol[p] == F.pcol[q]) PARSOR_FAIL(PMG_ERR_ARG_WRONG, "adjacent ranks %d and %d have the same colour %d", p, q, F.pcol[p]);
  /* ParallelSORPartitionNodes :311-331 */
  int64_t nmid_entries = 0;
  for (int32_t i = 0; i < n; ++i) {
    int istop = 0, isbot = 0;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t q = F.owner[colidx[k]];
      if (q == F.owner[i]) continue;
      if (F.pcol[q] < F.pcol[F.owner[i]]) istop = 1;
      if (F.pcol[q] > F.pcol[F.owner[i]]) isbot = 1;
    }
    F.cls[i] = !istop && !isbot ? CL_INT : !istop ? CL_BOT : !isbot ? CL_TOP : CL_MID;
    if (F.cls[i] == CL_MID) nmid_entries += rowptr[i + 1] - rowptr[i];
  }
  /* the INT1 / INT2 split :353-362: its costs are row lengths of the OFF-DIAGONAL block, 0 for every INT row, so the
     loop never breaks (all INT rows in INT1) unless the target is negative (all in INT2) */
  for (int32_t p = 0; p < nparts; ++p) {
    int64_t topcost = 0, botcost = 0;
    for (int32_t i = row_starts[p]; i < row_starts[p + 1]; ++i) {
      int64_t off = 0;
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) off += F.owner[colidx[k]] != p;
      if (F.cls[i] == CL_TOP) topcost += off;
      if (F.cls[i] == CL_BOT) botcost += off;
    }
    const int int_phase = roundf(0.5f * (float)(botcost - topcost)) < 0.0f ? PH_INT2 : PH_INT1;
    for (int32_t i = row_starts[p]; i < row_starts[p + 1]; ++i) F.phase[i] = F.cls[i] == CL_TOP ? PH_TOP : F.cls[i] == CL_MID ? PH_MID : F.cls[i] == CL_BOT ? PH_BOT : int_phase;
  }
  /* ghost columns that receive MID messages (global_to_lvec, :451-456) */
  F.midkeys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nmid_entries > 0 ? nmid_entries : 1));
  if (!F.midkeys) PARSOR_FAIL(PMG_ERR_MEM, "out of host memory");
  for (int32_t i = 0; i < n; ++i) {
    if (F.cls[i] != CL_MID) continue;
    const int32_t p = F.owner[i];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t j = colidx[k], q = F.owner[j];
      if (q != p && F.pcol[q] > F.pcol[p] && F.cls[j] == CL_MID) F.midkeys[F.nmidkeys++] = (int64_t)p * n + j;
    }
  }
  qsort(F.midkeys, (size_t)F.nmidkeys, sizeof(int64_t), cmp_i64);
  /* longest-path levels of "reads the new value of" (Kahn) */
  for (int32_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (colidx[k] != i && reads_new(&F, i, colidx[k])) indeg[i]++;
  int32_t head = 0, tail = 0, nlevels = 0;
  for (int32_t i = 0; i < n; ++i)
    if (!indeg[i]) queue[tail++] = i;
  while (head < tail) {
    const int32_t j = queue[head++];
    if (level[j] + 1 > nlevels) nlevels = level[j] + 1;
    for (int32_t k = rowptr[j]; k < rowptr[j + 1]; ++k) { /* symmetric pattern: the readers of j are among its columns */
      const int32_t i = colidx[k];
      if (i == j || !reads_new(&F, i, j)) continue;
      if (level[j] + 1 > level[i]) level[i] = level[j] + 1;
      if (--indeg[i] == 0) queue[tail++] = i;
    }
  }
  if (tail != n) PARSOR_FAIL(PMG_ERR_ARG_WRONG, "the sweep order has a cycle (%d of %d rows scheduled): the reference's MID phase would dead-lock", tail, n);
  /* the 2n x 2n matrix: [A with redirected columns; identity], colours = levels, snapshot rows in colour nlevels */
  {
    const int64_t nnz  = rowptr[n];
    int32_t      *rp   = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * (int64_t)n + 1));
    int32_t      *ci   = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz + n));
    double       *va   = (double *)malloc(sizeof(double) * (size_t)(nnz + n));
    int32_t      *cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * (int64_t)n));
    if (!rp || !ci || !va || !cols) {
      free(rp), free(ci), free(va), free(cols);
      PARSOR_FAIL(PMG_ERR_MEM, "out of host memory");
    }
    int64_t w = 0;
    for (int32_t i = 0; i < n; ++i) {
      rp[i] = (int32_t)w;
      for (int pass = 0; pass < 2; ++pass) /* the rank's diagonal block first, then its off-diagonal block */
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
          const int32_t j = colidx[k];
          if ((F.owner[j] != F.owner[i]) != pass) continue;
          const int old_copy = pass && !reads_new(&F, i, j) && level[j] <= level[i];
          ci[w]              = old_copy ? n + j : j;
          va[w++]            = vals[k];
        }
      cols[i] = level[i];
    }
    for (int32_t i = 0; i < n; ++i) {
      rp[n + i]   = (int32_t)w;
      ci[w]       = n + i;
      va[w++]     = 1.0;
      cols[n + i] = nlevels;
    }
    rp[2 * n] = (int32_t)w;
    *e_rowptr = rp, *e_colidx = ci, *e_vals = va, *e_colors = cols;
  }
  *nlevels_out = nlevels;
  if (proccols_out) memcpy(proccols_out, F.pcol, sizeof(int32_t) * (size_t)nparts);
  if (classes_out) memcpy(classes_out, F.cls, sizeof(int32_t) * (size_t)n);
done:
#undef PARSOR_FAIL
  flow_free(&F);
  free(padj);
  free(level);
  free(indeg);
  free(queue);
  return st;
}
